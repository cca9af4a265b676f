#!/bin/bash
# Interleaved A/B of ONE environment switch inside ONE gpurun call (box-to-box spread is ~4 %: comparisons across calls are worthless).
# usage: tools/gpu_ab.sh <logfile> <ENVVAR> <reps> <command ...>     runs the command with ENVVAR=1,0,1,0,... and appends the last
# output line of each run (prefixed with the setting) to <logfile>.
log=$1; var=$2; reps=$3; shift 3
mkdir -p $(dirname $log)
for i in $(seq 1 $reps); do
  for v in 1 0; do
    line=$(env $var=$v RUN_LABEL="$var=$v" timeout 300 "$@" 2>/dev/null | tail -1)
    echo "$var=$v $line" | tee -a $log | cut -c1-400
  done
done
