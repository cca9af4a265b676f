#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command itself (eager launches: a traced 100-step graph loop aborted inside
# rocprofv3 in round 1).  usage: tools/gpu_prof_bench.sh <tag>  -> gpurun_out/<tag>/bench_kernel_stats.txt (+ the bench line)
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
d=/tmp/prof_bench; rm -rf $d
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $d -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-graph --no-cpu-baseline --no-c3 --no-alt --steps 1 --warmup 0) > $out/bench_traced.json 2> $out/bench_traced.err
db=$(find $d -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_summary.py $db --top 30 > $out/bench_kernel_stats.txt 2>&1; fi
find $d -name "*stats*" | head -5
for f in $(find $d -name "*kernel_stats*.csv" | head -1); do cp $f $out/bench_kernel_stats.csv; done
head -12 $out/bench_kernel_stats.txt; grep -o '"avg_launch_ms": [0-9.]*' $out/bench_traced.json | head -2
