#!/bin/bash
# One gpurun call: GPU parity suite + a short bench + kernel trace of a forward.  usage: tools/gpu_round.sh <tag> [pytest args]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -s "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline > $out/bench_quick.json 2> $out/bench_quick.err; tail -c 1500 $out/bench_quick.json
