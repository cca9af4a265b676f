#!/bin/bash
# One gpurun call: GPU parity suite + a short bench.  usage: tools/gpu_round.sh <tag> [pytest args]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -s "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $out/pytest.log | tail -15
DIFFPIR_CONV=4 timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -s -k "b2_8nfe and f16x3" > $out/pytest_conv4.log 2>&1; grep -E "vs oracle|passed|failed" $out/pytest_conv4.log | tail -3
timeout 300 python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline > $out/bench_quick.json 2> $out/bench_quick.err; tail -c 700 $out/bench_quick.json
