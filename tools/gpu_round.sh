#!/bin/bash
# One gpurun call: GPU parity suite + smoke + the default bench.  usage: tools/gpu_round.sh <tag> [pytest args]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -s "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $out/pytest.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log; tail -5 $out/smoke.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json; tail -3 $out/bench.err
