#!/bin/bash
out=gpurun_out/r06_suite1; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $out/pytest.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log; tail -3 $out/smoke.log
