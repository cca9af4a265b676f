#!/bin/bash
out=gpurun_out/r06_dps; mkdir -p $out
for p in nan big one; do
DPS_POISON=$p RUN_LABEL=poison_$p timeout 600 python tools/dps_repeat.py 6 0 > $out/dps_poison_$p.log 2>&1; tail -4 $out/dps_poison_$p.log
done
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|DPS_y0 FFHQ" $out/pytest.log | tail -15
