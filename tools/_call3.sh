#!/bin/bash
out=gpurun_out/r06_fused1; mkdir -p $out
for v in 12 2 3 13; do
DPIR_PROX_VARIANT=$v RUN_LABEL=v$v timeout 300 python tools/prox_modes_check.py 12 > $out/check_v$v.log 2>&1; grep -E "B=|Error|error" $out/check_v$v.log | head -12
done
