#!/bin/bash
out=gpurun_out/r06_stagger; mkdir -p $out
for st in "0,0,0" "8,0,0" "16,0,0" "32,0,0" "0,16,0" "0,32,0" "0,64,0" "0,0,16" "0,0,32" "16,32,16" "32,64,32"; do
DPIR_FFT4_STAGGER=$st PROX_MODES=wave RUN_LABEL=st$st timeout 120 python tools/prox_modes_check.py 3 > $out/check_$st.log 2>&1; grep -E "B=16 256\^2 sf=1|B=64|Error|rror" $out/check_$st.log | head -3 | cut -c1-160
done
