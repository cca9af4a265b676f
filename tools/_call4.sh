#!/bin/bash
out=gpurun_out/r06_fft5c; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/xcc_probe tools/xcc_probe.hip 2>/dev/null; /tmp/xcc_probe > $out/xcc_probe.log 2>&1; head -4 $out/xcc_probe.log
for sy in 0 1 2 3; do
DPIR_PROX_SYNC=$sy DPIR_FFT5_WGS=2 PROX_MODES=wave,wave_fused RUN_LABEL=sync$sy timeout 300 python tools/prox_modes_check.py 24 > $out/check_$sy.log 2>&1; grep -E "B=|Error|error|rror" $out/check_$sy.log | head -4 | sed 's/wave evt.*| wave_fused/wave_fused/' | cut -c1-200
done
