#!/bin/bash
bash tools/gpu_evidence_round.sh r06 > gpurun_out/r06_evidence.log 2>&1
bash tools/gpu_prox_counters.sh r06 >> gpurun_out/r06_evidence.log 2>&1
PROX_MODES=launches,wave RUN_LABEL=final timeout 300 python tools/prox_modes_check.py 8 > gpurun_out/r06/prox_modes_check.log 2>&1
DIFFPIR_PRECISION=f16x3 timeout 600 python tools/layer_roofline.py 16 > gpurun_out/r06/layer_roofline_ffhq_b16.log 2>&1
tail -c 1500 gpurun_out/r06/bench_1gpu.json; grep "B=" gpurun_out/r06/prox_modes_check.log | cut -c1-300
