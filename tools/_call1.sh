#!/bin/bash
out=gpurun_out/r06_base; mkdir -p $out
timeout 200 python tools/prox_bench.py > $out/prox_bench.log 2>&1; grep prox $out/prox_bench.log
RUN_LABEL=churn timeout 900 python tools/dps_repeat.py 60 1 > $out/dps_repeat_churn.log 2>&1; tail -5 $out/dps_repeat_churn.log
RUN_LABEL=nochurn timeout 600 python tools/dps_repeat.py 40 0 > $out/dps_repeat_nochurn.log 2>&1; tail -3 $out/dps_repeat_nochurn.log
