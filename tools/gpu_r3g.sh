#!/bin/bash
# conv7 integration: bit-equality against conv6 (every residual form + fused statistics), the B = 16 full-size parity tests with conv7
# on the product path (the small-size tests never reach it: their launches are split-K), one forward timing.  Sized for the last
# GPU-minutes of the round.
tag=${1:-r3t}; out=gpurun_out/$tag; mkdir -p $out
timeout 50 python tools/conv7_check.py 5 > $out/conv7_check.log 2>&1; echo "conv7_check rc=$?" >> $out/conv7_check.log; tail -20 $out/conv7_check.log
timeout 90 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "benched_batch_b16 and f16x3" > $out/pytest_b16.log 2>&1; echo "pytest rc=$?" >> $out/pytest_b16.log; grep -E "ffhq 256|passed|failed|rc=" $out/pytest_b16.log | tail -6
RUN_LABEL=conv7 timeout 40 python tools/forward_time.py ffhq 16 256 2>&1 | tail -1 | tee $out/forward_conv7.log
