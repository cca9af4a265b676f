#!/bin/bash
out=gpurun_out/r06_tests2; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_refdata.py -m gpu -q -s -k "c3_100nfe or f16x1_reduced" > $out/pytest.log 2>&1; tail -4 $out/pytest.log | cut -c1-300; grep -E "C3 sr x4 100|f16x1 on the 5" $out/pytest.log | cut -c1-330
