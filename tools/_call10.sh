#!/bin/bash
out=gpurun_out/r06_wino; mkdir -p $out
timeout 300 python tools/winograd_bound.py 16 > $out/winograd_bound.log 2>&1; cat $out/winograd_bound.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_dps.py tests/test_gpu_fullsize.py -m gpu -q -s -k "repeatable or c3_100nfe" > $out/pytest.log 2>&1; tail -5 $out/pytest.log | cut -c1-300; grep "C3 sr x4 100" $out/pytest.log | cut -c1-300
