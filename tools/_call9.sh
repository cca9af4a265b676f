#!/bin/bash
out=gpurun_out/r06_shim; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_shim_trace.py tests/test_gpu_ops.py -m gpu -q -s -x > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log; tail -40 $out/pytest.log | cut -c1-300
