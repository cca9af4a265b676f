mkdir -p gpurun_out/r06_512
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "wave_kernels or prox or philox or device_noise" > gpurun_out/r06_512/ops.log 2>&1; tail -15 gpurun_out/r06_512/ops.log
for m in 0 1; do DPIR_PROX_MODE=$m python tools/prox_bench.py > gpurun_out/r06_512/prox_bench_mode$m.log 2>&1; cat gpurun_out/r06_512/prox_bench_mode$m.log | grep prox; done
