set -x
mkdir -p gpurun_out/r06_final
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -s -k "philox" > gpurun_out/r06_final/philox.log 2>&1
tail -15 gpurun_out/r06_final/philox.log
python -m pytest tests -m gpu -q > gpurun_out/r06_final/pytest_gpu_full.log 2>&1
tail -5 gpurun_out/r06_final/pytest_gpu_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final/smoke.log 2>&1; tail -3 gpurun_out/r06_final/smoke.log
python bench.py > gpurun_out/r06_final/bench.log 2>&1; tail -1 gpurun_out/r06_final/bench.log
