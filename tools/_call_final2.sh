set -x
mkdir -p gpurun_out/r06_final2
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_final2/pytest_gpu_full.log 2>&1
tail -3 gpurun_out/r06_final2/pytest_gpu_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final2/smoke.log 2>&1; tail -2 gpurun_out/r06_final2/smoke.log
python bench.py > gpurun_out/r06_final2/bench.log 2>&1; tail -1 gpurun_out/r06_final2/bench.log | cut -c1-400
export TMPDIR=/tmp; cd /tmp && rm -rf /tmp/prof_bench && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-alt --no-c3 > $GRAFT_REPO_ROOT/gpurun_out/r06_final2/bench_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; db=$(find /tmp/prof_bench -name "*.db" | head -1); python tools/rocpd_summary.py $db --top 40 > gpurun_out/r06_final2/bench_py_kernel_trace.txt 2>&1; head -5 gpurun_out/r06_final2/bench_py_kernel_trace.txt | cut -c1-200
tail -1 gpurun_out/r06_final2/bench_under_rocprof.log | cut -c1-300
