#!/bin/bash
# rocprofv3 --kernel-trace of the bench command ITSELF (graph replays included: ~59 000 dispatches for one warm-up + one timed
# 100-NFE pass at B = 16; round 1's rocprofv3 aborted on such a loop, ROCm 7.x does not) next to the line that run prints, so the
# per-kernel averages can be held against bench.py's own HIP-event figures (`roofline.avg_launch_ms`,
# `roofline.class_ms_per_forward_instrumented`).  usage: tools/gpu_prof_bench.sh <tag>
#   -> gpurun_out/<tag>/bench_kernel_trace.txt, bench_under_rocprof.json   (copy to profiles/rNN/bench_py_*)
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
d=/tmp/prof_bench; rm -rf $d
(cd /tmp && timeout 330 rocprofv3 --kernel-trace -d $d -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-c3 --no-alt --no-cpu-baseline) > $out/bench_under_rocprof.log 2>&1
echo "rc=$?" >> $out/bench_under_rocprof.log
db=$(find $d -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_summary.py $db --top 45 > $out/bench_kernel_trace.txt 2>&1; else echo "no db" > $out/bench_kernel_trace.txt; fi
grep -h '"metric"' $out/bench_under_rocprof.log | tail -1 > $out/bench_under_rocprof.json
head -12 $out/bench_kernel_trace.txt | cut -c1-200
grep -o '"avg_launch_ms": [0-9.]*' $out/bench_under_rocprof.json | head -1
cp $GRAFT_REPO_ROOT/.commit_id $out/commit.txt 2>/dev/null || true
