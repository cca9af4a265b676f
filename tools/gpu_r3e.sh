#!/bin/bash
# targeted correctness of the conv5 / conv6 rewrites at full size (GroupNorm-table and plane-emitting conv5 variants, dgrad), then a
# kernel trace of the FFHQ forward
tag=${1:-r3o}
out=$PWD/gpurun_out/$tag
mkdir -p $out
timeout 700 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dps.py -x -q -k "forward or b16 or overflow or topology or gradient_full_size" > $out/pytest_targeted.log 2>&1
tail -4 $out/pytest_targeted.log
export TMPDIR=/tmp PROF_MODEL=ffhq PROF_B=16 PROF_SF=1 DIFFPIR_PRECISION=f16x3
d=/tmp/prof_kt; rm -rf $d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $d -o kt -- python $GRAFT_REPO_ROOT/tools/prof_forward.py) > $out/kt.log 2>&1
db=$(find $d -name "*.db" | head -1)
python tools/rocpd_summary.py $db --top 40 > $out/ffhq_f16x3_kernel_trace.txt 2>&1
head -30 $out/ffhq_f16x3_kernel_trace.txt
