#!/bin/bash
# kernel trace of the FFHQ forward with conv7 on the product path (the last seconds of the round's GPU budget)
tag=${1:-r3u}; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp PROF_MODEL=ffhq PROF_B=16 PROF_SF=1 DIFFPIR_PRECISION=f16x3
d=/tmp/prof_kt7; rm -rf $d
(cd /tmp && timeout 40 rocprofv3 --kernel-trace -d $d -o kt -- python $GRAFT_REPO_ROOT/tools/prof_forward.py) > $out/kt.log 2>&1
db=$(find $d -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db --top 40 > $out/ffhq_f16x3_conv7_kernel_trace.txt 2>&1
head -12 $out/ffhq_f16x3_conv7_kernel_trace.txt
