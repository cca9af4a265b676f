#!/bin/bash
# FIRST GPU call of round 4 (prepared at the end of round 3, when the GPU budget was gone):
#   1. conv7x (test-only generalisation of conv7: all geometries, split-K, f16x1, dgrad scale, idle co-halves) against conv6, bit for bit;
#   2. conv7 on / off in ONE call on the FFHQ forward (round 3 only has the two numbers from different boxes) and on ImageNet-256;
#   3. the MFMA-busy PMC pass with conv7 on.
tag=${1:-r4a}; out=$PWD/gpurun_out/$tag; mkdir -p $out
timeout 120 python tools/conv7x_check.py 5 > $out/conv7x_check.log 2>&1; echo "conv7x_check rc=$?" >> $out/conv7x_check.log; tail -20 $out/conv7x_check.log
for v in 1 0 1 0; do
  RUN_LABEL=conv7=$v DPIR_CONV7=$v timeout 60 python tools/forward_time.py ffhq 16 256 2>&1 | tail -1 | tee -a $out/forward_ab_conv7.log
done
for v in 1 0; do
  RUN_LABEL=imagenet256_b8_conv7=$v DPIR_CONV7=$v timeout 120 python tools/forward_time.py imagenet256 8 256 2>&1 | tail -1 | tee -a $out/forward_ab_conv7.log
done
export TMPDIR=/tmp PROF_MODEL=ffhq PROF_B=16 PROF_SF=1 DIFFPIR_PRECISION=f16x3
d=/tmp/prof_pmc7; rm -rf $d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $d -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_forward.py) > $out/pmc.log 2>&1
db=$(find $d -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db --top 12 > $out/ffhq_f16x3_conv7_pmc_mfma.txt 2>&1
grep -A5 "conv7_mfma_kernel$" $out/ffhq_f16x3_conv7_pmc_mfma.txt | head -8
