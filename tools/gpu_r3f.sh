#!/bin/bash
# after the conv5 / conv6 pipeline fixes: the multi-GPU presets on one GPU, then the MFMA-busy PMC pass over the FFHQ forward
tag=${1:-r3q}
out=$PWD/gpurun_out/$tag; mkdir -p $out
bash tools/gpu_bench_presets.sh $tag
export TMPDIR=/tmp PROF_MODEL=ffhq PROF_B=16 PROF_SF=1 DIFFPIR_PRECISION=f16x3
d=/tmp/prof_pmc; rm -rf $d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $d -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_forward.py) > $out/pmc.log 2>&1
db=$(find $d -name "*.db" | head -1)
python tools/rocpd_summary.py $db --top 12 > $out/ffhq_f16x3_pmc_mfma.txt 2>&1
head -16 $out/ffhq_f16x3_pmc_mfma.txt
