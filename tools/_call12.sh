#!/bin/bash
out=$PWD/gpurun_out/r06_f16x1; mkdir -p $out
export TMPDIR=/tmp
DIFFPIR_PRECISION=f16x1 timeout 600 python tools/layer_roofline.py 16 > $out/layer_roofline_ffhq_b16_f16x1.log 2>&1; tail -45 $out/layer_roofline_ffhq_b16_f16x1.log | cut -c1-150
run() { name=$1; shift; d=/tmp/prof_$name; rm -rf $d
  (cd /tmp && timeout 500 rocprofv3 "$@" -d $d -o $name -- python $GRAFT_REPO_ROOT/tools/prof_forward.py) > $out/$name.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db --top 14 > $out/$name.txt 2>&1; else echo "no db" > $out/$name.txt; fi; echo "$name: $(head -1 $out/$name.txt)"; }
export PROF_MODEL=ffhq PROF_B=16 PROF_SF=1 DIFFPIR_PRECISION=f16x1
run ffhq_f16x1_pmc_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run ffhq_f16x1_pmc_wait --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVES
run ffhq_f16x1_pmc_FETCH_SIZE --kernel-trace --pmc FETCH_SIZE
run ffhq_f16x1_pmc_WRITE_SIZE --kernel-trace --pmc WRITE_SIZE
sed -n '/PMC sums/,/^void dpir::conv5/p' $out/ffhq_f16x1_pmc_mfma.txt | head -40 | cut -c1-140
