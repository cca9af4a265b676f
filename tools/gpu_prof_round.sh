#!/bin/bash
# The round's rocprofv3 passes (kernel trace and one PMC pass per counter group, never combined with other trace domains) over
# tools/prof_forward.py for: FFHQ f16x3 B=16 (headline), ImageNet-256 f16x3 B=32 + sf=4 prox (config 3), FFHQ f32, FFHQ f16x1.
# usage: tools/gpu_prof_round.sh <tag> [headline]   -> gpurun_out/<tag>/<case>_<pass>.txt   ("headline": only the FFHQ f16x3 passes)
# Copy the summaries into profiles/rNN/ and run `python tools/pmc_traffic.py profiles/rNN` to refresh profiles/pmc_traffic.json.
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
run() { # case, pass, rocprof args...
  name=$1_$2; shift; shift
  d=/tmp/prof_$name; rm -rf $d
  (cd /tmp && timeout 500 rocprofv3 "$@" -d $d -o $name -- python $GRAFT_REPO_ROOT/tools/prof_forward.py) > $out/$name.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db --top 40 > $out/$name.txt 2>&1; else echo "no db" > $out/$name.txt; tail -5 $out/$name.log >> $out/$name.txt; fi
  echo "$name: $(head -1 $out/$name.txt)"
}
export PROF_MODEL=ffhq PROF_B=16 PROF_SF=1 DIFFPIR_PRECISION=f16x3
run ffhq_f16x3 kernel_trace --kernel-trace
run ffhq_f16x3 pmc_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run ffhq_f16x3 pmc_FETCH_SIZE --kernel-trace --pmc FETCH_SIZE
run ffhq_f16x3 pmc_WRITE_SIZE --kernel-trace --pmc WRITE_SIZE
cp $GRAFT_REPO_ROOT/.commit_id $out/commit.txt 2>/dev/null || true
# configs[4]'s data step alone: 512 x 512, sf = 4, 8 images
export PROF_UNET=0 PROF_SIZE=512 PROF_B=8 PROF_SF=4
run prox512 kernel_trace --kernel-trace
run prox512 pmc_FETCH_SIZE --kernel-trace --pmc FETCH_SIZE
run prox512 pmc_WRITE_SIZE --kernel-trace --pmc WRITE_SIZE
export PROF_UNET=1 PROF_SIZE=256 PROF_B=16 PROF_SF=1
[ "$2" = "headline" ] && exit 0
export DIFFPIR_PRECISION=f32
run ffhq_f32 kernel_trace --kernel-trace
run ffhq_f32 pmc_FETCH_SIZE --kernel-trace --pmc FETCH_SIZE
run ffhq_f32 pmc_WRITE_SIZE --kernel-trace --pmc WRITE_SIZE
export DIFFPIR_PRECISION=f16x1
run ffhq_f16x1 kernel_trace --kernel-trace
export PROF_MODEL=imagenet256 PROF_B=32 PROF_SF=4 DIFFPIR_PRECISION=f16x3
run in256_f16x3 kernel_trace --kernel-trace
run in256_f16x3 pmc_FETCH_SIZE --kernel-trace --pmc FETCH_SIZE
run in256_f16x3 pmc_WRITE_SIZE --kernel-trace --pmc WRITE_SIZE
cp $GRAFT_REPO_ROOT/.commit_id $out/commit.txt 2>/dev/null || true
