#!/bin/bash
# rocprofv3 passes over tools/prof_forward.py: kernel trace, then one PMC pass per counter group (never combined with other
# trace domains).  usage: tools/gpu_prof.sh <tag>   -> gpurun_out/<tag>/{kernel_trace,pmc_*}.txt
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
run() { # name, rocprof args...
  name=$1; shift
  d=/tmp/prof_$name; rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 "$@" -d $d -o $name -- python $GRAFT_REPO_ROOT/tools/prof_forward.py) > $out/$name.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db --top 40 > $out/$name.txt 2>&1; else echo "no db" > $out/$name.txt; find $d | head >> $out/$name.txt; fi
  tail -2 $out/$name.log
}
run kernel_trace --kernel-trace
run pmc_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run pmc_FETCH_SIZE --kernel-trace --pmc FETCH_SIZE
run pmc_WRITE_SIZE --kernel-trace --pmc WRITE_SIZE
run pmc_lds --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
run pmc_wait --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
head -30 $out/kernel_trace.txt
