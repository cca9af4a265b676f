#!/bin/bash
# SQ counters on the FFT prox kernels (three separate --pmc passes over tools/prox_bench.py, never combined with other trace domains) + the VALU
# issue-rate probe.  usage: tools/gpu_prox_counters.sh <tag>   -> gpurun_out/<tag>/prox_pmc_sq{1,2,3}.txt, valu_issue_probe.log, prox_bench.log
# (build the probe first: hipcc --offload-arch=gfx950 -O3 -o tools/micro/build/valu_probe tools/micro/valu_issue_probe.hip)
out=$PWD/gpurun_out/${1:-proxpmc}; mkdir -p $out
export TMPDIR=/tmp
[ -x tools/micro/build/valu_probe ] && timeout 120 tools/micro/build/valu_probe > $out/valu_issue_probe.log 2>&1
pmc() { name=$1; shift; d=/tmp/prof_$name; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $d -o $name -- python $GRAFT_REPO_ROOT/tools/prox_bench.py) > $out/$name.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db --top 12 > $out/$name.txt 2>&1; else echo "no db" > $out/$name.txt; fi; }
pmc prox_pmc_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU
pmc prox_pmc_sq2 SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM
pmc prox_pmc_sq3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE
timeout 100 python tools/prox_bench.py > $out/prox_bench.log 2>&1; grep prox $out/prox_bench.log
