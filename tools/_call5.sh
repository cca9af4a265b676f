#!/bin/bash
out=$PWD/gpurun_out/r06_wave3; mkdir -p $out
export TMPDIR=/tmp
prof() { name=$1; shift; d=/tmp/prof_$name; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace "$@" -d $d -o $name -- python $GRAFT_REPO_ROOT/tools/prox_prof.py) > $out/$name.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db --top 8 > $out/$name.txt 2>&1; else echo "no db" > $out/$name.txt; fi; head -12 $out/$name.txt | cut -c1-200; }
PROX_MODE=wave PROX_B=16 prof wave_b16_trace
PROX_MODE=wave PROX_B=64 prof wave_b64_trace
PROX_MODE=launches PROX_B=16 prof old_b16_trace
PROX_MODE=wave PROX_B=16 prof wave_b16_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES
PROX_MODE=wave PROX_B=16 prof wave_b16_sq2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
PROX_MODE=wave PROX_B=16 prof wave_b16_fetch --pmc FETCH_SIZE
PROX_MODE=wave PROX_B=16 prof wave_b16_write --pmc WRITE_SIZE
grep -A12 "PMC" $out/wave_b16_sq1.txt | cut -c1-150 | head -60
