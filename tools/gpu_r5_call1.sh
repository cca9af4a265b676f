#!/bin/bash
# Round-5 first measurement call: (a) VALU issue-rate probe, (b) SQ counters on the FFT prox kernels, (c) split-K rule A/B on the low-res convs.
out=$PWD/gpurun_out/r5a; mkdir -p $out
export TMPDIR=/tmp
timeout 120 tools/micro/build/valu_probe > $out/valu_issue_probe.log 2>&1; echo "rc=$?" >> $out/valu_issue_probe.log
cat $out/valu_issue_probe.log
pmc() { # name, counters...
  name=$1; shift
  d=/tmp/prof_$name; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $d -o $name -- python $GRAFT_REPO_ROOT/tools/prox_bench.py) > $out/$name.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db --top 12 > $out/$name.txt 2>&1; else echo "no db" > $out/$name.txt; tail -5 $out/$name.log >> $out/$name.txt; fi
  grep -A8 "rfft_rows\|cfft_cols\|irfft_rows" $out/$name.txt | head -80
}
pmc prox_pmc_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU
pmc prox_pmc_sq2 SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM
pmc prox_pmc_sq3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE
timeout 100 python tools/prox_bench.py > $out/prox_bench_plain.log 2>&1; grep prox $out/prox_bench_plain.log
for cfg in "384 512" "256 256" "512 1024" "384 768" "384 512"; do
  set -- $cfg
  echo "== DPIR_SPLIT_BELOW=$1 DPIR_SPLIT_TARGET=$2" | tee -a $out/split_rule_ab.log
  DPIR_SPLIT_BELOW=$1 DPIR_SPLIT_TARGET=$2 timeout 120 python tools/layer_roofline.py 16 2>&1 | grep -v amdgpu.ids | grep "@  32\|@  16\|@   8\|sum" | tee -a $out/split_rule_ab.log | cut -c1-150
  DPIR_SPLIT_BELOW=$1 DPIR_SPLIT_TARGET=$2 RUN_LABEL="below=$1 target=$2" timeout 120 python tools/forward_time.py 2>/dev/null | tail -1 | tee -a $out/split_rule_ab.log | cut -c1-300
done
