#!/bin/bash
# Round-5 second call: full GPU parity suite on the regenerated (reference-executed) fixtures + conv8 A/B + split-rule variants.
out=$PWD/gpurun_out/r5b; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|fused output layer|DPS_yt" $out/pytest.log | tail -25
for rep in 1 2; do
for cfg in "0 384 512" "1 384 512" "1 256 256" "1 256 320" "1 320 320" "1 200 256"; do
  set -- $cfg
  DPIR_CONV8=$1 DPIR_SPLIT_BELOW=$2 DPIR_SPLIT_TARGET=$3 RUN_LABEL="conv8=$1 below=$2 target=$3" timeout 120 python tools/forward_time.py 2>/dev/null | tail -1 | tee -a $out/forward_ab.log | cut -c1-260
done
done
export TMPDIR=/tmp
d=/tmp/prof_fwd; rm -rf $d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $d -o fwd -- python $GRAFT_REPO_ROOT/tools/prof_forward.py) > $out/fwd_trace.log 2>&1
db=$(find $d -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db --top 40 > $out/ffhq_f16x3_kernel_trace.txt 2>&1
head -14 $out/ffhq_f16x3_kernel_trace.txt | cut -c1-170; grep conv8 $out/ffhq_f16x3_kernel_trace.txt | cut -c1-170
