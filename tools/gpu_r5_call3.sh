#!/bin/bash
# Round-5 third call: new / changed tests, conv8 (branch-free) A/B, kernel trace, bench.
out=$PWD/gpurun_out/r5c; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_refdata.py tests/test_gpu_benched_batches.py tests/test_gpu_unet.py -m gpu -q --maxfail=40 -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|fused output layer|vs the reference's main|C1 box" $out/pytest.log | cut -c1-330 | tail -30
for rep in 1 2; do
for cfg in "0 384 512" "1 384 512" "1 256 256"; do
  set -- $cfg
  DPIR_CONV8=$1 DPIR_SPLIT_BELOW=$2 DPIR_SPLIT_TARGET=$3 RUN_LABEL="conv8=$1 below=$2 target=$3" timeout 120 python tools/forward_time.py 2>/dev/null | tail -1 | tee -a $out/forward_ab.log | cut -c1-260
done
done
export TMPDIR=/tmp
d=/tmp/prof_fwd; rm -rf $d
(cd /tmp && DPIR_SPLIT_BELOW=256 DPIR_SPLIT_TARGET=256 timeout 300 rocprofv3 --kernel-trace -d $d -o fwd -- python $GRAFT_REPO_ROOT/tools/prof_forward.py) > $out/fwd_trace.log 2>&1
db=$(find $d -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db --top 40 > $out/ffhq_f16x3_kernel_trace.txt 2>&1
head -30 $out/ffhq_f16x3_kernel_trace.txt | cut -c1-170
DPIR_SPLIT_BELOW=256 DPIR_SPLIT_TARGET=256 timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 2500 $out/bench.json; tail -3 $out/bench.err
