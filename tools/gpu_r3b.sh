#!/bin/bash
# round-3 call: new fused kernels -- parity subset, A/B forward timing, short bench
out=gpurun_out/r3b; mkdir -p $out
timeout 500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_loop.py -m gpu -q -x -s > $out/pytest_a.log 2>&1; echo "rc=$?" >> $out/pytest_a.log
grep -E "passed|failed|rel err|Error|error" $out/pytest_a.log | tail -30
timeout 500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "b16 or ffhq_256_forward or c2_loop_matches or b4_at_256 or imagenet256_topology_at_256" > $out/pytest_b.log 2>&1; echo "rc=$?" >> $out/pytest_b.log
grep -E "passed|failed|rel err|vs oracle|vs LIVE|Error" $out/pytest_b.log | tail -30
for f in 0 1; do for m in 0 1; do DPIR_FUSE_SMALL=$f DPIR_EMIT_SKIP=$m RUN_LABEL=fuse$f-emit$m timeout 100 python tools/forward_time.py 2>&1 | grep fwd; done; done | tee $out/forward_ab.log
timeout 300 python bench.py --no-cpu-baseline --no-c3 --no-alt --steps 3 --warmup 1 > $out/bench_short.json 2> $out/bench_short.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r3b/bench_short.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('unet_forward_ms'), d['roofline'].get('unet_step_frac'), d['roofline'].get('frac'))
P
tail -3 $out/bench_short.err
