#!/bin/bash
out=gpurun_out/r3c; mkdir -p $out
for f in 0 1; do for m in 0 1; do DPIR_FUSE_SMALL=$f DPIR_EMIT_SKIP=$m RUN_LABEL=fuse$f-emit$m timeout 100 python tools/forward_time.py 2>&1 | grep fwd; done; done | tee $out/forward_ab.log
timeout 300 python -m pytest tests/test_gpu_unet.py -m gpu -q -x -s -k "f16x3 or tiny" > $out/pytest_a.log 2>&1; echo "rc=$?" >> $out/pytest_a.log
grep -E "passed|failed|Error|error" $out/pytest_a.log | tail -8
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_concurrency.py tests/test_gpu_degrade.py -m gpu -q -s > $out/pytest_b.log 2>&1; echo "rc=$?" >> $out/pytest_b.log
grep -E "passed|failed|Error|error|FAILED" $out/pytest_b.log | tail -15
timeout 700 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "c3_20nfe or (imagenet512 and f16x3) or b16" > $out/pytest_c.log 2>&1; echo "rc=$?" >> $out/pytest_c.log
grep -E "passed|failed|rel err|vs oracle|vs LIVE|Error|FAILED" $out/pytest_c.log | tail -20
timeout 300 python bench.py --no-cpu-baseline --no-c3 --no-alt --steps 3 --warmup 1 > $out/bench_short.json 2> $out/bench_short.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r3c/bench_short.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('unet_forward_ms'), d['roofline'].get('unet_step_frac'), d['roofline'].get('frac'), d.get('degrade_metrics'), d['config'])
P
tail -3 $out/bench_short.err
