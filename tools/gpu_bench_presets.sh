#!/bin/bash
# bench.py on the multi-GPU presets with ONE GPU (for the record: the driver's N > 1 runs use c4 by default): c4 = BASELINE configs[3]
# per-GPU work (FFHQ, motion PSF, B = 32), c5 = configs[4] per-GPU work (512^2 class-conditional, x4 SISR, B = 8).
out=gpurun_out/$1; mkdir -p $out
timeout 600 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $out/bench_c4_1gpu.json 2> $out/bench_c4.err; tail -c 600 $out/bench_c4_1gpu.json | head -c 300; echo
timeout 900 python bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $out/bench_c5_1gpu.json 2> $out/bench_c5.err; tail -3 $out/bench_c5.err
python - <<P
import json
for c in ("c4", "c5"):
    try:
        d = json.loads(open("$out/bench_%s_1gpu.json" % c).read().strip().splitlines()[-1])
        print(c, d["value"], d["ms_per_step"], d["config"]["workload"][:90], d["roofline"]["frac"], d["roofline"]["unet_step_frac"], (d.get("roofline_prox") or {}).get("frac"))
    except Exception as ex:
        print(c, "failed", ex)
P
