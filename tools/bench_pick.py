"""Print the few numbers of a bench.py JSON line that A/B runs compare.  usage: python bench.py ... | python tools/bench_pick.py"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
p = d.get("roofline_prox") or {}
print(json.dumps({"images_per_s": d["value"], "ms_per_step": d["ms_per_step"], "conv_frac": r.get("frac"), "unet_forward_ms": r.get("unet_forward_ms"),
                  "unet_step_frac": r.get("unet_step_frac"), "prox_us": p.get("us_per_apply"), "prox_frac": p.get("frac"),
                  "fused_us": (p.get("fused_data_step") or {}).get("us_per_step"), "classes": r.get("class_ms_per_forward_instrumented")}))
