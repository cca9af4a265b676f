"""Experiment: L independent image pipelines ("lanes": engine + stream + step graph each) on ONE GPU, B/L images per lane,
against one pipeline of B images.  The loop is a serial chain per image, and the low-resolution third of the UNet cannot fill
256 CUs at B=16, so lanes that are out of phase fill each other's gaps.  GPU box only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import restore, synth, script_util, weights

B = int(os.environ.get("LANES_B", "16")); NFE = int(os.environ.get("LANES_NFE", "30")); H = 256
model = os.environ.get("LANES_MODEL", "ffhq")
hp = weights.model_hp(model)
sd = weights.synth_state_dict(hp, 0)
cfg = restore.LoopConfig(task="deblur", iter_num=NFE, lambda_=7.0, zeta=0.3)
case = synth.make_case("deblur", B, H, H, seed=100, ksize=61)

def mk():
    e = diffpir_amd.Engine(0); e.set_precision("f16x3")
    m = script_util.create_model(**weights.create_model_kwargs(hp), engine=e); m.load_state_dict(sd)
    return e

engines = []
ref = None
for L in [int(v) for v in os.environ.get("LANES_LIST", "1,2,4").split(",")]:
    while len(engines) < L:
        engines.append(mk())
    n = B // L
    lanes = []
    for l in range(L):
        e = engines[l]; sl = slice(l * n, (l + 1) * n)
        lanes.append(dict(e=e, y=e.to_device(case["y"][sl]), k=e.to_device(case["k"][sl]), o=e.empty((n, 3, H, H)), keep={}, off=l * n))
    def run():
        for ln in lanes:
            restore.restore_batch(ln["e"], cfg, ln["y"], k=ln["k"], noise_source="device", seed=1234, image_offset=ln["off"],
                                  use_graph=True, out_f32=ln["o"], _cache=ln["keep"])
        for ln in lanes:
            ln["e"].sync()
    run()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    out = np.concatenate([ln["o"].numpy() for ln in lanes])
    if ref is None:
        ref = out
    t = min(ts)
    print(f"lanes={L} x B={n}: {t*1e3:.1f} ms for {NFE} NFE -> {t/NFE*1e3:.3f} ms/step, {B/(t*100/NFE):.3f} images/s @100NFE; "
          f"max|diff| vs 1 lane {np.abs(out-ref).max():.3e}; all {[round(x*1e3,1) for x in ts]}", flush=True)
