"""Steady-state time of one UNet forward (B=16, 256x256 by default) with the per-class breakdown; DIFFPIR_LIB selects a kernel
variant build.  usage: python tools/forward_time.py [model] [B] [H]   (GPU box only)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import script_util, weights
name = sys.argv[1] if len(sys.argv) > 1 else "ffhq"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
eng = diffpir_amd.Engine(0)
eng.set_precision(os.environ.get("DIFFPIR_PRECISION", "f16x3"))
hp = weights.model_hp(name)
model = script_util.create_model(**weights.create_model_kwargs(hp), engine=eng)
model.load_state_dict(weights.synth_state_dict(hp, 0))
x = eng.to_device(np.random.default_rng(0).standard_normal((B, 3, H, H)).astype(np.float32))
t = np.full(B, 500)
y = np.arange(B) % 1000 if hp.class_cond else None
out = eng.unet_forward(x, t, y)
for _ in range(3):
    eng.unet_forward(x, t, y, out=out)
eng.sync()
n = 8
t0 = time.perf_counter()
for _ in range(n):
    eng.unet_forward(x, t, y, out=out)
eng.sync()
wall = (time.perf_counter() - t0) / n * 1e3
eng.prof_enable(True); eng.prof_reset()
for _ in range(3):
    eng.unet_forward(x, t, y, out=out)
eng.sync()
prof = eng.prof_read(); eng.prof_enable(False)
fl = eng.unet_flops(H, H) * B
cls = {k: round(v[0] / 3, 3) for k, v in prof.items() if v[1]}
print(f"{os.environ.get('RUN_LABEL', os.environ.get('DIFFPIR_LIB', 'default')):40s} fwd {wall:7.3f} ms  {fl / wall / 1e9:6.1f} TF/s  frac833 {fl / wall / 1e9 / 833.3:5.3f} | {cls}")
