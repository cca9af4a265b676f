"""Who is the victim when FFT-prox kernels and f16 UNet kernels of two engines overlap?  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import restore, synth, script_util, weights, utils_sisr as sr
H = 256
hp = weights.model_hp("ffhq"); sd = weights.synth_state_dict(hp, 0)
def mk(prec):
    e = diffpir_amd.Engine(0); e.set_precision(prec)
    m = script_util.create_model(**weights.create_model_kwargs(hp), engine=e); m.load_state_dict(sd)
    return e
case = synth.make_case("deblur", 8, H, H, seed=100, ksize=61)
x = np.random.default_rng(0).standard_normal((8, 3, H, H)).astype(np.float32)
for prec in ("f16x3", "f32"):
    eu, ef = mk(prec), mk(prec)                     # eu: UNet forwards, ef: FFT prox only
    xu = eu.to_device(x); t = np.full(8, 500)
    ru = eu.unet_forward(xu, t); eu.sync(); ru_np = ru.numpy()
    ou = eu.empty(ru_np.shape)
    y, k = ef.to_device(case["y"]), ef.to_device(case["k"])
    pre = sr.pre_calculate(y, k, 1, engine=ef)
    x0h = (case["gt"] * 2 - 1).astype(np.float32)
    def prox_once(buf):
        buf.copy_from(x0h)
        ef._check(ef.lib.dpir_prox_fft_apply(ef.h, pre[0].spectra.handle, buf.ptr, 7e-7, 1.0))
    b0 = ef.empty(x0h.shape); prox_once(b0); ef.sync(); ref_p = b0.numpy()
    bufs = [ef.empty(x0h.shape) for _ in range(40)]
    # concurrent: 3 forwards on eu, 40 prox applications on ef
    for b in bufs: b.copy_from(x0h)
    ef.sync()
    for _ in range(3): eu.unet_forward(xu, t, out=ou)
    for b in bufs: ef._check(ef.lib.dpir_prox_fft_apply(ef.h, pre[0].spectra.handle, b.ptr, 7e-7, 1.0))
    eu.sync(); ef.sync()
    dp = [float(np.abs(b.numpy() - ref_p).max()) for b in bufs]
    print(f"[{prec}] UNet under concurrent FFT: max|diff| {np.abs(ou.numpy() - ru_np).max():.3e};  FFT prox under concurrent UNet: "
          f"worst {max(dp):.3e}, #changed {sum(d > 0 for d in dp)}/40 (prox output scale {np.abs(ref_p).max():.2f})", flush=True)
    # precalc under concurrency
    for _ in range(3): eu.unet_forward(xu, t, out=ou)
    pre2 = sr.pre_calculate(y, k, 1, engine=ef)
    b1 = ef.empty(x0h.shape); b1.copy_from(x0h)
    ef._check(ef.lib.dpir_prox_fft_apply(ef.h, pre2[0].spectra.handle, b1.ptr, 7e-7, 1.0))
    eu.sync(); ef.sync()
    print(f"[{prec}] precalc+prox under concurrent UNet: max|diff| {np.abs(b1.numpy() - ref_p).max():.3e}", flush=True)
    eu.close(); ef.close()
