"""Timing of the FFT data-fidelity prox (dpir_prox_fft_apply) against its algorithmic HBM bytes (GPU box only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import synth, utils_sisr as sr

eng = diffpir_amd.Engine(0)
for (B, H, sf) in ((16, 256, 1), (64, 256, 1), (16, 256, 4), (32, 256, 4), (32, 256, 2), (8, 512, 4)):
    case = synth.make_case("deblur", B, H, H, seed=1, ksize=25) if sf == 1 else None
    rng = np.random.default_rng(0)
    if sf == 1:
        y, k = eng.to_device(case["y"]), eng.to_device(case["k"])
    else:
        y = eng.to_device(rng.random((B, 3, H // sf, H // sf)).astype(np.float32))
        kk = rng.random((B, 1, 25, 25)).astype(np.float32); kk /= kk.sum(axis=(2, 3), keepdims=True)
        k = eng.to_device(kk)
    pre = sr.pre_calculate(y, k, sf)
    x0 = eng.to_device(rng.random((B, 3, H, H)).astype(np.float32) * 2 - 1)
    h = pre[0].spectra.handle
    for _ in range(3):
        eng._check(eng.lib.dpir_prox_fft_apply(eng.h, h, x0.ptr, 0.05, 1.0))
    eng.sync()
    n = 50
    eng.prof_enable(True); eng.prof_reset()
    for _ in range(n):
        eng._check(eng.lib.dpir_prox_fft_apply(eng.h, h, x0.ptr, 0.05, 1.0))
    eng.sync()
    ms, cnt = eng.prof_read()["fft_prox"]
    eng.prof_enable(False)
    us = ms / n * 1e3
    half = H // 2 + 1
    alg = B * (3 * H * H * 4 * 2 + 3 * H * half * 8 + H * half * 4 + (H * half * 8 if sf > 1 else 0))
    print(f"prox B={B} {H}x{H} sf={sf}: {us:8.1f} us/apply  algorithmic {alg/1e6:7.2f} MB -> {alg/us/1e6:6.3f} TB/s = {alg/us/1e6/8.0*100:5.1f}% of 8 TB/s")
