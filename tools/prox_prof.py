"""Workload for rocprofv3 passes over the FFT prox: PROX_MODE (launches | fused | wave), PROX_B, PROX_SF, PROX_N applies of dpir_prox_fft_apply."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import utils_sisr as sr
B, sf, n = int(os.environ.get("PROX_B", "16")), int(os.environ.get("PROX_SF", "1")), int(os.environ.get("PROX_N", "50"))
H = int(os.environ.get("PROX_H", "256"))
eng = diffpir_amd.Engine(0)
eng.set_prox_launch(os.environ.get("PROX_MODE", "wave"))
rng = np.random.default_rng(0)
y = eng.to_device(rng.random((B, 3, H // sf, H // sf)).astype(np.float32))
kk = rng.random((B, 1, 25, 25)).astype(np.float32); kk /= kk.sum(axis=(2, 3), keepdims=True)
pre = sr.pre_calculate(y, eng.to_device(kk), sf)
x0 = eng.to_device(rng.random((B, 3, H, H)).astype(np.float32) * 2 - 1)
for _ in range(n):
    eng._check(eng.lib.dpir_prox_fft_apply(eng.h, pre[0].spectra.handle, x0.ptr, 0.05, 1.0))
eng.sync()
print("done", B, sf, n)
