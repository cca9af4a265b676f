"""FFT data-fidelity operators with the reference's signatures (utils/utils_sisr.py:65-95).

    FB, FBC, F2B, FBFy = sr.pre_calculate(y, k_tensor, sf)
    x0_p = sr.data_solution(x0_p, FB, FBC, F2B, FBFy, tau, sf)

The four returned objects are views of one engine-owned spectra object (dpir_prox); they are passed
back verbatim exactly as the reference's loop does (main_ddpir.py:320, 397).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import Engine, DeviceArray, _ptr, default_engine, EngineError


class Spectra:
    """Engine-owned FB / F2B / FBFy of one batch (stored bit-reversed, see csrc/fft.hip)."""

    def __init__(self, engine: Engine, handle, B, H, W, sf):
        self.engine, self.handle, self.B, self.H, self.W, self.sf = engine, handle, B, H, W, sf

    def __del__(self):
        try:
            if self.handle and self.engine.h:
                self.engine.lib.dpir_prox_free(self.engine.h, self.handle)
        except Exception:
            pass
        self.handle = None

    def read(self, which: int) -> np.ndarray:
        planes = 3 * self.B if which == 2 else self.B
        dt = np.float32 if which == 1 else np.complex64
        out = np.empty((self.B, planes // self.B, self.H, self.W), dt)
        self.engine._check(self.engine.lib.dpir_prox_read(self.engine.h, self.handle, which, out.ctypes.data, out.nbytes))
        return out


class SpectrumView:
    def __init__(self, spectra: Spectra, which: str):
        self.spectra, self.which = spectra, which

    def numpy(self):
        s = self.spectra
        if self.which == "FB":
            return s.read(0)
        if self.which == "FBC":
            return np.conj(s.read(0))
        if self.which == "F2B":
            return s.read(1)
        return s.read(2)


def pre_calculate(x, k, sf, engine: Engine = None):
    """utils_sisr.py:78-95.  x: LR input [N,3,h,w] in [0,1]; k: [N,1,kh,kw] (device arrays)."""
    eng = engine or getattr(x, "engine", None) or default_engine(0)
    B, Cc, h, w = x.shape
    if Cc != 3:
        raise EngineError("pre_calculate: 3-channel images expected")
    kb, kc, kh, kw = k.shape
    if kb != B or kc != 1:
        raise EngineError("pre_calculate: k must be [N,1,kh,kw] with one PSF per image")
    hnd = C.c_void_p()
    eng._check(eng.lib.dpir_prox_fft_precalc(eng.h, _ptr(x), _ptr(k), kh, kw, int(sf), B, h * sf, w * sf, C.byref(hnd)))
    sp = Spectra(eng, hnd, B, h * sf, w * sf, int(sf))
    return tuple(SpectrumView(sp, n) for n in ("FB", "FBC", "F2B", "FBFy"))


def data_solution(x, FB, FBC, F2B, FBFy, alpha, sf, out=None):
    """utils_sisr.py:65-75.  alpha: python float / numpy scalar / 1-element array (the reference passes a
    [1,1,1,1] tensor)."""
    sp = FB.spectra
    if int(sf) != sp.sf:
        raise EngineError("data_solution: sf differs from pre_calculate's")
    a = float(np.asarray(alpha.numpy() if hasattr(alpha, "numpy") else alpha, dtype=np.float32).reshape(-1)[0])
    eng = sp.engine
    if out is None:
        out = eng.empty(x.shape)
    eng._check(eng.lib.dpir_data_solution(eng.h, sp.handle, _ptr(x), a, _ptr(out)))
    return out
