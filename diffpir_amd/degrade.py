"""Degradation synthesis and metrics on the device: the steps either side of the restoration loop.

`degrade` mirrors CustomDataset.__getitem__'s arithmetic (main_ddpir.py:84-114) for a batch of uint8 ground-truth images;
`metrics` mirrors main_ddpir.py:482-517 (PSNR and PSNR on the Y channel).  Both are thin wrappers over dpir_degrade /
dpir_metrics; PSF and mask GENERATION (a few hundred scalar operations per image under the numpy RNG) stays on the host so that
kernels and masks remain bit-identical to the reference's.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .engine import Engine, _ptr
from .restore import TASKS


def engine_task(task: str, sr_mode: str = "blur") -> int:
    if task == "sr":
        return TASKS["sr_blur"] if sr_mode == "blur" else TASKS["sr_cubic"]
    return TASKS[task]


def degrade(engine: Engine, task: str, gt_u8, k=None, mask=None, *, noise_level_img: float, sf: int = 1, sr_mode: str = "blur",
            seed: int = 0, image_offset: int = 0, noise=None, out=None):
    """gt_u8: uint8 [B,H,W,3] (numpy or device); k: [B,1,kh,kw] float32; mask: uint8 [B,3,H,W]; noise: optional host-fed
    standard-normal [B,3,H/sf,W/sf] (parity runs).  Returns (y DeviceArray [B,3,H/sf,W/sf] in [0,1], device copies of gt/k/mask)."""
    keep = {}

    def dev(a, dt, name):
        if a is None:
            return None
        if isinstance(a, np.ndarray):
            a = engine.to_device(np.ascontiguousarray(a, dtype=dt))
        keep[name] = a
        return a
    gt_d, k_d, m_d, n_d = dev(gt_u8, np.uint8, "gt"), dev(k, np.float32, "k"), dev(mask, np.uint8, "mask"), dev(noise, np.float32, "noise")
    B, H, W, _ = gt_d.shape
    d = _lib.DegradeDesc()
    d.task, d.B, d.H, d.W, d.sf = engine_task(task, sr_mode), B, H, W, sf
    if k_d is not None:
        d.kh, d.kw = k_d.shape[2], k_d.shape[3]
    d.noise_level_img, d.seed, d.image_offset = float(noise_level_img), seed, image_offset
    if out is None:
        out = engine.empty((B, 3, H // sf, W // sf))
    engine._check(engine.lib.dpir_degrade(engine.h, C.byref(d), _ptr(gt_d), _ptr(k_d), _ptr(m_d), _ptr(n_d), _ptr(out)))
    return out, keep


def metrics(engine: Engine, x0, gt_u8_dev):
    """Per-image (PSNR, PSNR-Y) in dB as float32 arrays; x0: DeviceArray [B,3,H,W] in [0,1]."""
    B, _, H, W = x0.shape
    p = np.empty(B, np.float32)
    py = np.empty(B, np.float32)
    engine._check(engine.lib.dpir_metrics(engine.h, _ptr(x0), _ptr(gt_u8_dev), B, H, W, p.ctypes.data, py.ctypes.data))
    return p, py
