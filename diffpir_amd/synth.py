"""Synthetic restoration problems (no datasets or checkpoints exist offline).

Ground truth: low-pass filtered Gaussian noise fields rescaled to [0,1] (seeded, numpy).  Degradations
follow CustomDataset.__getitem__ (main_ddpir.py:46-117): wrap-around blur (scipy.ndimage.convolve,
mode='wrap', :99), antialiased bicubic x1/sf (the Resizer weights, utils_resizer.py), box / random
masks through the mask_generator mirror, then AWGN in [-1,1] space (:112-114).  Host side only.
"""
from __future__ import annotations

import numpy as np

from .utils_inpaint import mask_generator


def smooth_images(B, H, W, seed):
    from scipy import ndimage
    out = np.empty((B, 3, H, W), np.float32)
    for b in range(B):
        rng = np.random.default_rng([seed, b])
        img = rng.standard_normal((3, H, W))
        img = ndimage.gaussian_filter(img, sigma=(0.6, H / 32.0, W / 32.0), mode="wrap")
        img -= img.min()
        img /= img.max()
        out[b] = img
    return out


def gaussian_psf(size=61, std=3.0):
    """utils_deblur.py:659-664 (Blurkernel 'gaussian')."""
    from scipy import ndimage
    n = np.zeros((size, size))
    n[size // 2, size // 2] = 1
    return ndimage.gaussian_filter(n, sigma=std).astype(np.float32)


def motion_psf(size=61, seed=0):
    """A normalised random-walk line PSF (the `motionblur` package is not available offline)."""
    rng = np.random.default_rng(seed)
    k = np.zeros((size, size), np.float64)
    pos = np.array([size / 2, size / 2])
    ang = rng.uniform(0, 2 * np.pi)
    for _ in range(4 * size):
        ang += rng.normal(0, 0.25)
        pos = np.clip(pos + 0.5 * np.array([np.sin(ang), np.cos(ang)]), 2, size - 3)
        k[int(pos[0]), int(pos[1])] += 1
    return (k / k.sum()).astype(np.float32)


def bicubic_psf_x4():
    """25x25 analytic stand-in for kernels_bicubicx234.mat[0,2] (an ESTIMATED x4 bicubic PSF, centred at
    index 10.5, with negative lobes): outer product of the antialiased cubic taps.  The real file kernel
    is kept as a parity fixture in tests/golden/operators.npz['k_bic4']."""
    a = (np.arange(25, dtype=np.float64) - 10.5) / 4.0
    w = _cubic(a) / 4
    k = np.outer(w, w)
    return (k / k.sum()).astype(np.float32)


def _cubic(x):
    ax = np.abs(x)
    return ((1.5 * ax ** 3 - 2.5 * ax ** 2 + 1) * (ax <= 1) + (-0.5 * ax ** 3 + 2.5 * ax ** 2 - 4 * ax + 2) * ((1 < ax) & (ax <= 2)))


def resize_down(x, sf):
    """Antialiased cubic x1/sf along H then W with mirror boundary (host numpy, for LR synthesis)."""
    def band(n):
        m, scale = n // sf, 1.0 / sf
        kw = 4.0 / scale
        u = (np.arange(1, m + 1)) / scale + 0.5 * (1 - 1 / scale)
        left = np.floor(u - kw / 2)
        idx = left[:, None] + np.arange(int(np.ceil(kw)) + 2)[None] - 1
        wts = scale * _cubic(scale * (u[:, None] - idx - 1))
        wts /= wts.sum(1, keepdims=True)
        mir = np.concatenate([np.arange(n), np.arange(n - 1, -1, -1)])
        return wts, mir[np.mod(idx.astype(int), 2 * n)]
    wh, ih = band(x.shape[2])
    t = np.einsum("ot,bcotw->bcow", wh, x[:, :, ih, :])
    ww, iw = band(x.shape[3])
    return np.einsum("ot,bchot->bcho", ww, t[:, :, :, iw]).astype(np.float32)


def make_case(task, B, H, W, seed=42, sf=1, noise_level_img=None, blur="gaussian", ksize=61, sr_mode="blur"):
    """Returns dict(gt, y, k, mask) as numpy arrays with the loop's layouts:
    gt [B,3,H,W] in [0,1]; y [B,3,H/sf,W/sf] in [0,1] (noisy); k [B,1,kh,kw]; mask uint8 [B,3,H,W]."""
    from scipy import ndimage
    gt = smooth_images(B, H, W, seed)
    out = dict(gt=gt, k=None, mask=None)
    if noise_level_img is None:
        noise_level_img = 0.0 if task == "inpaint" else 12.75 / 255.0
    if task == "deblur":
        ks = []
        y = np.empty_like(gt)
        for b in range(B):
            if blur == "gaussian":
                np.random.seed(seed=(b * 10) % (2 ** 31))                 # main_ddpir.py:59-61
                k = gaussian_psf(ksize, 3.0 * np.abs(np.random.rand() * 2 + 1))
            else:
                k = motion_psf(ksize, seed=b)
            ks.append(k)
            for c in range(3):
                y[b, c] = ndimage.convolve(gt[b, c], k, mode="wrap")
        out["k"] = np.stack(ks)[:, None].astype(np.float32)
    elif task == "sr":
        y = resize_down(gt, sf)
        out["k"] = np.broadcast_to(bicubic_psf_x4(), (B, 1, 25, 25)).copy() if sf == 4 else None
        if sf != 4 and sr_mode == "blur":
            raise NotImplementedError("synthetic sr-blur cases are provided for sf=4")
    elif task == "inpaint":
        np.random.seed(seed)
        gen = mask_generator("box", [H // 2, H // 2 + 1], [0.5, 0.5], image_size=H, margin=(H // 16, H // 16))
        mask = np.concatenate([gen((1, 3, H, W)) for _ in range(B)], 0)
        out["mask"] = mask
        y = gt * mask
    else:
        raise ValueError(task)
    rng = np.random.default_rng([seed, 991])
    y = y * 2 - 1
    y = y + rng.normal(0, noise_level_img * 2, y.shape)                     # main_ddpir.py:112-114
    out["y"] = (y / 2 + 0.5).astype(np.float32)
    if task == "inpaint":
        out["y"] = (out["y"] * out["mask"]).astype(np.float32)                # img_L * mask (main_ddpir.py:312)
    return out
