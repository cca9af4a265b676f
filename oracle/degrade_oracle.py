"""Oracle: degradation synthesis and metrics (the steps either side of the loop), numpy / scipy / torch-CPU.

TEST INFRASTRUCTURE ONLY.  Follows main_ddpir.py:84-114 (CustomDataset.__getitem__: scipy.ndimage.convolve(mode='wrap') on the
uint8 image, utils_image.imresize_np, img_H * mask / 255, AWGN in [-1,1] space in float64) and main_ddpir.py:482-517 with
utils/utils_image.py:601-610 (calculate_psnr_batch) and :470-490 (rgb2ycbcr_batch, only_y).  scipy is the reference's own
dependency for the blur (requirements.txt), so it is called, not restated.
"""
import numpy as np
import torch
from scipy import ndimage

from . import diffpir_oracle as do


def degrade(task, gt_u8, k=None, mask=None, noise_level_img=0.0, sf=1, noise=None):
    """gt_u8 [B,H,W,3] uint8; k [B,1,kh,kw]; mask uint8 [B,3,H,W]; noise: standard normal [B,3,h,w] (float32) or None.
    Returns y [B,3,h,w] float32 (NCHW, like util.single2tensor4_batch(img_L))."""
    B = gt_u8.shape[0]
    ys = []
    for b in range(B):
        img_H = gt_u8[b]
        if task == "deblur":
            img_L = ndimage.convolve(img_H, np.expand_dims(k[b, 0], axis=2), mode="wrap")       # main_ddpir.py:99 (uint8 in -> uint8 out)
            img_L = np.float32(img_L / 255.)                                                       # util.uint2single
        elif task == "sr":
            x = torch.from_numpy(np.float32(img_H / 255.).transpose(2, 0, 1))[None]
            img_L = do.resizer_apply(x, 1 / sf)[0].numpy().transpose(1, 2, 0)                     # == utils_image.imresize_np (2e-7)
        else:
            img_L = img_H * mask[b].transpose(1, 2, 0) / 255.                                      # main_ddpir.py:108
        img_L = img_L * 2 - 1
        # np.random.normal returns float64; `img_L += ...` (main_ddpir.py:113) is IN PLACE, so img_L keeps its dtype: float32 for
        # deblur / sr (uint2single, imresize_np), float64 for inpainting (uint8 * mask / 255.)
        nz = np.zeros(img_L.shape, np.float64) if (noise is None or noise_level_img == 0) else noise[b].transpose(1, 2, 0).astype(np.float64) * (noise_level_img * 2)
        img_L += nz
        img_L = img_L / 2 + 0.5
        y = np.float32(img_L).transpose(2, 0, 1)
        if task == "inpaint":
            y = y * mask[b].astype(np.float32)                                                    # main_ddpir.py:311-313
        ys.append(y)
    return np.stack(ys).astype(np.float32)


def metrics(x0, gt_u8):
    """x0 [B,3,H,W] float32 in [0,1]; returns per-image (psnr, psnr_y) float32 arrays: the terms whose mean
    calculate_psnr_batch returns."""
    a = torch.from_numpy(x0) * 2 - 1
    b = torch.from_numpy(gt_u8.transpose(0, 3, 1, 2)) / 255 * 2 - 1

    def per_image(u, v):
        mse = torch.mean((u - v) ** 2, axis=(1, 2, 3))
        return torch.where(mse == 0, torch.full_like(mse, float("inf")), 20 * torch.log10(2.0 / torch.sqrt(mse + 1e-10))).numpy()

    def y_only(t):
        r = torch.zeros_like(t)
        r[:, 0] = 0.299 * t[:, 0] + 0.587 * t[:, 1] + 0.114 * t[:, 2]
        return r
    return per_image(a, b), per_image(y_only(a), y_only(b))
