"""Golden vectors for degradation synthesis and metrics from the LIVE reference's own functions (utils_image.imresize_np,
calculate_psnr_batch, rgb2ycbcr_batch) and from scipy.ndimage.convolve, the reference's blur (main_ddpir.py:99).
TEST INFRASTRUCTURE ONLY; build container only.   python -m oracle.gen_golden_degrade  ->  tests/golden/degrade.npz"""
import os
import numpy as np
import torch
from scipy import ndimage

from . import ref_import, degrade_oracle as dg

OUT = os.environ.get("DIFFPIR_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    ns = ref_import.load()
    ui = ns.utils_image
    rng = np.random.default_rng(3)
    B, H, W = 2, 64, 64
    base = ndimage.gaussian_filter(rng.random((B, H, W, 3)), sigma=(0, 2, 2, 0))
    gt = np.clip((base - base.min()) / (base.max() - base.min()) * 255, 0, 255).round().astype(np.uint8)
    gt[0, :8, :8] = 200                                   # a flat patch: sum(k) < 1 makes the uint8 cast land on 199
    out = {"gt": gt}
    # deblur: main_ddpir.py:99 verbatim, two different Gaussian PSFs (utils_deblur.py:659-664)
    ks = []
    for b in range(B):
        n = np.zeros((15, 15)); n[7, 7] = 1
        ks.append(ndimage.gaussian_filter(n, sigma=1.5 + b).astype(np.float32))
    k = np.stack(ks)[:, None]
    blur = np.stack([np.float32(ndimage.convolve(gt[b], np.expand_dims(k[b, 0], axis=2), mode="wrap") / 255.) for b in range(B)])
    noise = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    sig = 12.75 / 255
    # main_ddpir.py:112-114 verbatim (np.random.normal(0, s, shape) == s * standard normal, float64; `+=` keeps img_L float32)
    img_L = blur.copy()
    img_L = img_L * 2 - 1
    img_L += noise.transpose(0, 2, 3, 1).astype(np.float64) * (sig * 2)
    img_L = img_L / 2 + 0.5
    assert img_L.dtype == np.float32
    z = blur.copy() * 2 - 1
    z += np.zeros(z.shape, np.float64)
    z = z / 2 + 0.5                                       # the same three lines at noise level 0 (not an identity in float32)
    out.update(k=k, noise=noise, deblur_y=img_L.transpose(0, 3, 1, 2), deblur_y_clean=blur.transpose(0, 3, 1, 2),
               deblur_y_sigma0=z.transpose(0, 3, 1, 2))
    assert np.abs(dg.degrade("deblur", gt, k=k, noise_level_img=sig, noise=noise) - out["deblur_y"]).max() == 0
    # sr: utils_image.imresize_np(util.uint2single(img_H), 1/sf)
    sr = np.stack([ui.imresize_np(np.float32(gt[b] / 255.), 1 / 4) for b in range(B)]).transpose(0, 3, 1, 2).astype(np.float32)
    out["sr4_y_clean"] = (sr * 2 - 1) / 2 + 0.5          # float32 throughout, noise level 0
    print("imresize_np vs oracle:", np.abs(dg.degrade("sr", gt, sf=4) - sr).max())
    # inpaint
    mask = (rng.random((B, 1, H, W)) > 0.5).astype(np.uint8).repeat(3, 1)
    out.update(mask=mask, inpaint_y=dg.degrade("inpaint", gt, mask=mask))
    # metrics of the reference's own functions on a perturbed image
    x0 = np.clip(gt.transpose(0, 3, 1, 2) / 255. + 0.05 * rng.standard_normal((B, 3, H, W)), -0.1, 1.1).astype(np.float32)
    a = torch.from_numpy(x0) * 2 - 1
    b = torch.from_numpy(gt.transpose(0, 3, 1, 2)) / 255 * 2 - 1
    out.update(x0=x0, psnr=np.array([ui.calculate_psnr_batch(a[i:i + 1], b[i:i + 1]) for i in range(B)], np.float32),
               psnr_y=np.array([ui.calculate_psnr_batch(ui.rgb2ycbcr_batch(a[i:i + 1], only_y=True), ui.rgb2ycbcr_batch(b[i:i + 1], only_y=True))
                                for i in range(B)], np.float32),
               psnr_batch=np.float32(ui.calculate_psnr_batch(a, b)))
    p, py = dg.metrics(x0, gt)
    print("metrics oracle vs live:", np.abs(p - out["psnr"]).max(), np.abs(py - out["psnr_y"]).max())
    np.savez_compressed(os.path.join(OUT, "degrade.npz"), **out)


if __name__ == "__main__":
    main()
