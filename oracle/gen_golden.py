"""Generate tests/golden/*.npz from the LIVE reference (run in the build container only):

    python -m oracle.gen_golden

TEST INFRASTRUCTURE ONLY.  Every array written here is an OUTPUT OF THE REFERENCE'S OWN
CODE (imported from /root/reference via oracle/ref_import.py) on seeded inputs; the inputs
are stored next to the outputs so the fixtures are self-contained on the GPU box, where
/root/reference does not exist.  The oracle restatement (oracle/*.py) and the HIP engine
are both checked against these files.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import ref_import, ref_exec
from . import unet_oracle as uo
from . import diffpir_oracle as do

OUT = os.environ.get("DIFFPIR_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def seeded_noise_fn(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda like: torch.randn(like.shape, generator=g, dtype=torch.float32)


def smooth_images(B, H, W, seed):
    """Synthetic GT in [0,1]: low-pass filtered noise (no dataset offline)."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((B, 3, H, W))
    img = ndimage.gaussian_filter(img, sigma=(0, 0.6, H / 32.0, W / 32.0), mode="wrap")
    img -= img.min(axis=(1, 2, 3), keepdims=True)
    img /= img.max(axis=(1, 2, 3), keepdims=True)
    return img.astype(np.float32)


def gaussian_kernel(size, std):
    """utils_deblur.py:659-664 (Blurkernel 'gaussian'): gaussian_filter of a centred delta."""
    from scipy import ndimage
    n = np.zeros((size, size))
    n[size // 2, size // 2] = 1
    return ndimage.gaussian_filter(n, sigma=std).astype(np.float32)


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = ref_import.load()
    torch.set_num_threads(8)

    # ---------------------------------------------------------------- 1. schedule tables
    sched = {}
    for name, cfg in dict(
            deblur100=do.LoopConfig("deblur", 100, 12.75 / 255, 7.0, 0.3),
            inpaint20=do.LoopConfig("inpaint", 20, 0.0, 1.0, 1.0),
            sr100=do.LoopConfig("sr", 100, 12.75 / 255, 6.0, 0.25, sf=4)).items():
        # the live tables, computed exactly like main_ddpir.py:184-190, 274-286, 327-344
        betas = torch.from_numpy(np.linspace(cfg.beta_start, cfg.beta_end, cfg.T, dtype=np.float32))
        ac = np.cumprod((1.0 - betas).cpu(), axis=0)
        s_ac, s_1m = torch.sqrt(ac), torch.sqrt(1. - ac)
        red = torch.div(s_1m, s_ac)
        sigmas = torch.tensor([red[cfg.T - 1 - i] for i in range(cfg.T)])
        rhos = torch.tensor([cfg.lambda_ * (cfg.sigma ** 2) / ((s_1m[i] / s_ac[i]) ** 2) for i in range(cfg.T)])
        seq = np.sqrt(np.linspace(0, cfg.T ** 2, cfg.iter_num))
        seq = [int(s) for s in list(seq)]
        seq[-1] -= 1
        t_is = [int(ns.utils_model.find_nearest(red, sigmas[s].cpu().numpy())) for s in seq]
        sched[name + "_t"] = np.array(t_is, dtype=np.int64)
        sched[name + "_tau"] = np.array([float(rhos[t].float()) for t in t_is], dtype=np.float32)
    # the same tables out of the reference's OWN statements (oracle/ref_exec.py: test_rho executed with a recording stand-in for the
    # network and for data_solution): every t_i it visits and every tau it hands to the prox
    for name, task, over in (("deblur100", "deblur", dict(iter_num=100)), ("sr100", "sr", dict(iter_num=100, zeta=0.25))):
        ts, taus = ref_exec.reference_step_trace(ref_exec.yaml_for(task, **over), (1, 3, 8, 8), sweep=True)
        n = len(sched[name + "_t"])
        i = 4 if task == "sr" else 0            # the sr sweep runs lambda = 2 .. 12 (main_ddpir.py:556): lambda 6 is its fifth pass
        assert np.array_equal(ts[i * n:(i + 1) * n], sched[name + "_t"]) and np.array_equal(taus[i * (n - 1):(i + 1) * (n - 1)], sched[name + "_tau"][:n - 1]), name
        print("schedule", name, ": executed test_rho (incl. the reference's lambda/zeta sweep) visits the same t_i / tau")
    _, diffusion = ref_exec.build_unet(uo.tiny_hp(), uo.synth_state_dict(uo.tiny_hp(), 0))
    sched["sqrt_recip_ac"] = diffusion.sqrt_recip_alphas_cumprod
    sched["sqrt_recipm1_ac"] = diffusion.sqrt_recipm1_alphas_cumprod
    sched["drv_sqrt_ac"] = s_ac.numpy()
    sched["drv_sqrt_1m_ac"] = s_1m.numpy()
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **sched)

    # ---------------------------------------------------------------- 2. UNet forward
    for tag, hp, hw, B, cls in (("tiny", uo.tiny_hp(), 32, 2, None),
                                ("tinycc", uo.tiny_hp(class_cond=True), 32, 2, [3, 7]),
                                ("ffhq", uo.ffhq_hp(), 32, 1, None)):
        sd = uo.synth_state_dict(hp, 0)
        model, _ = ref_exec.build_unet(hp, sd)
        g = torch.Generator().manual_seed(11)
        x = torch.randn((B, 3, hw, hw), generator=g)
        t = torch.tensor([999, 37][:B])
        yl = None if cls is None else torch.tensor(cls)
        with torch.no_grad():
            out = model(x, t) if yl is None else model(x, t, yl)
        d = dict(x=x.numpy(), t=t.numpy(), out=out.numpy())
        if yl is not None:
            d["y"] = yl.numpy()
        np.savez_compressed(os.path.join(OUT, f"unet_{tag}.npz"), **d)
        print("unet", tag, float(out.std()))

    # ---------------------------------------------------------------- 3. operators
    import scipy.io
    kb = scipy.io.loadmat(os.path.join(ref_import.REF_ROOT, "kernels", "kernels_bicubicx234.mat"))["kernels"]
    k_bic4 = kb[0, 2].astype(np.float32)                                # main_ddpir.py:54-56
    gt = smooth_images(2, 64, 64, 5)
    ops = dict(k_bic4=k_bic4)
    # deblur sf=1, 15x15 gaussian
    kg = gaussian_kernel(15, 2.0)
    from scipy import ndimage
    yb = np.stack([ndimage.convolve(gt[b].transpose(1, 2, 0), kg[:, :, None], mode="wrap").transpose(2, 0, 1) for b in range(2)])
    yb = (yb + np.random.default_rng(1).normal(0, 0.05, yb.shape)).astype(np.float32)
    kt = torch.from_numpy(np.stack([kg, kg]))[:, None]
    FB, FBC, F2B, FBFy = ns.utils_sisr.pre_calculate(torch.from_numpy(yb), kt, 1)
    z = torch.from_numpy(smooth_images(2, 64, 64, 6))
    for a in (1e-5, 0.02, 3.0):
        ops[f"deblur_out_{a}"] = ns.utils_sisr.data_solution(z, FB, FBC, F2B, FBFy, torch.tensor(a).float().repeat(1, 1, 1, 1), 1).numpy()
    ops.update(deblur_k=kt.numpy(), deblur_y=yb, deblur_z=z.numpy(), deblur_FB=FB.numpy(), deblur_F2B=F2B.numpy(), deblur_FBFy=FBFy.numpy())
    # sr sf=4 with the bicubic PSF
    ylr = ns.utils_resizer.Resizer((2, 3, 64, 64), 0.25)(torch.from_numpy(gt)).numpy().astype(np.float32)
    k4 = torch.from_numpy(np.stack([k_bic4, k_bic4]))[:, None]
    FB, FBC, F2B, FBFy = ns.utils_sisr.pre_calculate(torch.from_numpy(ylr), k4, 4)
    for a in (1e-4, 0.05, 2.0):
        ops[f"sr4_out_{a}"] = ns.utils_sisr.data_solution(z, FB, FBC, F2B, FBFy, torch.tensor(a).float().repeat(1, 1, 1, 1), 4).numpy()
    ops.update(sr4_y=ylr, sr4_FB=FB.numpy(), sr4_F2B=F2B.numpy(), sr4_FBFy=FBFy.numpy())
    # sf=2 with a non-symmetric kernel (catches transposes)
    rng = np.random.default_rng(3)
    kr = rng.random((7, 9)).astype(np.float32)
    kr /= kr.sum()
    y2 = rng.random((2, 3, 32, 32)).astype(np.float32)
    k2 = torch.from_numpy(np.stack([kr, kr[::-1].copy()]))[:, None]
    FB, FBC, F2B, FBFy = ns.utils_sisr.pre_calculate(torch.from_numpy(y2), k2, 2)
    ops["sf2_out"] = ns.utils_sisr.data_solution(z, FB, FBC, F2B, FBFy, torch.tensor(0.1).float().repeat(1, 1, 1, 1), 2).numpy()
    ops.update(sf2_k=k2.numpy(), sf2_y=y2, sf2_FB=FB.numpy(), sf2_FBFy=FBFy.numpy())
    # Resizer down x4 and torch bicubic up (init, main_ddpir.py:295)
    ops["resizer_in"] = gt
    ops["resizer_out"] = ylr
    ops["bicubic_up"] = torch.nn.functional.interpolate(torch.from_numpy(ylr), size=(64, 64), mode="bicubic", align_corners=False).numpy()
    # box / random masks under the driver's seed (utils_inpaint.py:86-137, main_ddpir.py:106-109, 166)
    np.random.seed(42)
    mg = ns.utils_inpaint.mask_generator("box", [128, 129], [0.5, 0.5])
    ops["mask_box"] = mg(torch.zeros(1, 3, 256, 256)).numpy().astype(np.uint8)
    mg = ns.utils_inpaint.mask_generator("random", [128, 129], [0.5, 0.5])
    ops["mask_random"] = mg(torch.zeros(1, 3, 256, 256)).numpy().astype(np.uint8)
    # tensor2uint_batch + psnr
    xo = torch.from_numpy(rng.random((2, 3, 16, 16)).astype(np.float32) * 1.2 - 0.1)
    ops["u8_in"] = xo.numpy()
    ops["u8_out"] = ns.utils_image.tensor2uint_batch(xo.clone())
    ops["psnr"] = np.float64(ns.utils_image.calculate_psnr_batch(xo * 2 - 1, torch.from_numpy(gt[:, :, :16, :16]) * 2 - 1))
    np.savez_compressed(os.path.join(OUT, "operators.npz"), **ops)

    # ---------------------------------------------------------------- 4. whole loop, tiny UNet
    hp = uo.tiny_hp()
    sd = uo.synth_state_dict(hp, 0)
    model, diffusion = ref_exec.build_unet(hp, sd)
    gt = smooth_images(2, 32, 32, 9)
    loops = dict(gt=gt)
    # deblur
    kg = gaussian_kernel(9, 1.2)
    yb = np.stack([ndimage.convolve(gt[b].transpose(1, 2, 0), kg[:, :, None], mode="wrap").transpose(2, 0, 1) for b in range(2)])
    yb = (yb + np.random.default_rng(2).normal(0, 0.05, yb.shape)).astype(np.float32)
    kt = torch.from_numpy(np.stack([kg, kg]))[:, None]
    cfg = do.LoopConfig("deblur", 6, 12.75 / 255, 7.0, 0.3)
    loops["deblur_y"], loops["deblur_k"] = yb, kt.numpy()
    loops["deblur_out"] = ref_exec.restore_ref(model, diffusion, cfg, torch.from_numpy(yb), k=kt, noise_fn=seeded_noise_fn(42)).numpy()
    # deblur with eta != 0 (exercises the n1 term)
    cfg = do.LoopConfig("deblur", 5, 12.75 / 255, 7.0, 0.3, eta=0.7)
    loops["deblur_eta_out"] = ref_exec.restore_ref(model, diffusion, cfg, torch.from_numpy(yb), k=kt, noise_fn=seeded_noise_fn(43)).numpy()
    # inpaint (box mask scaled to 32x32)
    m = np.ones((2, 3, 32, 32), np.float32)
    m[0, :, 8:24, 6:22] = 0
    m[1, :, 4:20, 12:28] = 0
    yi = (gt * m).astype(np.float32)
    cfg = do.LoopConfig("inpaint", 6, 0.0, 1.0, 1.0)
    loops["inpaint_y"], loops["inpaint_mask"] = yi, m.astype(np.uint8)
    loops["inpaint_out"] = ref_exec.restore_ref(model, diffusion, cfg, torch.from_numpy(yi), mask=torch.from_numpy(m), noise_fn=seeded_noise_fn(44)).numpy()
    # sr x4, blur mode (bicubic PSF) and cubic mode (IBP)
    gt64 = smooth_images(2, 64, 64, 10)
    ylr = ns.utils_resizer.Resizer((2, 3, 64, 64), 0.25)(torch.from_numpy(gt64))
    ylr = (ylr + torch.from_numpy(np.random.default_rng(4).normal(0, 0.05, ylr.shape).astype(np.float32))).float()
    k4 = torch.from_numpy(np.stack([k_bic4, k_bic4]))[:, None]
    loops["sr_gt"], loops["sr_y"] = gt64, ylr.numpy()
    cfg = do.LoopConfig("sr", 5, 12.75 / 255, 6.0, 0.25, sf=4)
    loops["sr_blur_out"] = ref_exec.restore_ref(model, diffusion, cfg, ylr, k=k4, noise_fn=seeded_noise_fn(45)).numpy()
    cfg = do.LoopConfig("sr", 5, 12.75 / 255, 6.0, 0.25, sf=4, sr_mode="cubic", inIter=2, gamma=0.5)
    loops["sr_cubic_out"] = ref_exec.restore_ref(model, diffusion, cfg, ylr, k=k4, noise_fn=seeded_noise_fn(46)).numpy()
    np.savez_compressed(os.path.join(OUT, "loops.npz"), **loops)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
