"""Golden vectors on the inputs the reference itself SHIPS, produced by running the reference's main() END TO END (oracle/ref_exec.run_main:
parse_args_and_config -> CustomDataset.__getitem__ -> DataLoader -> create_model_and_diffusion + torch.load -> test_rho -> the sweep),
round 5:

  * `c1_*`     : BASELINE config 1 -- configs/inpaint.yaml with mask_type 'box' (mask_generator under np.random.seed(config.seed), drawn by
                 the reference's own dataset), FFHQ topology, the five testsets/demo_test PNGs as ONE batch, 20 NFE;
  * `c2lev_*`  : BASELINE config 2 as its text says -- configs/deblur.yaml with `use_DIY_kernel: false`, i.e. kernels/Levin09.mat[0, 0]
                 (19 x 19, main_ddpir.py:69-72), FFHQ topology, the five demo PNGs, 4 NFE; `c2lev20_*`: the same at 20 NFE;
  * `c3bic_*`  : BASELINE config 3's operator -- configs/sisr.yaml, kernels/kernels_bicubicx234.mat[0, 2] read by the dataset itself,
                 ImageNet-256 topology, demo image 69037.png, 12 NFE; main()'s sr sweep runs lambda = 2 .. 12 (11 passes over ONE noise
                 stream): pass 0 (lambda 2) and pass 4 (lambda 6) are stored with the number of draws each pass consumed.

Weights are synthetic (no checkpoint offline): written to model_zoo/<model_name>.pt of the scratch cwd and read back by the reference's
own torch.load / load_state_dict.  Stored per case: what DataLoader handed to test_rho (img_H u8, img_L -> `y`, k, mask) and what test_rho
produced (`out` = x_0), plus for the FFT-prox cases the reference's own fp32 rounding floor (distance to the same loop with the
closed-form prox in float64), as in fullsize.npz.

TEST INFRASTRUCTURE ONLY; build container only (needs /root/reference and /opt/conda/bin/python3.9 + h5py for Levin09.mat).
    python -m oracle.gen_golden_refdata [c1] [c2lev] [c2lev20] [c3bic]   ->  tests/golden/refdata.npz (cases given are refreshed, others kept)
"""
import glob
import os
import sys
import time

import numpy as np
import torch

from . import ref_exec, ref_import, unet_oracle as uo, diffpir_oracle as do

C3_NFE = 12          # >= 8: the flat 1e-3 dB bar applies (a 4-NFE run of this case still carries the first steps' rounding noise: its |dPSNR| is luck)
OUT = os.environ.get("DIFFPIR_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class CountingNoise:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.n = 0

    def __call__(self, like):
        self.n += 1
        return torch.randn(like.shape, generator=self.g, dtype=torch.float32)


def seeded(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda like: torch.randn(like.shape, generator=g, dtype=torch.float32)


def nchw(t):
    return np.ascontiguousarray(t.numpy().transpose(0, 3, 1, 2))


def seeded_skip(seed, skip, shape):
    """the stream of `seeded(seed)` after `skip` draws of `shape` (a later pass of the reference's sweep over one noise stream)"""
    g = torch.Generator().manual_seed(seed)
    for _ in range(skip):
        torch.randn(shape, generator=g, dtype=torch.float32)
    return lambda like: torch.randn(like.shape, generator=g, dtype=torch.float32)


def floor_of(out, tag, hp, sd, cfg, y, k, seed, ref, skip=0, shape=None, gt_u8=None):
    """reference vs the same loop with an exact (float64) prox: the yardstick of tests/gpu_common.py::fft_prox_parity (max, rms and the
    reference's own PSNR shift against exact arithmetic)."""
    nf = (lambda: seeded_skip(seed, skip, shape)) if skip else (lambda: seeded(seed))
    with torch.no_grad():
        ora = do.restore(sd, hp, cfg, torch.from_numpy(y), k=torch.from_numpy(k), noise_fn=nf()).numpy()
        exact = do.restore(sd, hp, cfg, torch.from_numpy(y), k=torch.from_numpy(k), noise_fn=nf(), exact_prox=True).numpy()
    d = ref - exact
    out.update({f"{tag}_floor_max": np.array(np.abs(d).max()), f"{tag}_floor_rms": np.array(np.sqrt(np.mean(d * d)))})
    if gt_u8 is not None:
        gt = torch.from_numpy(np.ascontiguousarray(gt_u8.transpose(0, 3, 1, 2)).astype(np.float32) / np.float32(255)) * 2 - 1
        gap = abs(do.psnr_batch(torch.from_numpy(ref) * 2 - 1, gt) - do.psnr_batch(torch.from_numpy(exact) * 2 - 1, gt))
        out[f"{tag}_floor_dpsnr"] = np.array(float(gap))
        print(f"{tag}: the reference's own |dPSNR| against exact arithmetic {float(gap):.3e} dB", flush=True)
    print(f"{tag}: reference main() vs oracle max abs diff {np.abs(ref - ora).max():.3e} | reference vs exact-prox loop: max {np.abs(d).max():.3e} "
          f"rms {np.sqrt(np.mean(d * d)):.3e}", flush=True)


def main():
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    path = os.path.join(OUT, "refdata.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    demo = {os.path.basename(f): f for f in sorted(glob.glob(os.path.join(ref_import.REF_ROOT, "testsets", "demo_test", "*.png")))}
    assert len(demo) == 5, demo
    hp_f = uo.ffhq_hp()
    sd_f = uo.synth_state_dict(hp_f, 0)

    if not only or "c1" in only:
        t0 = time.time()
        yd = ref_exec.yaml_for("inpaint", mask_type="box", iter_num=20)
        r = ref_exec.run_main(yd, sd_f, demo, noise_fn=seeded(61))
        (img_H, img_L, names, k, mask), x0 = r["batches"][0], r["x0"][0].numpy()
        out.update(c1_gt=img_H.numpy(), c1_y=nchw(img_L).astype(np.float32), c1_mask=nchw(mask).astype(np.uint8), c1_out=x0,
                   c1_seed=np.array(61), c1_nfe=np.array(20), c1_names=np.array(list(names)))
        cfg = do.LoopConfig("inpaint", 20, 0.0, 1.0, 1.0)
        with torch.no_grad():
            ora = do.restore(sd_f, hp_f, cfg, torch.from_numpy(out["c1_y"]), mask=torch.from_numpy(out["c1_mask"]).float(), noise_fn=seeded(61)).numpy()
        print(f"c1 (inpaint box, 5 demo images, 20 NFE): reference main() {time.time() - t0:.0f} s; vs oracle max abs diff {np.abs(x0 - ora).max():.3e}; "
              f"mask holes per image {[int((m == 0).sum()) // 3 for m in out['c1_mask']]}", flush=True)

    for tag, nfe, seed in (("c2lev", 4, 62), ("c2lev20", 20, 63)):
        if only and tag not in only:
            continue
        t0 = time.time()
        yd = ref_exec.yaml_for("deblur", use_DIY_kernel=False, iter_num=nfe)
        r = ref_exec.run_main(yd, sd_f, demo, noise_fn=seeded(seed))
        (img_H, img_L, names, k, mask), x0 = r["batches"][0], r["x0"][0].numpy()
        kk = k.numpy()[:, None].astype(np.float32)
        out.update({f"{tag}_out": x0, f"{tag}_seed": np.array(seed), f"{tag}_nfe": np.array(nfe)})
        out.update(c2lev_gt=img_H.numpy(), c2lev_y=nchw(img_L).astype(np.float32), c2lev_k=kk)     # same seed, same dataset draws: shared by both
        print(f"{tag}: reference main() {time.time() - t0:.0f} s, kernel {kk.shape} sum {kk[0].sum():.6f}", flush=True)
        cfg = do.LoopConfig("deblur", nfe, 12.75 / 255, 1 * 7, 0.1 * 3)                             # the sweep's values (main_ddpir.py:565-568)
        floor_of(out, tag, hp_f, sd_f, cfg, out["c2lev_y"], kk, seed, x0, gt_u8=out["c2lev_gt"])

    if "floors" in only:            # only the floors of existing records (no reference run)
        for tag, nfe, seed in (("c2lev", 4, 62), ("c2lev20", 20, 63)):
            floor_of(out, tag, hp_f, sd_f, do.LoopConfig("deblur", nfe, 12.75 / 255, 1 * 7, 0.1 * 3), out["c2lev_y"], out["c2lev_k"], seed, out[f"{tag}_out"],
                     gt_u8=out["c2lev_gt"])
        hp_i = uo.imagenet256_hp()
        sd_i = uo.synth_state_dict(hp_i, 0)
        floor_of(out, "c3bic", hp_i, sd_i, do.LoopConfig("sr", C3_NFE, 12.75 / 255, 2.0, 0.25, sf=4), out["c3bic_y"], out["c3bic_k"], 64, out["c3bic_out_pass0"],
                 gt_u8=out["c3bic_gt"])
        floor_of(out, "c3bic_p4", hp_i, sd_i, do.LoopConfig("sr", C3_NFE, 12.75 / 255, 6.0, 0.25, sf=4), out["c3bic_y"], out["c3bic_k"], 64,
                 out["c3bic_out_pass4"], skip=4 * int(out["c3bic_draws_per_pass"]), shape=(1, 3, 256, 256), gt_u8=out["c3bic_gt"])
    if not only or "c3bic" in only:
        t0 = time.time()
        hp_i = uo.imagenet256_hp()
        sd_i = uo.synth_state_dict(hp_i, 0)
        noise = CountingNoise(64)
        yd = ref_exec.yaml_for("sr", model_name="256x256_diffusion_uncond", iter_num=C3_NFE)
        r = ref_exec.run_main(yd, sd_i, {"69037.png": demo["69037.png"]}, noise_fn=noise)
        assert len(r["x0"]) == 11 and noise.n % 11 == 0, (len(r["x0"]), noise.n)
        img_H, img_L, names, k, mask = r["batches"][0]
        kk = k.numpy()[:, None].astype(np.float32)
        out.update(c3bic_gt=img_H.numpy(), c3bic_y=nchw(img_L).astype(np.float32), c3bic_k=kk, c3bic_seed=np.array(64), c3bic_nfe=np.array(C3_NFE),
                   c3bic_draws_per_pass=np.array(noise.n // 11), c3bic_out_pass0=r["x0"][0].numpy(), c3bic_out_pass4=r["x0"][4].numpy(),
                   c3bic_lambdas=np.array([1 * i for i in range(2, 13)], np.float64))
        print(f"c3bic: reference main() (11-pass lambda sweep) {time.time() - t0:.0f} s, {noise.n // 11} draws per pass, y {out['c3bic_y'].shape}", flush=True)
        cfg = do.LoopConfig("sr", C3_NFE, 12.75 / 255, 2.0, 0.25, sf=4)
        floor_of(out, "c3bic", hp_i, sd_i, cfg, out["c3bic_y"], kk, 64, out["c3bic_out_pass0"], gt_u8=out["c3bic_gt"])
        floor_of(out, "c3bic_p4", hp_i, sd_i, do.LoopConfig("sr", C3_NFE, 12.75 / 255, 6.0, 0.25, sf=4), out["c3bic_y"], kk, 64,
                 out["c3bic_out_pass4"], skip=4 * (noise.n // 11), shape=(1, 3, 256, 256), gt_u8=out["c3bic_gt"])
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), flush=True)


if __name__ == "__main__":
    main()
