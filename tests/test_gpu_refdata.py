"""-m gpu: the engine on the inputs the reference itself ships -- tests/golden/refdata.npz, produced by running the reference's main() end
to end (oracle/gen_golden_refdata.py -> oracle/ref_exec.run_main): the five testsets/demo_test PNGs, kernels/Levin09.mat[0, 0],
kernels/kernels_bicubicx234.mat[0, 2], mask_generator('box') drawn by the reference's own dataset, the sweep's lambda / zeta.
What DataLoader handed to test_rho (`*_y`, `*_k`, `*_mask`) goes into restore_batch; what test_rho produced (`*_out`) is the target."""
import os

import numpy as np
import pytest
import torch

import diffpir_amd
from diffpir_amd import restore
from oracle import unet_oracle as uo
from tests.gpu_common import make_model, seeded_noise_fn_np, fft_prox_parity

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdata.npz")


@pytest.fixture(scope="module")
def ref():
    return np.load(GOLDEN)


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def ffhq(request):
    e = diffpir_amd.Engine(0)
    e.set_precision(request.param)
    make_model(e, uo.ffhq_hp())
    yield e, request.param
    e.close()


def _gt01(g_u8):
    return np.ascontiguousarray(g_u8.transpose(0, 3, 1, 2)).astype(np.float32) / np.float32(255)


@pytest.mark.parametrize("graph", [False, True])
def test_c1_box_inpainting_of_the_demo_images_matches_the_reference_run(ffhq, ref, graph):
    """BASELINE config 1 (configs/inpaint.yaml, mask_type box, 20 NFE), the five demo PNGs as one batch, masks drawn by the reference's
    dataset.  The masked pixels' logic is integer: the mask must be binary and identical on the three channels."""
    e, precision = ffhq
    m = ref["c1_mask"]
    assert m.dtype == np.uint8 and set(np.unique(m)) == {0, 1} and (m[:, 0] == m[:, 1]).all() and (m[:, 0] == m[:, 2]).all()
    assert all(int((mm[0] == 0).sum()) == 128 * 128 for mm in m)                      # mask_len_range [128, 129): utils_inpaint.py:102-106
    cfg = restore.LoopConfig(task="inpaint", iter_num=int(ref["c1_nfe"]), lambda_=1.0, zeta=1.0, noise_level_img=0.0)
    out = restore.restore_batch(e, cfg, ref["c1_y"], mask=m, noise_source="host", noise_fn=seeded_noise_fn_np(int(ref["c1_seed"])), use_graph=graph).numpy()
    gt = _gt01(ref["c1_gt"])
    d = out - ref["c1_out"]
    gap = abs(restore.psnr_batch(out * 2 - 1, gt * 2 - 1) - restore.psnr_batch(ref["c1_out"] * 2 - 1, gt * 2 - 1))
    print(f"C1 box inpainting, 5 demo images, 20 NFE [{precision}, graph={graph}] vs the reference's main(): max {np.abs(d).max():.3e} rms {np.sqrt(np.mean(d * d)):.3e} "
          f"|dPSNR| {gap:.2e} dB")
    assert gap <= 1e-3 and np.abs(d).max() < 2e-3


@pytest.mark.parametrize("tag", ["c2lev", "c2lev20"])
def test_c2_levin09_deblurring_of_the_demo_images_matches_the_reference_run(ffhq, ref, tag):
    """BASELINE config 2 as worded ("Levin09 kernel"): configs/deblur.yaml with use_DIY_kernel false -> kernels/Levin09.mat[0, 0] (19 x 19),
    lambda 1 x 7, zeta 0.1 x 3 from the reference's sweep; 4 NFE against the conditioning-aware gate, 20 NFE against the flat 1e-3 dB bar."""
    e, precision = ffhq
    nfe = int(ref[f"{tag}_nfe"])
    assert ref["c2lev_k"].shape == (5, 1, 19, 19) and abs(float(ref["c2lev_k"][0].sum()) - 1.0) < 1e-5
    cfg = restore.LoopConfig(task="deblur", iter_num=nfe, lambda_=1 * 7, zeta=0.1 * 3)
    out = restore.restore_batch(e, cfg, ref["c2lev_y"], k=ref["c2lev_k"], noise_source="host", noise_fn=seeded_noise_fn_np(int(ref[f"{tag}_seed"])),
                                use_graph=True).numpy()
    fft_prox_parity(out, ref[f"{tag}_out"], _gt01(ref["c2lev_gt"]), f"C2 Levin09[0] on the 5 demo images, {nfe} NFE [{precision}] vs the reference's main()",
                    floor=(float(ref[f"{tag}_floor_max"]), float(ref[f"{tag}_floor_rms"])), nfe=nfe, floor_dpsnr=float(ref[f"{tag}_floor_dpsnr"]))


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_c3_bicubic_psf_sisr_of_a_demo_image_matches_the_reference_sweep(ref, precision):
    """BASELINE config 3's operator on a demo image: kernels_bicubicx234.mat[0, 2] read by the reference's dataset, ImageNet-256 topology,
    12 NFE; main()'s sr sweep makes 11 passes (lambda 2 .. 12) over ONE noise stream -- pass 0 (lambda 2) and pass 4 (lambda 6) are compared,
    the latter after skipping the draws the first four passes consumed."""
    e = diffpir_amd.Engine(0)
    e.set_precision(precision)
    make_model(e, uo.imagenet256_hp())
    y, k, gt = ref["c3bic_y"], ref["c3bic_k"], _gt01(ref["c3bic_gt"])
    assert k.shape == (1, 1, 25, 25) and y.shape == (1, 3, 64, 64)
    per_pass, lambdas = int(ref["c3bic_draws_per_pass"]), ref["c3bic_lambdas"]
    for ps in (0, 4):
        g = torch.Generator().manual_seed(int(ref["c3bic_seed"]))
        for _ in range(ps * per_pass):
            torch.randn((1, 3, 256, 256), generator=g)
        nf = lambda shape: torch.randn(tuple(shape), generator=g, dtype=torch.float32).numpy()
        cfg = restore.LoopConfig(task="sr", iter_num=int(ref["c3bic_nfe"]), lambda_=float(lambdas[ps]), zeta=0.25, sf=4)
        out = restore.restore_batch(e, cfg, y, k=k, noise_source="host", noise_fn=nf, use_graph=True).numpy()
        tgt = ref[f"c3bic_out_pass{ps}"]
        ftag = "c3bic" if ps == 0 else "c3bic_p4"
        fft_prox_parity(out, tgt, gt, f"C3 bicubic PSF on 69037.png, {int(ref['c3bic_nfe'])} NFE, sweep pass {ps} (lambda {lambdas[ps]:g}) [{precision}] vs the reference's main()",
                        floor=(float(ref[f"{ftag}_floor_max"]), float(ref[f"{ftag}_floor_rms"])), floor_dpsnr=float(ref[f"{ftag}_floor_dpsnr"]), nfe=int(ref["c3bic_nfe"]))
    e.close()


def test_f16x1_reduced_precision_quality_on_the_demo_images(ref):
    """SURVEY 8f-2, round-5 review item 6: the opt-in reduced-precision mode (f16 operands, ONE MFMA per product -- the reference's own use_fp16 recipe,
    guided_diffusion/fp16_util.py:15-32, unet.py:618-632) on the inputs the reference ships, not only on a 64^2 synthetic: its |dPSNR| against the
    reference's fp32 run for C1 (box inpainting, 20 NFE) and C2 (Levin09[0], 20 NFE) on the five demo PNGs.  NOT inside the 1e-3 dB parity contract;
    the measured change is the mode's quality contract (printed; bounded here at 0.05 dB so that a regression of the mode is caught)."""
    e = diffpir_amd.Engine(0)
    e.set_precision("f16x1")
    make_model(e, uo.ffhq_hp())
    try:
        cfg = restore.LoopConfig(task="inpaint", iter_num=int(ref["c1_nfe"]), lambda_=1.0, zeta=1.0, noise_level_img=0.0)
        out = restore.restore_batch(e, cfg, ref["c1_y"], mask=ref["c1_mask"], noise_source="host", noise_fn=seeded_noise_fn_np(int(ref["c1_seed"])), use_graph=True).numpy()
        gt = _gt01(ref["c1_gt"])
        gap1 = abs(restore.psnr_batch(out * 2 - 1, gt * 2 - 1) - restore.psnr_batch(ref["c1_out"] * 2 - 1, gt * 2 - 1))
        d1 = float(np.abs(out - ref["c1_out"]).max())
        nfe = int(ref["c2lev20_nfe"])
        cfg = restore.LoopConfig(task="deblur", iter_num=nfe, lambda_=1 * 7, zeta=0.1 * 3)
        out = restore.restore_batch(e, cfg, ref["c2lev_y"], k=ref["c2lev_k"], noise_source="host", noise_fn=seeded_noise_fn_np(int(ref["c2lev20_seed"])), use_graph=True).numpy()
        gt = _gt01(ref["c2lev_gt"])
        gap2 = abs(restore.psnr_batch(out * 2 - 1, gt * 2 - 1) - restore.psnr_batch(ref["c2lev20_out"] * 2 - 1, gt * 2 - 1))
        d2 = float(np.abs(out - ref["c2lev20_out"]).max())
        print(f"f16x1 on the 5 demo images vs the reference's fp32 main(): C1 box inpainting 20 NFE |dPSNR| {gap1:.2e} dB (max pixel diff {d1:.2e}); "
              f"C2 Levin09[0] {nfe} NFE |dPSNR| {gap2:.2e} dB (max pixel diff {d2:.2e})")
        assert gap1 < 0.05 and gap2 < 0.05 and np.isfinite(out).all()
    finally:
        e.close()
