"""-m gpu: on-device degradation synthesis and metrics (dpir_degrade / dpir_metrics) against fixtures produced by
the reference's own functions (tests/golden/degrade.npz) and the f16x1 reduced-precision mode's quality contract (8f-2)."""
import numpy as np
import pytest
import torch

import diffpir_amd
from diffpir_amd import degrade as dgr, restore, synth
from oracle import unet_oracle as uo, diffpir_oracle as do, degrade_oracle as dgo
from tests.gpu_common import make_model, seeded_noise_fn_np, seeded_noise_fn_torch, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    e = diffpir_amd.Engine(0)
    yield e
    e.close()


def test_blur_of_the_uint8_image_is_bit_exact_with_scipy(engine, golden):
    g = golden("degrade")
    y, keep = dgr.degrade(engine, "deblur", g["gt"], k=g["k"], noise_level_img=0.0)
    got = y.numpy()
    # float64 accumulation in scipy's tap order, C cast to uint8: identical quantised values (incl. the flat 200 -> 199 patch)
    np.testing.assert_array_equal(got, g["deblur_y_sigma0"])
    y, _ = dgr.degrade(engine, "deblur", g["gt"], k=g["k"], noise_level_img=12.75 / 255, noise=g["noise"])
    np.testing.assert_allclose(y.numpy(), g["deblur_y"], atol=1e-7)


def test_downsampling_and_masking_match_the_reference(engine, golden):
    g = golden("degrade")
    y, _ = dgr.degrade(engine, "sr", g["gt"], noise_level_img=0.0, sf=4)
    np.testing.assert_allclose(y.numpy(), g["sr4_y_clean"], atol=2e-6)            # utils_image.imresize_np
    y, _ = dgr.degrade(engine, "inpaint", g["gt"], mask=g["mask"], noise_level_img=0.0)
    np.testing.assert_array_equal(y.numpy(), g["inpaint_y"])


def test_device_noise_is_awgn_of_the_right_level_and_shard_invariant(engine, golden):
    g = golden("degrade")
    sig = 12.75 / 255
    a, _ = dgr.degrade(engine, "deblur", g["gt"], k=g["k"], noise_level_img=sig, seed=5, image_offset=10)
    b, _ = dgr.degrade(engine, "deblur", g["gt"][1:], k=g["k"][1:], noise_level_img=sig, seed=5, image_offset=11)
    np.testing.assert_array_equal(a.numpy()[1], b.numpy()[0])
    n = (a.numpy() - g["deblur_y_sigma0"]) * 2 / (2 * sig)          # back to the standard normal
    assert abs(n.mean()) < 0.02 and abs(n.std() - 1) < 0.02


def test_metrics_match_calculate_psnr_batch_and_the_y_channel(engine, golden):
    g = golden("degrade")
    gt_d = engine.to_device(g["gt"])
    p, py = dgr.metrics(engine, engine.to_device(g["x0"]), gt_d)
    np.testing.assert_allclose(p, g["psnr"], atol=2e-5)
    np.testing.assert_allclose(py, g["psnr_y"], atol=2e-5)
    # identical images: infinite PSNR, as torch.where(mse == 0, inf, ...)
    same = (g["gt"].transpose(0, 3, 1, 2) / 255.0).astype(np.float32)
    p, _ = dgr.metrics(engine, engine.to_device(same), gt_d)
    assert np.all(p > 80)


def test_yaml_driver_runs_end_to_end_on_the_device(tmp_path):
    """configs/engine_example.yaml-style run through the driver: device degradation -> loop -> device metrics."""
    import yaml
    from diffpir_amd import main_ddpir
    cfg = yaml.safe_load(open("configs/engine_example.yaml"))
    cfg.update(iter_num=4, batch_size=2, task="deblur")
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump(cfg))
    res = main_ddpir.main(["--opt", str(p), "--synthetic", "2", "--max-sweeps", "1"])
    assert len(res) == 1 and np.isfinite(res[0])


@pytest.mark.parametrize("mode,ddim", [("DPS_y0", False), ("DPS_yt", True)])
def test_yaml_driver_runs_the_dps_modes_with_host_noise(tmp_path, mode, ddim):
    """generate_mode DPS_y0 / DPS_yt through the YAML driver with engine_noise: host (round-3 advisor: the driver pre-drew the DiffPIR
    loop's noise and the DPS path then refused it) -- the DPS modes pull their own draw order through noise_fn; ddim_sample reaches the loop."""
    import yaml
    from diffpir_amd import main_ddpir
    cfg = yaml.safe_load(open("configs/engine_example.yaml"))
    cfg.update(iter_num=3, batch_size=2, task="sr", sf=4, sr_mode="cubic", generate_mode=mode, ddim_sample=ddim, engine_noise="host",
               noise_init_img=100.0 if mode == "DPS_yt" else "max")
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump(cfg))
    res = main_ddpir.main(["--opt", str(p), "--synthetic", "2", "--max-sweeps", "1"])
    assert len(res) == 1 and np.isfinite(res[0])


def test_f16x1_mode_quality_contract():
    """precision 'f16x1' (SURVEY.md 8f-2): f16 operands, ONE MFMA per product, fp32 accumulation and fp32 GroupNorm / softmax /
    residual stream -- the reference's use_fp16 recipe.  It is a reduced-precision mode: NOT held to the 1e-3 dB parity bar, but
    to its own contract: per-forward relative error <= 5e-3 and |dPSNR| <= 0.05 dB on a 20-NFE restoration."""
    e = diffpir_amd.Engine(0)
    try:
        e.set_precision("f16x1")
        hp = uo.ffhq_hp()
        model, sd = make_model(e, hp)
        x = torch.randn((2, 3, 64, 64), generator=torch.Generator().manual_seed(3))
        t = torch.tensor([700, 50])
        ref = uo.unet_forward(sd, hp, x, t).numpy()
        out = e.unet_forward(e.to_device(x.numpy()), t.numpy()).numpy()
        err = rel_err(out, ref)
        case = synth.make_case("inpaint", B=1, H=64, W=64, seed=4)
        cfg = restore.LoopConfig(task="inpaint", iter_num=20, noise_level_img=0.0, lambda_=1.0, zeta=1.0)
        got = restore.restore_batch(e, cfg, case["y"], mask=case["mask"], noise_source="host", noise_fn=seeded_noise_fn_np(8), use_graph=True).numpy()
        oref = do.restore(sd, hp, do.LoopConfig("inpaint", 20, 0.0, 1.0, 1.0), torch.from_numpy(case["y"]),
                          mask=torch.from_numpy(case["mask"]).float(), noise_fn=seeded_noise_fn_torch(8)).numpy()
        gt = case["gt"] * 2 - 1
        gap = abs(restore.psnr_batch(got * 2 - 1, gt) - restore.psnr_batch(oref * 2 - 1, gt))
        print(f"f16x1: forward rel err {err:.3e}; 20-NFE inpaint max|diff| {np.abs(got - oref).max():.3e}, |dPSNR| {gap:.3e} dB")
        assert 1e-5 < err < 5e-3            # it IS reduced precision (not the f16x3 path by accident) and within the contract
        assert gap < 0.05
    finally:
        e.close()
