"""-m gpu: the whole restoration loop (dpir_run_loop, eager and hipGraph) and the stepwise drop-in loop
against the live-reference loop fixtures and the oracle; full-size PSNR parity on config C1."""
import numpy as np
import pytest
import torch

from diffpir_amd import restore, script_util
from oracle import unet_oracle as uo, diffpir_oracle as do
from tests.gpu_common import make_model, seeded_noise_fn_np, seeded_noise_fn_torch, oracle_pair, fft_prox_parity

pytestmark = pytest.mark.gpu

CASES = [
    ("deblur", dict(task="deblur", iter_num=6, lambda_=7.0, zeta=0.3), 42),
    ("deblur_eta", dict(task="deblur", iter_num=5, lambda_=7.0, zeta=0.3, eta=0.7), 43),
    ("inpaint", dict(task="inpaint", iter_num=6, noise_level_img=0.0, lambda_=1.0, zeta=1.0), 44),
    ("sr_blur", dict(task="sr", iter_num=5, lambda_=6.0, zeta=0.25, sf=4), 45),
    ("sr_cubic", dict(task="sr", iter_num=5, lambda_=6.0, zeta=0.25, sf=4, sr_mode="cubic", inIter=2, gamma=0.5), 46),
]


@pytest.fixture(scope="module")
def engine():
    import diffpir_amd
    e = diffpir_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def tiny(engine):
    return make_model(engine, uo.tiny_hp())


def _inputs(golden, name):
    g, ops = golden("loops"), golden("operators")
    k = mask = None
    if name.startswith("deblur"):
        y, k = g["deblur_y"], g["deblur_k"]
    elif name == "inpaint":
        y, mask = g["inpaint_y"], g["inpaint_mask"]
    else:
        y = g["sr_y"]
        k = np.stack([ops["k_bic4"], ops["k_bic4"]])[:, None].astype(np.float32)
    return y, k, mask, g[name + "_out"]


@pytest.mark.parametrize("name,kw,seed", CASES)
@pytest.mark.parametrize("graph", [False, True])
def test_run_loop_matches_live_reference_fixture(engine, tiny, golden, name, kw, seed, graph):
    y, k, mask, ref = _inputs(golden, name)
    cfg = restore.LoopConfig(**kw)
    out = restore.restore_batch(engine, cfg, y, k=k, mask=mask, noise_source="host", noise_fn=seeded_noise_fn_np(seed),
                                use_graph=graph).numpy()
    gt = golden("loops")["gt" if not name.startswith("sr") else "sr_gt"]
    if name in ("deblur", "deblur_eta", "sr_blur"):
        # FFT prox: the bound follows the reference's own fp32 rounding noise (tests/gpu_common.py::fft_prox_parity).  The
        # yardstick is the oracle loop with the prox in float64 (`exact`), run here.  The fp32 oracle rerun (`oref`) equals the
        # live-reference fixture bit for bit on the machine that generated it (tests/test_oracle_golden.py) but NOT on another
        # CPU (different FFT / GEMM code paths: measured 2e-3..8e-3 on the MI355X box's host) -- the same ill-conditioning the
        # gate is built around, so the rerun is held to the same floor as the engine rather than to bit equality.
        ocfg = do.LoopConfig(kw["task"], kw["iter_num"], 12.75 / 255, kw["lambda_"], kw["zeta"], eta=kw.get("eta", 0.0), sf=kw.get("sf", 1))
        _, sd = tiny
        oref, exact = oracle_pair("loops_" + name, sd, uo.tiny_hp(), ocfg, y, k, seed)
        fl = ref - exact
        assert np.sqrt(np.mean((oref - ref) ** 2)) <= 2.0 * np.sqrt(np.mean(fl * fl)) + 2e-5
        fft_prox_parity(out, ref, gt, f"{name} graph={graph}", exact=exact)
    else:
        assert np.abs(out - ref).max() < 2e-3
        dpsnr = abs(restore.psnr_batch(out * 2 - 1, gt * 2 - 1) - restore.psnr_batch(ref * 2 - 1, gt * 2 - 1))
        assert dpsnr < 1e-3, dpsnr


@pytest.mark.parametrize("name,kw,seed", CASES[:1] + CASES[2:4])
def test_stepwise_plug_loop_equals_run_loop(engine, tiny, golden, name, kw, seed):
    model, _ = tiny
    diffusion = script_util.create_gaussian_diffusion(steps=1000, learn_sigma=True)
    y, k, mask, ref = _inputs(golden, name)
    cfg = restore.LoopConfig(**kw)
    dev = lambda a, dt=np.float32: None if a is None else engine.to_device(a, dt)
    out = restore.restore_batch_stepwise(model, diffusion, cfg, dev(y), k=dev(k), mask=dev(mask, np.uint8),
                                         noise_fn=seeded_noise_fn_np(seed)).numpy()
    if name in ("deblur", "sr_blur"):
        _, sd = tiny
        ocfg = do.LoopConfig(kw["task"], kw["iter_num"], 12.75 / 255, kw["lambda_"], kw["zeta"], sf=kw.get("sf", 1))
        oref, exact = oracle_pair("loops_" + name, sd, uo.tiny_hp(), ocfg, y, k, seed)
        fft_prox_parity(out, ref, golden("loops")["gt" if name == "deblur" else "sr_gt"], f"stepwise {name}", exact=exact)
    else:
        assert np.abs(out - ref).max() < 2e-3


def test_graph_replay_is_bitwise_repeatable_and_device_noise_is_shard_invariant(engine, tiny, golden):
    y, k, mask, _ = _inputs(golden, "deblur")
    cfg = restore.LoopConfig(task="deblur", iter_num=6, lambda_=7.0, zeta=0.3)
    yd, kd = engine.to_device(y), engine.to_device(k)
    o = engine.empty((2, 3, 32, 32))
    a = restore.restore_batch(engine, cfg, yd, k=kd, noise_source="device", seed=7, use_graph=True, out_f32=o).numpy()
    b = restore.restore_batch(engine, cfg, yd, k=kd, noise_source="device", seed=7, use_graph=True, out_f32=o).numpy()
    np.testing.assert_array_equal(a, b)
    c = restore.restore_batch(engine, cfg, yd, k=kd, noise_source="device", seed=7, use_graph=False).numpy()
    np.testing.assert_array_equal(a, c)
    # image 1 restored alone with image_offset=1 equals image 1 of the batch (sharding invariance)
    d = restore.restore_batch(engine, cfg, y[1:], k=k[1:], noise_source="device", seed=7, image_offset=1).numpy()
    assert np.abs(d[0] - a[1]).max() < 1e-4
    assert np.isfinite(a).all()


def test_u8_output_and_skip_dead_final_eval(engine, tiny, golden):
    y, k, mask, ref = _inputs(golden, "inpaint")
    cfg = restore.LoopConfig(task="inpaint", iter_num=6, noise_level_img=0.0, lambda_=1.0, zeta=1.0)
    f, u = restore.restore_batch(engine, cfg, y, mask=mask, noise_source="host", noise_fn=seeded_noise_fn_np(44), return_u8=True)
    np.testing.assert_array_equal(u.numpy(), do.tensor2uint_batch(torch.from_numpy(f.numpy())))
    g = restore.restore_batch(engine, cfg, y, mask=mask, noise_source="host", noise_fn=seeded_noise_fn_np(44),
                              skip_dead_final_eval=True).numpy()
    np.testing.assert_array_equal(g, f.numpy())        # Q2: the last UNet evaluation never reaches the output


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_config_c1_full_size_psnr_parity(precision):
    """BASELINE config 1: FFHQ topology, 256x256 box inpainting, 20 NFE, B=1 -- engine vs oracle on identical
    y / mask / weights / host-drawn noise: |dPSNR| <= 1e-3 dB (north-star tolerance)."""
    import diffpir_amd
    from diffpir_amd import synth
    engine = diffpir_amd.Engine(0)
    engine.set_precision(precision)
    hp = uo.ffhq_hp()
    model, sd = make_model(engine, hp)
    case = synth.make_case("inpaint", B=1, H=256, W=256, seed=42)
    cfg = restore.LoopConfig(task="inpaint", iter_num=20, noise_level_img=0.0, lambda_=1.0, zeta=1.0)
    out = restore.restore_batch(engine, cfg, case["y"], mask=case["mask"], noise_source="host",
                                noise_fn=seeded_noise_fn_np(42)).numpy()
    ocfg = do.LoopConfig("inpaint", 20, 0.0, 1.0, 1.0)
    ref = do.restore(sd, hp, ocfg, torch.from_numpy(case["y"]), mask=torch.from_numpy(case["mask"]).float(),
                     noise_fn=seeded_noise_fn_torch(42)).numpy()
    gt = case["gt"] * 2 - 1
    p_eng, p_ref = restore.psnr_batch(out * 2 - 1, gt), restore.psnr_batch(ref * 2 - 1, gt)
    print(f"C1 [{precision}] PSNR engine {p_eng:.5f} dB, oracle {p_ref:.5f} dB, max|diff| {np.abs(out - ref).max():.3e}")
    assert abs(p_eng - p_ref) <= 1e-3
    # integer mask semantics: kept pixels follow the data term exactly as in the oracle
    assert np.abs(out - ref).max() < 5e-3
    engine.close()


@pytest.mark.parametrize("mode", ["repaint", "vanilla"])
@pytest.mark.parametrize("graph", [False, True])
def test_inpaint_generate_modes_match_live_reference_fixture(engine, tiny, golden, mode, graph):
    """generate_mode repaint / vanilla (main_ddpir.py:349-358, 385, 448) through dpir_run_loop and through the stepwise plugs."""
    g, gm = golden("loops"), golden("loops_modes")
    cfg = restore.LoopConfig(task="inpaint", iter_num=6, noise_level_img=0.0, lambda_=1.0, zeta=1.0, generate_mode=mode)
    seed = int(gm[f"inpaint_{mode}_seed"])
    ref = gm[f"inpaint_{mode}_out"]
    out = restore.restore_batch(engine, cfg, g["inpaint_y"], mask=g["inpaint_mask"], noise_source="host",
                                noise_fn=seeded_noise_fn_np(seed), use_graph=graph).numpy()
    assert np.abs(out - ref).max() < 2e-3
    if not graph:
        model, _ = tiny
        from diffpir_amd import script_util
        diffusion = script_util.create_gaussian_diffusion(steps=1000, learn_sigma=True)
        eng = engine
        sw = restore.restore_batch_stepwise(model, diffusion, cfg, eng.to_device(g["inpaint_y"]), mask=eng.to_device(g["inpaint_mask"], np.uint8),
                                            noise_fn=seeded_noise_fn_np(seed)).numpy()
        assert np.abs(sw - ref).max() < 2e-3
        # device-noise path (Philox draw 3 for the repaint mix): runs, is finite and deterministic
        a = restore.restore_batch(engine, cfg, g["inpaint_y"], mask=g["inpaint_mask"], noise_source="device", seed=7).numpy()
        b = restore.restore_batch(engine, cfg, g["inpaint_y"], mask=g["inpaint_mask"], noise_source="device", seed=7, use_graph=True).numpy()
        assert np.isfinite(a).all() and np.array_equal(a, b)


def test_generate_modes_are_inpainting_only(engine, tiny, golden):
    g = golden("loops")
    with pytest.raises(NotImplementedError):
        restore.restore_batch(engine, restore.LoopConfig(task="deblur", iter_num=3, generate_mode="repaint"), g["deblur_y"], k=g["deblur_k"])
