"""-m gpu: the HIP UNet (conv / GroupNorm / attention kernels) against the oracle, layer by layer, and
against the live-reference fixtures.  Everything goes through the C ABI (diffpir_amd.Engine)."""
import numpy as np
import pytest
import torch

from oracle import unet_oracle as uo
from tests.gpu_common import make_model, rel_err

pytestmark = pytest.mark.gpu

TOL_LAYER = 2e-5     # max-abs / max-abs per layer: fp32 with a different summation order (measured: <= 3.5e-6 in both modes)
TOL_OUT = 2e-5


@pytest.fixture(scope="module")
def engine():
    import diffpir_amd
    e = diffpir_amd.Engine(0)
    yield e
    e.close()


def _run_and_compare(engine, hp, B, H, W, labels=None, seed=3, check_taps=True):
    model, sd = make_model(engine, hp)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, 3, H, W), generator=g)
    t = torch.tensor([999, 37, 500, 0, 123, 7][:B])
    y = None if labels is None else torch.tensor(labels)
    taps = {}
    ref = uo.unet_forward(sd, hp, x, t, y, taps=taps)
    out = engine.unet_forward(engine.to_device(x.numpy()), t.numpy(), None if y is None else y.numpy()).numpy()
    worst = ("", 0.0)
    if check_taps:
        for name, tv in taps.items():
            if name == "emb":
                continue
            got = engine.read_tap(name).reshape(tv.shape)
            err = rel_err(got, tv.numpy())
            if err > worst[1]:
                worst = (name, err)
            assert err < TOL_LAYER, f"layer {name}: rel err {err:.3e}"
    err = rel_err(out, ref.numpy())
    assert err < TOL_OUT, f"output rel err {err:.3e} (worst layer {worst})"
    return out, ref.numpy()


def test_tiny_unet_layers(engine):
    _run_and_compare(engine, uo.tiny_hp(), 2, 32, 32)


def test_tiny_unet_batch_and_odd_batch(engine):
    _run_and_compare(engine, uo.tiny_hp(), 3, 64, 64)
    _run_and_compare(engine, uo.tiny_hp(), 1, 32, 32)


def test_uniform_timestep_shares_one_film_row(engine):
    """Round 5: when every image of the batch has the same timestep (what model_fn always passes, utils_model.py:217) and the model is
    class-unconditional, the time embedding and the FiLM projection are evaluated for ONE row and shared.  Same bits as the per-image rows
    (a forward whose timesteps differ only in the LAST image takes the per-image route: its first images must come out identical), and the
    oracle agrees."""
    hp = uo.tiny_hp()
    model, sd = make_model(engine, hp)
    g = torch.Generator().manual_seed(9)
    x = torch.randn((3, 3, 64, 64), generator=g)
    xd = engine.to_device(x.numpy())
    uni = engine.unet_forward(xd, np.array([321, 321, 321])).numpy()
    mixed = engine.unet_forward(xd, np.array([321, 321, 17])).numpy()
    assert np.array_equal(uni[:2], mixed[:2])
    ref = uo.unet_forward(sd, hp, x, torch.tensor([321, 321, 321])).numpy()
    assert rel_err(uni, ref) < TOL_OUT
    # a class-conditional model never shares the row (labels differ per image)
    hpc = uo.tiny_hp(class_cond=True)
    make_model(engine, hpc)
    a = engine.unet_forward(xd, np.array([50, 50, 50]), np.array([1, 2, 3])).numpy()
    b = engine.unet_forward(xd, np.array([50, 50, 50]), np.array([1, 2, 9])).numpy()
    assert np.array_equal(a[:2], b[:2]) and not np.array_equal(a[2], b[2])


def test_tiny_unet_non_square(engine):
    _run_and_compare(engine, uo.tiny_hp(), 2, 32, 64)


def test_tiny_class_conditional(engine):
    _run_and_compare(engine, uo.tiny_hp(class_cond=True), 2, 32, 32, labels=[3, 7])


@pytest.mark.parametrize("tag,hp", [("tiny", uo.tiny_hp()), ("tinycc", uo.tiny_hp(class_cond=True)), ("ffhq", uo.ffhq_hp())])
def test_unet_matches_live_reference_fixture(engine, golden, tag, hp):
    g = golden("unet_" + tag)
    model, _ = make_model(engine, hp)
    y = g["y"] if "y" in g else None
    out = engine.unet_forward(engine.to_device(g["x"]), g["t"], y).numpy()
    assert rel_err(out, g["out"]) < TOL_OUT


def test_ffhq_topology_64(engine):
    _run_and_compare(engine, uo.ffhq_hp(), 2, 64, 64)


def test_ffhq_topology_256_full_size(engine):
    out, ref = _run_and_compare(engine, uo.ffhq_hp(), 1, 256, 256, check_taps=False)
    assert engine.unet_flops(256, 256) == pytest.approx(387.934e9, rel=1e-5)


@pytest.fixture()
def engine_f16x3():
    """A separate engine in operand-split f16x3 MFMA mode (precision is fixed before the weights are packed)."""
    import diffpir_amd
    e = diffpir_amd.Engine(0)
    e.set_precision("f16x3")
    yield e
    e.close()


def test_f16x3_mode_layers_match_oracle_like_fp32(engine_f16x3):
    """Same per-layer tolerance as the exact-fp32 kernels: the split product keeps 22 mantissa bits and accumulates in fp32."""
    _run_and_compare(engine_f16x3, uo.tiny_hp(), 2, 32, 32)
    _run_and_compare(engine_f16x3, uo.tiny_hp(), 3, 64, 64)
    _run_and_compare(engine_f16x3, uo.ffhq_hp(), 2, 64, 64)


def test_f16x3_mode_full_size_and_fixture(engine_f16x3, golden):
    out, ref = _run_and_compare(engine_f16x3, uo.ffhq_hp(), 1, 256, 256, check_taps=False)
    assert rel_err(out, ref) < 1e-5
    g = golden("unet_ffhq")
    o = engine_f16x3.unet_forward(engine_f16x3.to_device(g["x"]), g["t"]).numpy()
    assert rel_err(o, g["out"]) < 1e-5


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_imagenet256_topology_64(precision):
    """BASELINE configs C3/C5 topology (256x256_diffusion_uncond: 552.8 M parameters, 256 base channels, two ResBlocks per
    level, attention at 32/16/8, learn_sigma) at a 64x64 input, layer by layer against the oracle, both arithmetic modes."""
    import diffpir_amd
    e = diffpir_amd.Engine(0)
    try:
        e.set_precision(precision)
        _run_and_compare(e, uo.imagenet256_hp(), 2, 64, 64)
    finally:
        e.close()


def test_missing_weight_is_reported(engine):
    import diffpir_amd
    from diffpir_amd import script_util
    hp = uo.tiny_hp()
    sd = {k: v.numpy() for k, v in uo.synth_state_dict(hp, 0).items()}
    sd.pop("middle_block.1.qkv.weight")
    model = script_util.create_model(64, 64, 1, channel_mult="1,2,2", learn_sigma=True, attention_resolutions="16,32",
                                     num_head_channels=64, use_scale_shift_norm=True, resblock_updown=True, engine=engine)
    with pytest.raises(diffpir_amd.EngineError, match="middle_block.1.qkv.weight"):
        model.load_state_dict(sd)


def test_model_fn_plug_matches_oracle(engine):
    """utils_model.model_fn drop-in: same signature, pred_xstart with the float64-table coefficients."""
    from diffpir_amd import utils_model, script_util, schedule
    from oracle import diffpir_oracle as do
    hp = uo.tiny_hp()
    model, sd = make_model(engine, hp)
    diffusion = script_util.create_gaussian_diffusion(steps=1000, learn_sigma=True)
    dt = schedule.DriverTables.make()
    odt, odtab = do.DriverTables(), do.DiffusionTables()
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 3, 32, 32), generator=g)
    for t in (999, 400, 3):
        sig = dt.reduced[t]
        x0 = utils_model.model_fn(engine.to_device(x.numpy()), noise_level=sig * 255, model_out_type="pred_xstart",
                                  model_diffusion=model, diffusion=diffusion, ddim_sample=False,
                                  alphas_cumprod=dt.alphas_cumprod).numpy()
        ref = do.model_fn_xstart(sd, hp, x, sig * 255, odt, odtab).numpy()
        assert np.abs(x0 - ref).max() < 5e-4
        assert x0.min() >= -1 and x0.max() <= 1


def test_model_fn_ddim_flag_matches_live_reference_fixture(engine, golden):
    """ddim_sample=True is accepted and equals the live reference's pred_xstart (tests/golden/model_fn.npz)."""
    from diffpir_amd import utils_model, script_util, schedule
    g = golden("model_fn")
    model, _ = make_model(engine, uo.tiny_hp())
    diffusion = script_util.create_gaussian_diffusion(steps=1000, learn_sigma=True)
    dt = schedule.DriverTables.make()
    for j, sig in enumerate(g["noise_levels"]):
        for ddim in (False, True):
            x0 = utils_model.model_fn(engine.to_device(g["x"]), noise_level=float(sig) * 255, model_out_type="pred_xstart",
                                      model_diffusion=model, diffusion=diffusion, ddim_sample=ddim,
                                      alphas_cumprod=dt.alphas_cumprod).numpy()
            assert np.abs(x0 - g[f"x0_{j}_ddim"]).max() < 5e-4
