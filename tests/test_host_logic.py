"""CPU tests: host-side schedule tables against the live-reference fixtures and the oracle, the C-ABI
library's exported symbols, and the fail-loudly contract (no GPU -> no fallback)."""
import os
import re

import numpy as np
import pytest
import torch

from diffpir_amd import _lib, schedule
from diffpir_amd.restore import LoopConfig
from oracle import diffpir_oracle as do

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,kw", [
    ("deblur100", dict(iter_num=100, sigma=12.75 / 255, lambda_=7.0, zeta=0.3)),
    ("inpaint20", dict(iter_num=20, sigma=0.001, lambda_=1.0, zeta=1.0)),
    ("sr100", dict(iter_num=100, sigma=12.75 / 255, lambda_=6.0, zeta=0.25)),
])
def test_steps_match_reference_tables(golden, name, kw):
    g = golden("schedule")
    dt, steps, arr = schedule.build_steps(**kw)
    assert [s["t"] for s in steps] == list(g[name + "_t"])
    np.testing.assert_array_equal(np.array([s["tau"] for s in steps], np.float32), g[name + "_tau"])
    np.testing.assert_array_equal(dt.sqrt_ac, g["drv_sqrt_ac"])
    np.testing.assert_array_equal(dt.sqrt_1m_ac, g["drv_sqrt_1m_ac"])
    d = schedule.DiffusionTables.make()
    np.testing.assert_array_equal(d.sqrt_recip_ac, g["sqrt_recip_ac"])
    assert len(arr) == len(steps) and arr[0].t == 999 and arr[len(steps) - 1].last == 1


@pytest.mark.parametrize("eta,zeta", [(0.0, 0.3), (0.7, 0.3), (0.0, 1.0), (1.0, 0.0)])
def test_renoise_coefficients_match_torch_expression(eta, zeta):
    """k1,q,es,k2 must reproduce main_ddpir.py:451-456 evaluated with torch 0-dim float32 tensors."""
    dt, steps, _ = schedule.build_steps(iter_num=12, sigma=0.05, lambda_=7.0, zeta=zeta, eta=eta)
    odt = do.DriverTables()
    g = torch.Generator().manual_seed(0)
    x, x0, n1, n2 = (torch.randn(1, 3, 8, 8, generator=g) for _ in range(4))
    for st in steps:
        if st["last"]:
            continue
        ref = do.renoise(x, x0, odt, st["t"], st["t_im1"], eta, zeta, n1, n2)
        eps = (x - np.float32(st["sa_t"]) * x0) / np.float32(st["s1m_t"])
        mine = np.float32(st["sa_p"]) * x0 + np.float32(st["k1"]) * (np.float32(st["q"]) * eps + np.float32(st["es"]) * n1) \
            + np.float32(st["k2"]) * n2
        np.testing.assert_allclose(mine.numpy(), ref.numpy(), rtol=0, atol=1e-6)
        # scalars themselves are bit-exact
        es = eta * odt.sqrt_1m_ac[st["t_im1"]] / odt.sqrt_1m_ac[st["t"]] * torch.sqrt(odt.betas[st["t"]])
        assert np.float32(st["es"]) == np.float32(float(es))
        q = torch.sqrt(odt.sqrt_1m_ac[st["t_im1"]] ** 2 - es ** 2)
        assert np.float32(st["q"]) == np.float32(float(q))
        k2 = np.sqrt(zeta) * odt.sqrt_1m_ac[st["t_im1"]]
        assert np.float32(st["k2"]) == np.float32(float(k2))


def test_uniform_skip_sequence():
    assert schedule.make_seq(1000, 10, "uniform") == [i * 100 for i in range(10)] + [999]
    assert schedule.make_seq(1000, 20) == do.make_seq(1000, 20)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "diffpir_engine.h")).read()
    declared = set(re.findall(r"\b(dpir_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dpir_status", "dpir_task"}
    lib = _lib.load()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.dpir_version() == _lib.ABI_VERSION == 2
    # the development-only header (probes and the isolated conv bench) resolves from its OWN library; the product library exports
    # none of it
    dbg = open(os.path.join(ROOT, "include", "diffpir_debug.h")).read()
    dbg_syms = set(re.findall(r"\b(dpir_debug_[a-z0-9_]+)\s*\(", dbg))
    assert len(dbg_syms) >= 5
    assert not [n for n in sorted(dbg_syms) if hasattr(lib, n)], "development probes leaked into libdiffpir_hip.so"
    dlib = _lib.load_debug()
    assert not [n for n in sorted(dbg_syms) if not hasattr(dlib, n)]


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(_lib.Step) == 48
    assert C.sizeof(_lib.UNetDesc) == 4 * 7 + 32 + 4 + 32 + 4
    assert _lib.LoopDesc.y_dev.offset == 48
    # every struct of the header, as a C compiler lays it out, against the ctypes mirror
    import shutil
    import subprocess
    import tempfile
    if shutil.which("gcc"):
        pairs = [("dpir_unet_desc", _lib.UNetDesc), ("dpir_tensor", _lib.Tensor), ("dpir_step", _lib.Step), ("dpir_loop_desc", _lib.LoopDesc),
                 ("dpir_dps_coef", _lib.DpsCoef), ("dpir_psample_coef", _lib.PSampleCoef), ("dpir_degrade_desc", _lib.DegradeDesc)]
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "sz.c")
            open(src, "w").write('#include "diffpir_engine.h"\n#include <stdio.h>\nint main(void){' +
                                 "".join(f'printf("%zu\\n", sizeof({c}));' for c, _ in pairs) + "return 0;}\n")
            subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(td, "sz")], check=True)
            sizes = [int(v) for v in subprocess.run([os.path.join(td, "sz")], check=True, capture_output=True, text=True).stdout.split()]
        assert sizes == [C.sizeof(t) for _, t in pairs], list(zip([c for c, _ in pairs], sizes, [C.sizeof(t) for _, t in pairs]))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_fails_loudly_without_gpu():
    import diffpir_amd
    with pytest.raises(diffpir_amd.EngineError):
        diffpir_amd.Engine(0)


def test_unsupported_modes_raise():
    with pytest.raises(NotImplementedError):
        LoopConfig(generate_mode="DPS_y0").check_supported()
    from diffpir_amd import script_util
    with pytest.raises(NotImplementedError):
        script_util.create_model(256, 128, 1, learn_sigma=True, num_head_channels=64, use_scale_shift_norm=True,
                                 resblock_updown=True, use_fp16=True)


def test_factory_mirrors_reference_call_sequence():
    """main_ddpir.py:219-233 runs unchanged up to the point where a GPU is needed."""
    from diffpir_amd import utils_model, script_util
    args = utils_model.create_argparser(dict(model_path="", num_channels=128, num_res_blocks=1, attention_resolutions="16")).parse_args([])
    model, diffusion = script_util.create_model_and_diffusion(
        **script_util.args_to_dict(args, script_util.model_and_diffusion_defaults().keys()))
    assert model.desc.model_channels == 128 and model.desc.out_channels == 6
    assert list(model.desc.attention_ds)[:1] == [16] and model.desc.n_channel_mult == 0
    assert diffusion.sqrt_recip_alphas_cumprod.dtype == np.float64


def test_product_weight_schema_equals_oracle_and_reference():
    """diffpir_amd.weights (product) and oracle.unet_oracle (checker) derive the reference's state-dict schema
    independently; both must agree with each other (keys, shapes, order, values)."""
    from diffpir_amd import weights
    from oracle import unet_oracle as uo
    for name, ohp in (("ffhq", uo.ffhq_hp()), ("tiny", uo.tiny_hp()), ("imagenet512", uo.imagenet512_hp())):
        spec = weights.state_dict_spec(weights.model_hp(name))
        ospec = uo.state_dict_spec(ohp)
        assert [(k, tuple(s)) for k, s, _ in spec] == [(k, tuple(s)) for k, s, _ in ospec]
    a = weights.synth_state_dict("tiny", 0)
    b = uo.synth_state_dict(uo.tiny_hp(), 0)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k].numpy())


def test_yaml_driver_accepts_reference_style_configs(tmp_path):
    """The driver's config surface: same flat keys as the reference's configs/*.yaml, same derived fields and sweeps."""
    from diffpir_amd import main_ddpir as drv
    ref_dir = "/root/reference/configs"
    paths = [os.path.join(ROOT, "configs", "engine_example.yaml")]
    if os.path.isdir(ref_dir):
        paths += [os.path.join(ref_dir, f) for f in ("deblur.yaml", "inpaint.yaml", "sisr.yaml")]
    for p in paths:
        c = drv.parse_config(p)
        lam, zeta = drv.sweeps(c)[0]
        cfg = drv.loop_config(c, lam, zeta)
        cfg.check_supported()
        assert cfg.sigma == max(0.001, c.noise_level_img)
        if c.task == "deblur":
            assert (lam, zeta) == (c.lambda_ * 7, c.zeta * 3) and cfg.engine_task() == 0
        if c.task == "sr":
            assert len(drv.sweeps(c)) == 11 and cfg.engine_task() == 1
        if c.task == "inpaint":
            assert cfg.engine_task() == 2 and cfg.iter_num == 20


def test_host_noise_draw_order_with_repaint_matches_the_oracle_loop():
    """restore.draw_host_noise must consume noise_fn exactly like the reference loop: init, then per step
    [repaint mix], p_sample (dead), and unless last: eta draw, zeta draw (main_ddpir.py:315, 355-358, 448-456)."""
    import torch
    from diffpir_amd import restore
    from oracle import diffpir_oracle as do, unet_oracle as uo
    cfg = restore.LoopConfig(task="inpaint", iter_num=5, noise_level_img=0.0, lambda_=1.0, zeta=1.0, generate_mode="repaint")
    _, steps, _ = restore._steps(cfg)
    shape = (1, 3, 8, 8)
    counter = {"n": 0}
    def nf(shp):
        counter["n"] += 1
        return np.full(shp, counter["n"], np.float32)
    init, n1, n2, nrp = restore.draw_host_noise(nf, steps, shape, need_n1=True, repaint=True)
    # oracle: same loop with a stub denoiser, recording which draw index feeds which use
    seen = []
    ocount = {"n": 0}
    def onf(like):
        ocount["n"] += 1
        seen.append(ocount["n"])
        return torch.full(like.shape, float(ocount["n"]))
    ocfg = do.LoopConfig(task="inpaint", iter_num=5, noise_level_img=0.0, lambda_=1.0, zeta=1.0, generate_mode="repaint")
    y = torch.rand(shape); mask = (torch.rand(shape) > 0.5).float()
    do.restore(None, uo.tiny_hp(), ocfg, y, mask=mask, noise_fn=onf, denoiser=lambda x, t: x * 0.5)
    assert counter["n"] == ocount["n"]                       # same number of draws
    assert init.flat[0] == 1
    per = 4                                                  # repaint, p_sample, eta, zeta
    for i, s in enumerate(steps):
        assert nrp[i].flat[0] == 2 + per * i                 # the repaint draw is the first draw of every step
        if not s["last"]:
            assert n1[i].flat[0] == 2 + per * i + 2 and n2[i].flat[0] == 2 + per * i + 3
    # without repaint the layout is the round-1 one
    counter["n"] = 0
    init, n1, n2 = restore.draw_host_noise(nf, steps, shape, need_n1=False)
    assert n1 is None and n2[0].flat[0] == 4


def test_generate_mode_validation():
    from diffpir_amd import restore
    restore.LoopConfig(task="inpaint", generate_mode="repaint").check_supported()
    restore.LoopConfig(task="inpaint", generate_mode="vanilla", ddim_sample=True).check_supported()
    for bad in (dict(task="deblur", generate_mode="repaint"), dict(task="inpaint", generate_mode="DPS_y0"),
                dict(task="inpaint", iter_num_U=2), dict(task="deblur", model_output_type="pred_x_prev")):
        with pytest.raises(NotImplementedError):
            restore.LoopConfig(**bad).check_supported()


@pytest.mark.parametrize("skip_type,iter_num", [("uniform", 5), ("uniform", 10), ("quad", 7)])
def test_engine_step_table_equals_oracle_step_table(skip_type, iter_num):
    """The per-step scalars the engine uploads (schedule.build_steps) against the oracle's restatement of main_ddpir.py:274-347
    (pinned to the live reference by the loop fixtures): same timesteps, same tau bits, same 'last' flag."""
    _, steps, arr = schedule.build_steps(iter_num=iter_num, sigma=0.05, lambda_=7.0, zeta=0.3, skip_type=skip_type)
    _, osteps = do.step_tables(do.LoopConfig(task="deblur", iter_num=iter_num, noise_level_img=0.05, lambda_=7.0, zeta=0.3,
                                             skip_type=skip_type))
    assert [s["t"] for s in steps] == [s["t_i"] for s in osteps]
    assert [bool(s["last"]) for s in steps] == [bool(s["last"]) for s in osteps]
    assert [s["t_im1"] for s in steps if not s["last"]] == [s["t_im1"] for s in osteps if not s["last"]]
    np.testing.assert_array_equal(np.array([s["tau"] for s in steps], np.float32),
                                  np.array([float(s["tau"]) for s in osteps], np.float32))
    assert arr[len(steps) - 1].last == 1 and sum(a.last for a in arr) == 1


def test_load_checkpoint_reads_a_reference_style_state_dict(tmp_path):
    """weights.load_checkpoint: torch.load of a guided-diffusion state dict (fp16 or fp32 tensors) -> fp32 numpy, reference keys kept."""
    from diffpir_amd import weights
    from oracle import unet_oracle as uo
    sd = uo.synth_state_dict(uo.tiny_hp(), 3)
    half = {k: (v.half() if i % 2 else v) for i, (k, v) in enumerate(sd.items())}      # checkpoints saved with use_fp16 hold f16 tensors
    path = os.path.join(tmp_path, "tiny.pt")
    torch.save(half, path)
    got = weights.load_checkpoint(path)
    assert list(got) == list(sd)
    for i, (k, v) in enumerate(sd.items()):
        assert got[k].dtype == np.float32
        ref = v.half().float().numpy() if i % 2 else v.numpy()
        np.testing.assert_array_equal(got[k], ref)


def test_t_start_and_duplicate_final_steps_in_the_engine_step_table():
    """noise_init_img / skip_noise_model_t reach the step table (round-1 advisor finding: silently dropped), and a quad schedule
    with iter_num > T/2 yields TWO final steps, like the reference's `seq[i] == seq[-1]` test."""
    from diffpir_amd import restore
    from oracle import diffpir_oracle as do
    cfg = restore.LoopConfig(task="inpaint", iter_num=8, noise_level_img=0.0, lambda_=1.0, zeta=1.0, noise_init_img=60.0)
    dt, steps, arr = restore._steps(cfg)
    ocfg = do.LoopConfig(task="inpaint", iter_num=8, noise_level_img=0.0, lambda_=1.0, zeta=1.0, noise_init_img=60.0)
    odt, osteps = do.step_tables(ocfg)
    t_start = ocfg.t_start(odt)
    assert restore.t_start_of(cfg, dt.reduced) == t_start < 999
    assert [s["t"] for s in steps] == [s["t_i"] for s in osteps if s["t_i"] <= t_start]
    assert len(steps) < len(osteps)
    dt, steps, arr = restore._steps(restore.LoopConfig(task="inpaint", iter_num=520, noise_level_img=0.0, lambda_=1.0, zeta=1.0))
    assert [s["last"] for s in steps][-3:] == [0, 1, 1]
    # skip_noise_model_t: accepted while the branch it guards is dead (iter_num <= T - noise_model_t), refused beyond
    restore._steps(restore.LoopConfig(task="deblur", iter_num=100, skip_noise_model_t=True))
    with pytest.raises(NotImplementedError):
        restore._steps(restore.LoopConfig(task="deblur", iter_num=999, skip_noise_model_t=True))


@pytest.mark.parametrize("mode", ["DPS_y0", "DPS_yt"])
def test_dps_host_noise_draw_order_matches_the_oracle_loop(mode):
    """restore.dps_host_noise_shapes -- what the YAML driver pre-draws (and what a rank with an empty shard consumes to keep the shared
    generator in step) -- is the sequence of randn_like shapes the reference's DPS loop requests: init, then per step the sampler's draw
    (also on the dead final step) and, DPS_yt on non-final steps, the y_t draw (main_ddpir.py:315, 371-373, 440; gaussian_diffusion.py:430)."""
    from diffpir_amd import restore
    from oracle import unet_oracle as uo, diffpir_oracle as do
    hp = uo.tiny_hp()
    sd = uo.synth_state_dict(hp, 0)
    B, h, sf, nfe = 1, 16, 4, 4
    kw = dict(noise_init_img=100.0) if mode == "DPS_yt" else {}
    ocfg = do.LoopConfig("sr", nfe, 12.75 / 255, 6.0, 0.25, sf=sf, sr_mode="cubic", generate_mode=mode, **kw)
    seen = []

    def nf(like):
        seen.append(tuple(like.shape))
        return torch.zeros(like.shape)
    do.restore_dps_y0(sd, hp, ocfg, torch.rand((B, 3, h, h)), noise_fn=nf)
    cfg = restore.LoopConfig(task="sr", iter_num=nfe, lambda_=6.0, zeta=0.25, sf=sf, sr_mode="cubic", generate_mode=mode, **kw)
    _, steps, _ = restore._steps(cfg)
    assert restore.dps_host_noise_shapes(cfg, steps, B, h * sf, h * sf) == seen


def test_gradient_mode_plugs_refuse_what_they_cannot_do_before_touching_the_device():
    """The argument errors of the round-4 plugs (model_fn output types, grad_and_value operators, Resizer factors) are host-side."""
    from diffpir_amd import utils_model
    from diffpir_amd.utils_resizer import Resizer
    with pytest.raises(ValueError, match="model_out_type"):
        utils_model.model_fn(None, 10.0, None, model_out_type="pred_nonsense", alphas_cumprod=np.linspace(0.99, 0.01, 1000))
    with pytest.raises(NotImplementedError, match="Resizer"):
        utils_model.grad_and_value(operator=lambda v: v, x=None, x_hat=None, measurement=None)
    with pytest.raises(NotImplementedError, match="integer factor"):
        Resizer((1, 3, 64, 64), 1 / 2.5)
    assert Resizer((1, 3, 64, 64), 1 / 4).sf == 4 and Resizer((1, 3, 64, 64), 0.5).to("cuda").sf == 2
    # per-sample timesteps / 'epsilon' with vec_t: the reference reads an undefined t_step there (utils_model.py:248, 252)

    class D:
        sqrt_recip_alphas_cumprod = np.ones(1000)
        sqrt_recipm1_alphas_cumprod = np.ones(1000)
    ac = np.linspace(0.999, 0.001, 1000)
    with pytest.raises(NameError):
        utils_model.model_fn(None, 10.0, None, vec_t=np.array([5, 5]), model_out_type="epsilon", diffusion=D, alphas_cumprod=ac)
    with pytest.raises(NotImplementedError, match="per-sample"):
        utils_model.model_fn(None, 10.0, None, vec_t=np.array([5, 6]), model_out_type="pred_xstart", diffusion=D, alphas_cumprod=ac)


def test_yaml_driver_reads_the_reference_kernel_files(tmp_path):
    """diffpir_amd.main_ddpir.make_operators honours `use_DIY_kernel: false` (Levin09[0] out of <cwd>/kernels) and reads the sr PSF from
    kernels_bicubicx234.mat when the file is there; the arrays equal what the reference's dataset handed to test_rho."""
    from diffpir_amd import main_ddpir as drv
    g = np.load(os.path.join(ROOT, "tests", "golden", "refdata.npz"))
    kdir = tmp_path / "kernels"
    kdir.mkdir()
    import scipy.io
    cell = np.empty((1, 3), dtype=object)
    for i in range(3):
        cell[0, i] = g["c3bic_k"][0, 0].astype(np.float64) if i == 2 else np.zeros((25, 25))
    scipy.io.savemat(str(kdir / "kernels_bicubicx234.mat"), {"kernels": cell})
    np.savez(str(kdir / "Levin09.npz"), k0=g["c2lev_k"][0, 0].astype(np.float64), k1=np.zeros((17, 17)))
    # a v7.3 header (version word 0x0200, 'IM'): scipy refuses it ("Please use HDF reader") -> h5py is absent here -> the .npz sidecar is used
    (kdir / "Levin09.mat").write_bytes(b"MATLAB 7.3 MAT-file, Platform: test".ljust(124, b" ") + b"\x00\x02IM" + b"\0" * 512)
    cfg = drv.Config(dict(cwd=str(tmp_path), task="sr", sf=4))
    k, _ = drv.make_operators(cfg, 2, 0, 256, 256)
    assert k.shape == (2, 1, 25, 25) and np.array_equal(k[1, 0], g["c3bic_k"][0, 0])
    cfg = drv.Config(dict(cwd=str(tmp_path), task="deblur", use_DIY_kernel=False, blur_mode="Gaussian", kernel_size=61))
    k, _ = drv.make_operators(cfg, 3, 0, 256, 256)
    assert k.shape == (3, 1, 19, 19) and np.array_equal(k[2, 0], g["c2lev_k"][0, 0])
    with pytest.raises(FileNotFoundError):
        drv.make_operators(drv.Config(dict(cwd=str(tmp_path / "nowhere"), task="deblur", use_DIY_kernel=False)), 1, 0, 256, 256)


def test_yaml_driver_honours_load_mask(tmp_path):
    """`load_mask: true` + `mask_path` (main_ddpir.py:103-104): util.imread_uint(mask_path, n_channels=3).astype(bool), the SAME mask for every image.
    PNG (grey and RGB) through PIL, non-zero bytes keep the pixel; relative paths resolve under cwd; a size mismatch is an error."""
    from diffpir_amd import main_ddpir as drv
    from PIL import Image
    rng = np.random.default_rng(3)
    m = (rng.random((64, 64)) > 0.4).astype(np.uint8) * 255
    m[0, 0] = 1                                                # any non-zero byte is True after astype(bool)
    Image.fromarray(m).save(str(tmp_path / "grey.png"))
    rgb = np.stack([m, m, np.zeros_like(m)], axis=2)
    Image.fromarray(rgb).save(str(tmp_path / "rgb.png"))
    cfg = drv.Config(dict(cwd=str(tmp_path), task="inpaint", load_mask=True, mask_path="grey.png"))
    _, mask = drv.make_operators(cfg, 3, 0, 64, 64)
    assert mask.dtype == np.uint8 and mask.shape == (3, 3, 64, 64)
    assert np.array_equal(mask[2, 1], (m != 0).astype(np.uint8)) and np.array_equal(mask[0], mask[2])
    cfg = drv.Config(dict(cwd=str(tmp_path), task="inpaint", load_mask=True, mask_path=str(tmp_path / "rgb.png")))
    _, mask = drv.make_operators(cfg, 1, 0, 64, 64)
    assert np.array_equal(mask[0, 0], (m != 0).astype(np.uint8)) and not mask[0, 2].any()
    with pytest.raises(ValueError):
        drv.make_operators(cfg, 1, 0, 128, 128)
    with pytest.raises(ValueError):
        drv.make_operators(drv.Config(dict(cwd=str(tmp_path), task="inpaint", load_mask=True)), 1, 0, 64, 64)
