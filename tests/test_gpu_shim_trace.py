"""-m gpu: the reference's OWN loop body against the engine's plug mirrors, by call trace (round-5 review item 5).

tests/golden/shim_trace.npz (oracle/gen_golden_shim_trace.py, build container) holds, for four runs of the reference's unmodified `test_rho`
(main_ddpir.py:341-470; deblur, box inpainting, sr x4 with the FFT prox, sr x4 DPS_y0), every call the loop body made to its five plugs --
`utils_model.model_fn`, `utils_model.grad_and_value`, `sr.pre_calculate`, `sr.data_solution`, `Resizer` -- with the arguments it passed and what
the reference's plug returned.  Here each recorded call is issued, in order, to diffpir_amd's mirror of that plug with the recorded arguments and the
return is compared with the recorded one: INTEGRATION.md section A ("bind these names, the loop body runs unchanged") exercised with the arguments
the real loop body produces, not with a restated body.  Then the run's final x_0 is compared with dpir_run_loop on the same inputs and noise."""
import json
import os

import numpy as np
import pytest
import torch

import diffpir_amd
from diffpir_amd import restore, schedule, script_util, utils_model as um, utils_sisr as sr
from diffpir_amd.utils_resizer import Resizer
from oracle import unet_oracle as uo, diffpir_oracle as do
from tests.gpu_common import make_model, fft_prox_parity, oracle_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def setup():
    e = diffpir_amd.Engine(0)
    e.set_precision("f16x3")
    e.enable_grad()
    model, sd = make_model(e, uo.tiny_hp())
    diffusion = script_util.create_gaussian_diffusion(steps=1000, learn_sigma=True)
    g = np.load(os.path.join(ROOT, "tests", "golden", "shim_trace.npz"))
    yield e, model, sd, diffusion, g
    e.close()


def _tol_prox(ref, x, y, k, sf, alpha):
    """|reference - exact| of this data_solution call: the reference's own fp32 noise (its closed form divides a near-cancelling difference by alpha,
    DESIGN.md section 4), measured with the oracle's float64 evaluation of the same expression on the same inputs."""
    pre64 = do.pre_calculate(torch.from_numpy(y).double(), torch.from_numpy(k).double(), sf)
    ex = do.data_solution(torch.from_numpy(x).double(), *pre64, torch.tensor(alpha).double().repeat(1, 1, 1, 1), sf).numpy()
    return float(np.abs(ref - ex).max())


@pytest.mark.parametrize("case", ["deblur", "inpaint", "sr_blur", "dps_y0"])
def test_reference_call_trace_through_the_engine_plugs(setup, case):
    e, model, sd, diffusion, g = setup
    meta = json.loads(bytes(g[f"{case}/meta"]).decode())
    cfgd, calls = meta["cfg"], meta["calls"]
    arr = lambda key: g[f"{case}/{key}"]
    dt = schedule.DriverTables.make(0.1 / 1000, 20 / 1000, 1000)
    gen = torch.Generator().manual_seed(cfgd["seed"])
    draw = lambda shape: torch.randn(tuple(shape), generator=gen, dtype=torch.float32).numpy()
    y, k = arr("y"), (arr("k") if f"{case}/k" in g.files else None)
    last_draw, pre, op, last_fn = None, None, None, None
    seen = {}
    for c in calls:
        fn = c["fn"]
        seen[fn] = seen.get(fn, 0) + 1
        if fn == "randn_like":
            last_draw = draw(c["shape"])                      # the stream position of the reference run; p_sample's draw is the last one before model_fn
            continue
        if fn == "model_fn":
            x = e.to_device(arr(c["x"]))
            um.set_randn_like(lambda xx, d=last_draw: xx.engine.to_device(d))
            out = um.model_fn(x, noise_level=c["noise_level"], model_diffusion=model, model_out_type=c["model_out_type"], diffusion=diffusion,
                              ddim_sample=c["ddim_sample"], alphas_cumprod=dt.alphas_cumprod)
            um.set_randn_like(None)
            outs = out if isinstance(out, tuple) else (out,)
            for i, o in enumerate(outs):
                ref = arr(c[f"out{i}"])
                err = float(np.abs(o.numpy() - ref).max())
                assert err < 2e-4 * max(1.0, float(np.abs(ref).max())), (case, "model_fn", c["model_out_type"], i, err)
            last_fn = (x, outs)
        elif fn == "pre_calculate":
            pre = sr.pre_calculate(e.to_device(arr(c["y"])), e.to_device(arr(c["k"])), c["sf"])
            np.testing.assert_allclose(pre[0].numpy(), arr(c["FB"]), atol=2e-6)
            np.testing.assert_allclose(pre[2].numpy(), arr(c["F2B"]), atol=2e-6)
            fy = arr(c["FBFy"])
            np.testing.assert_allclose(pre[3].numpy(), fy, atol=1e-6 * float(np.abs(fy).max()), rtol=1e-5)
        elif fn == "data_solution":
            xin, ref = arr(c["x"]), arr(c["out"])
            out = sr.data_solution(e.to_device(xin), *pre, np.float32(c["alpha"]), c["sf"]).numpy()
            floor = _tol_prox(ref, xin, y, k, c["sf"], c["alpha"])
            err = float(np.abs(out - ref).max())
            assert err <= 1.5 * floor + 3e-5, (case, "data_solution", c["alpha"], err, floor)
        elif fn == "Resizer":
            op = Resizer(tuple(c["in_shape"]), 1.0 / c["sf"], engine=e)
        elif fn == "Resizer_forward":
            out = op(e.to_device(arr(c["x"]))).numpy()
            np.testing.assert_allclose(out, arr(c["out"]), atol=2e-6)
        elif fn == "grad_and_value":
            x_dev, outs = last_fn                              # DPS_y0: x is the input of the last model_fn call, x_hat its pred_xstart (main_ddpir.py:433-436)
            assert not c["x_is_x_hat"] and np.array_equal(arr(c["x"]), x_dev.numpy())
            gn, nv = um.grad_and_value(operator=op, x=x_dev, x_hat=outs[1], measurement=e.to_device(arr(c["measurement"])))
            ref_g, ref_n = arr(c["norm_grad"]), float(arr(c["norm"]))
            assert abs(float(nv.numpy()[0]) - ref_n) <= 2e-5 * ref_n, (case, "norm", float(nv.numpy()[0]), ref_n)
            err = float(np.abs(gn.numpy() - ref_g).max())
            assert err < 2e-4 * float(np.abs(ref_g).max()), (case, "norm_grad", err, float(np.abs(ref_g).max()))
        else:
            raise AssertionError(f"unknown plug in the trace: {fn}")
    print(f"{case}: replayed {sum(v for kk, v in seen.items() if kk != 'randn_like')} plug calls of the reference's loop body {seen}")
    # the whole run: dpir_run_loop on the same y / k / mask with the same noise stream against the reference's x_0
    kw = dict(task=cfgd["task"], iter_num=cfgd["iter_num"], lambda_=cfgd["lambda_"], zeta=cfgd["zeta"], sf=cfgd["sf"], noise_level_img=cfgd["noise_level_img"])
    if cfgd["task"] == "sr":
        kw.update(sr_mode=cfgd["sr_mode"], generate_mode=cfgd["generate_mode"])
    cfg = restore.LoopConfig(**kw)
    g2 = torch.Generator().manual_seed(cfgd["seed"])
    noise = lambda shape: torch.randn(tuple(shape), generator=g2, dtype=torch.float32).numpy()
    mask = arr("mask").astype(np.uint8) if f"{case}/mask" in g.files else None
    out = restore.restore_batch(e, cfg, y, k=None if cfg.generate_mode != "DiffPIR" or cfgd["task"] == "inpaint" else k, mask=mask, noise_source="host",
                                noise_fn=noise).numpy()
    ref = arr("x0")
    if case in ("deblur", "sr_blur"):
        ocfg = do.LoopConfig(cfgd["task"], cfgd["iter_num"], cfgd["noise_level_img"], cfgd["lambda_"], cfgd["zeta"], sf=cfgd["sf"])
        _, exact = oracle_pair("shim_" + case, sd, uo.tiny_hp(), ocfg, y, k, cfgd["seed"])
        gt = np.zeros_like(ref)
        fft_prox_parity(out, ref, gt, f"shim trace {case}: dpir_run_loop vs the traced reference run", exact=exact)
    else:
        err = float(np.abs(out - ref).max())
        print(f"{case}: dpir_run_loop vs the traced reference run: max|diff| {err:.3e}")
        assert err < 2e-3 * max(1.0, float(np.abs(ref).max()))
