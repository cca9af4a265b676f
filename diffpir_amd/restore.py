"""One batch of the DiffPIR restoration loop (main_ddpir.py:259-470) on the HIP engine.

Two entry points with identical results:
  * restore_batch          -- the fast path: host builds the per-step scalar table once, then ONE call
                              into dpir_run_loop (optionally a cached hipGraph) does init -> N x
                              (UNet -> prox -> re-noise) -> finalize without returning to Python.
  * restore_batch_stepwise -- the reference's loop body, line for line, calling the drop-in plugs
                              (utils_model.model_fn, utils_sisr.pre_calculate/data_solution, ...).
                              Exists to show (and test) that the reference loop runs unmodified
                              against the engine's operator surface.
Noise: noise_source="host" draws N(0,1) on the host in the reference's order (SURVEY.md 8 a-R: init,
then per step p_sample / n1 / n2) through `noise_fn(shape) -> np.ndarray` and uploads what the loop
consumes; noise_source="device" uses the engine's Philox stream keyed by (seed, global image index,
draw index), which makes results independent of how a batch is sharded across GPUs.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np

from . import _lib
from .engine import Engine, DeviceArray, _ptr, EngineError
from .schedule import build_steps, find_nearest

TASKS = {"deblur": 0, "sr_blur": 1, "inpaint": 2, "sr_cubic": 3}


@dataclass
class LoopConfig:
    """The YAML keys that reach the loop (configs/*.yaml; derived fields main_ddpir.py:138-158)."""
    task: str = "deblur"               # deblur | sr | inpaint
    iter_num: int = 100
    noise_level_img: float = 12.75 / 255.0     # already /255 (main_ddpir.py:138)
    lambda_: float = 7.0
    zeta: float = 0.3
    eta: float = 0.0
    guidance_scale: float = 1.0
    sf: int = 1
    sr_mode: str = "blur"
    inIter: int = 1
    gamma: float = 0.01
    skip_type: str = "quad"
    num_train_timesteps: int = 1000
    beta_start: float = 0.0001
    beta_end: float = 0.02
    generate_mode: str = "DiffPIR"
    model_output_type: str = "pred_xstart"
    sub_1_analytic: bool = True
    ddim_sample: bool = False
    iter_num_U: int = 1
    noise_init_img: object = "max"      # 'max' -> t_start = T-1, else a noise level in /255 units (main_ddpir.py:197-200)
    skip_noise_model_t: bool = False    # main_ddpir.py:192-195, 391

    @property
    def sigma(self):
        return max(0.001, self.noise_level_img)

    def engine_task(self) -> int:
        if self.task == "deblur":
            return TASKS["deblur"]
        if self.task == "inpaint":
            return TASKS["inpaint"]
        if self.task == "sr":
            return TASKS["sr_blur"] if self.sr_mode == "blur" else TASKS["sr_cubic"]
        raise ValueError(f"unknown task {self.task}")

    def check_supported(self):
        # ddim_sample is accepted: with model_output_type=pred_xstart it selects the same x0 prediction and the same number
        # of RNG draws as p_sample (utils_model.py:219-240; tests/golden/model_fn.npz).  iter_num_U > 1 cannot be mirrored:
        # the reference raises IndexError on `seq[i+1]` at its last step (main_ddpir.py:448-451 with u < iter_num_U-1).
        if not self.sub_1_analytic and not (self.generate_mode == "DiffPIR" and self.task == "sr"):
            # main_ddpir.py:420-430: the first-order data step differentiates through degrade_op, which only exists for task sr
            # (Resizer) in a runnable form (deblur: SURVEY Q1; inpaint: "TODO first order solver for inpainting")
            raise NotImplementedError("sub_1_analytic=false is implemented for generate_mode DiffPIR, task 'sr'")
        if self.generate_mode not in GENERATE_MODES or self.model_output_type != "pred_xstart" or self.iter_num_U != 1:
            raise NotImplementedError("only generate_mode in (DiffPIR, repaint, vanilla), model_output_type=pred_xstart, "
                                      "iter_num_U=1 are on the accelerated path (SURVEY.md 8f); the gradient-based modes DPS_y0 / DPS_yt and "
                                      "sub_1_analytic=false exist for task sr")
        if self.generate_mode in ("DPS_y0", "DPS_yt"):
            # main_ddpir.py:370-373, 433-445.  Runnable in the reference for task 'sr' only: the deblurring operator raises at :302
            #  and the inpainting branch never defines xt.
            if self.task != "sr":
                raise NotImplementedError("generate_mode DPS_y0 / DPS_yt are implemented for task 'sr' (the only one the reference can run)")
        elif self.generate_mode != "DiffPIR" and self.task != "inpaint":
            # main_ddpir.py:448: outside DiffPIR mode x is only re-noised for inpainting; other tasks would leave x untouched
            raise NotImplementedError("generate_mode repaint / vanilla are inpainting modes in the reference")


GENERATE_MODES = {"DiffPIR": 0, "repaint": 1, "vanilla": 2, "DPS_y0": 3, "DPS_yt": 4}


def t_start_of(cfg: LoopConfig, reduced) -> int:
    """main_ddpir.py:197-200: the timestep the forward-noised initial image is placed at; steps above it are skipped (:346)."""
    if cfg.noise_init_img == "max":
        return cfg.num_train_timesteps - 1
    return find_nearest(reduced, 2 * float(cfg.noise_init_img) / 255)


def _steps(cfg: LoopConfig):
    T = cfg.num_train_timesteps
    t_start = None
    if cfg.noise_init_img != "max" or cfg.skip_noise_model_t:
        from .schedule import DriverTables
        red = DriverTables.make(cfg.beta_start, cfg.beta_end, T).reduced
        t_start = t_start_of(cfg, red)
        if cfg.skip_noise_model_t:
            # main_ddpir.py:391 compares the LOOP INDEX with T - noise_model_t; the branch it guards (a switch to pred_x_prev that
            # persists for the rest of the run, :407-413) is dead for every iter_num below that bound -- and not mirrored beyond it
            noise_model_t = find_nearest(red, 2 * cfg.noise_level_img)
            if cfg.iter_num > T - noise_model_t:
                raise NotImplementedError("skip_noise_model_t with iter_num > T - noise_model_t selects the reference's pred_x_prev "
                                          "fallback (main_ddpir.py:407-413), which is not on the accelerated path")
    return build_steps(iter_num=cfg.iter_num, sigma=cfg.sigma, lambda_=cfg.lambda_, zeta=cfg.zeta, eta=cfg.eta,
                       skip_type=cfg.skip_type, T=T, beta_start=cfg.beta_start, beta_end=cfg.beta_end, t_start=t_start,
                       generate_mode=cfg.generate_mode, model_output_type=cfg.model_output_type)


def draw_host_noise(noise_fn: Callable, steps, shape, need_n1: bool, repaint: bool = False):
    """Consume noise_fn in the reference's order; keep only what the loop uses.  Per step: [repaint mix], p_sample, then
    (unless last) the eta draw and the zeta draw (main_ddpir.py:355-358, gaussian_diffusion.py:430, main_ddpir.py:454-456)."""
    init = np.asarray(noise_fn(shape), dtype=np.float32)
    n_re = sum(1 for s in steps if not s["last"])
    n1 = np.empty((n_re,) + tuple(shape), np.float32) if need_n1 else None
    n2 = np.empty((n_re,) + tuple(shape), np.float32)
    nrp = np.empty((len(steps),) + tuple(shape), np.float32) if repaint else None
    j = 0
    for i, s in enumerate(steps):
        if repaint:
            nrp[i] = noise_fn(shape)
        noise_fn(shape)                                   # p_sample's randn_like (gaussian_diffusion.py:430), unused
        if not s["last"]:
            a = noise_fn(shape)
            if need_n1:
                n1[j] = a
            n2[j] = noise_fn(shape)
            j += 1
    return (init, n1, n2, nrp) if repaint else (init, n1, n2)


def restore_batch(engine: Engine, cfg: LoopConfig, y, k=None, mask=None, labels=None, noise_source="device",
                  noise_fn: Optional[Callable] = None, seed: int = 0, image_offset: int = 0, use_graph: bool = False,
                  skip_dead_final_eval: bool = False, out_f32=None, out_u8=None, return_u8: bool = False, _cache: dict = None,
                  predrawn=None):
    """y: [B,3,h,w] in [0,1]; k: [B,1,kh,kw]; mask: uint8 [B,3,H,W] -- device arrays (or numpy, uploaded).
    Returns a DeviceArray [B,3,H,W] = x_0 in [0,1] (un-clamped, main_ddpir.py:470), and the u8 NHWC
    array as well when return_u8."""
    cfg.check_supported()
    if cfg.generate_mode in ("DPS_y0", "DPS_yt"):
        if predrawn is not None or mask is not None:
            raise NotImplementedError("DPS modes take host noise through noise_fn (their draw order differs from the DiffPIR loop's) and no mask")
        return _restore_dps(engine, cfg, y, labels, noise_source, noise_fn, seed, image_offset, skip_dead_final_eval, out_f32, out_u8, return_u8)
    dt, steps, arr = _steps(cfg)
    keep = []

    def dev(a, dtype):
        if a is None:
            return None
        if isinstance(a, np.ndarray):
            d = engine.to_device(a, dtype)
            keep.append(d)
            return d
        return a
    y = dev(y, np.float32); k = dev(k, np.float32); mask = dev(mask, np.uint8)
    B, _, h, w = y.shape
    H, W = h * cfg.sf, w * cfg.sf
    d = _lib.LoopDesc()
    d.task = cfg.engine_task()
    d.B, d.H, d.W, d.sf = B, H, W, cfg.sf
    if k is not None:
        d.kh, d.kw = k.shape[2], k.shape[3]
    d.in_iter, d.gamma, d.guidance = cfg.inIter, cfg.gamma, cfg.guidance_scale
    t_start = t_start_of(cfg, dt.reduced)
    d.sa_start, d.s1m_start = float(dt.sqrt_ac[t_start]), float(dt.sqrt_1m_ac[t_start])
    d.y_dev, d.k_dev, d.mask_dev = _ptr(y), _ptr(k), _ptr(mask)
    lab = None
    if labels is not None:
        lab = np.ascontiguousarray(labels, dtype=np.int64)
        d.labels_host = lab.ctypes.data
    if noise_source == "host":
        rp = cfg.generate_mode == "repaint"
        if predrawn is not None:                 # (init, n1, n2[, nrp]) already drawn in the reference's order (sharded runs)
            drawn = predrawn
        elif noise_fn is None:
            raise EngineError("noise_source='host' needs noise_fn")
        else:
            drawn = draw_host_noise(noise_fn, steps, (B, 3, H, W), cfg.eta != 0, repaint=rp)
        init, n1, n2 = drawn[:3]
        if rp:
            drp = engine.to_device(drawn[3])
            keep.append(drp)
            d.noise_rp_dev = drp.ptr
        di, d2 = engine.to_device(init), engine.to_device(n2)
        keep += [di, d2]
        d.noise_init_dev, d.noise_n2_dev = di.ptr, d2.ptr
        if n1 is not None:
            d1 = engine.to_device(n1)
            keep.append(d1)
            d.noise_n1_dev = d1.ptr
    elif noise_source != "device":
        raise ValueError("noise_source must be 'host' or 'device'")
    d.seed, d.image_offset = seed, image_offset
    d.use_graph, d.skip_dead_final_eval = int(use_graph), int(skip_dead_final_eval)
    d.generate_mode = GENERATE_MODES[cfg.generate_mode]
    d.first_order = 0 if cfg.sub_1_analytic else 1
    if out_f32 is None:
        out_f32 = engine.empty((B, 3, H, W))
    if out_u8 is None and return_u8:
        out_u8 = engine.empty((B, H, W, 3), np.uint8)
    engine._check(engine.lib.dpir_run_loop(engine.h, C.byref(d), arr, len(steps), _ptr(out_f32), _ptr(out_u8)))
    if _cache is not None:
        _cache["keep"] = keep
    else:
        engine.sync()
    return (out_f32, out_u8) if return_u8 else out_f32


def dps_host_noise_shapes(cfg: LoopConfig, steps, B: int, H: int, W: int):
    """Shapes of the reference's randn_like draws in the DPS modes, in call order: init, then per step the sampler's draw
    (gaussian_diffusion.py:430 / :577, every step incl. the dead final one) and, DPS_yt on non-final steps, the y_t draw (main_ddpir.py:440)."""
    h, w = H // cfg.sf, W // cfg.sf
    shapes = [(B, 3, H, W)]
    for st in steps:
        shapes.append((B, 3, H, W))
        if cfg.generate_mode == "DPS_yt" and not st["last"]:
            shapes.append((B, 3, h, w))
    return shapes


def _restore_dps(engine, cfg, y, labels, noise_source, noise_fn, seed, image_offset, skip_dead_final_eval, out_f32, out_u8, return_u8):
    """generate_mode 'DPS_y0' / 'DPS_yt' (main_ddpir.py:370-373, 433-445): dpir_run_dps_loop.  Host noise order: init, then per step the
    p_sample draw and (DPS_yt, non-final steps) the y_t draw; there is no re-noising in these modes (main_ddpir.py:448)."""
    from .schedule import DiffusionTables
    yt_mode = cfg.generate_mode == "DPS_yt"
    if not yt_mode and not engine.grad:
        raise EngineError("generate_mode DPS_y0 needs Engine.enable_grad() before the model is loaded")
    dt, steps, arr = _steps(cfg)
    dtab = DiffusionTables.make(cfg.num_train_timesteps)
    coefs = (_lib.DpsCoef * len(steps))()
    for i, st in enumerate(steps):
        coefs[i].pc1, coefs[i].pc2, coefs[i].min_log, coefs[i].max_log = [float(v) for v in dtab.dps_coef(st["t"])]
        coefs[i].sa_prev, coefs[i].s1m_prev = [float(v) for v in dtab.ddim_coef(st["t"])]
    yd = engine.to_device(y, np.float32) if isinstance(y, np.ndarray) else y
    B, _, h, w = yd.shape
    H, W = h * cfg.sf, w * cfg.sf
    d = _lib.LoopDesc()
    d.task = cfg.engine_task()
    d.B, d.H, d.W, d.sf = B, H, W, cfg.sf
    t_start = t_start_of(cfg, dt.reduced)
    d.sa_start, d.s1m_start = float(dt.sqrt_ac[t_start]), float(dt.sqrt_1m_ac[t_start])
    d.y_dev = _ptr(yd)
    lab = None
    if labels is not None:
        lab = np.ascontiguousarray(labels, dtype=np.int64)
        d.labels_host = lab.ctypes.data
    keep = [yd]
    nps = nyt = None
    if noise_source == "host":
        if noise_fn is None:
            raise EngineError("noise_source='host' needs noise_fn")
        init = np.asarray(noise_fn((B, 3, H, W)), dtype=np.float32)
        ps, yt = [], []
        for st in steps:
            ps.append(np.asarray(noise_fn((B, 3, H, W)), dtype=np.float32))
            yt.append(np.asarray(noise_fn((B, 3, h, w)), dtype=np.float32) if yt_mode and not st["last"] else np.zeros((B, 3, h, w), np.float32))
        di, nps = engine.to_device(init), engine.to_device(np.stack(ps))
        keep += [di, nps]
        if yt_mode:
            nyt = engine.to_device(np.stack(yt))
            keep.append(nyt)
        d.noise_init_dev = di.ptr
    elif noise_source != "device":
        raise ValueError("noise_source must be 'host' or 'device'")
    d.seed, d.image_offset = seed, image_offset
    d.skip_dead_final_eval = int(skip_dead_final_eval)
    d.generate_mode = GENERATE_MODES[cfg.generate_mode]
    # config.ddim_sample reaches model_fn(..., 'pred_x_prev_and_start') (main_ddpir.py:371-373): xt is then ddim_sample(eta=0)'s sample
    d.ddim_sample = 1 if cfg.ddim_sample else 0
    if out_f32 is None:
        out_f32 = engine.empty((B, 3, H, W))
    if out_u8 is None and return_u8:
        out_u8 = engine.empty((B, H, W, 3), np.uint8)
    engine._check(engine.lib.dpir_run_dps_loop(engine.h, C.byref(d), arr, coefs, len(steps), 1 if yt_mode else 0, float(cfg.lambda_),
                                               _ptr(nps), _ptr(nyt), 1.0, _ptr(out_f32), _ptr(out_u8)))
    engine.sync()
    return (out_f32, out_u8) if return_u8 else out_f32


def restore_batch_stepwise(model, diffusion, cfg: LoopConfig, y, k=None, mask=None, noise_fn: Callable = None, labels=None):
    """The reference's loop body (main_ddpir.py:291-470) against the drop-in plugs, host-fed noise.  Every branch the monolithic
    loops implement: the DiffPIR analytic step, the first-order data step (sub_1_analytic: false), DPS_y0 and DPS_yt -- the latter
    three written with the reference's own expressions on device arrays (`xt - norm_grad * 1.`, ...)."""
    from . import utils_model, utils_sisr as sr
    from .utils_resizer import Resizer
    eng: Engine = model.engine
    cfg.check_supported()
    dt, steps, arr = _steps(cfg)
    B, _, h, w = y.shape
    H, W = h * cfg.sf, w * cfg.sf
    shape = (B, 3, H, W)
    lib, hnd = eng.lib, eng.h
    x = eng.empty(shape)
    dps = "DPS" in cfg.generate_mode
    # (3) initialize x (main_ddpir.py:293-315)
    if cfg.task == "sr":
        degrade_op = Resizer(shape, 1 / cfg.sf, engine=eng)
        src = eng.empty(shape)
        eng._check(lib.dpir_bicubic_up(hnd, _ptr(y), src.ptr, cfg.sf, B, h, w))
        xs = src.numpy()
    elif cfg.task == "deblur":
        xs = y.numpy()
    else:
        xs = y.numpy() * mask.numpy().astype(np.float32)
    t_start = t_start_of(cfg, dt.reduced)
    n0 = np.asarray(noise_fn(shape), np.float32)
    x.copy_from(dt.sqrt_ac[t_start] * (np.float32(2) * xs - np.float32(1)) + dt.sqrt_1m_ac[t_start] * n0)
    if cfg.task in ("sr", "deblur") and not (cfg.task == "sr" and cfg.sr_mode == "cubic") and not dps and cfg.sub_1_analytic:
        FB, FBC, F2B, FBFy = sr.pre_calculate(y, k, cfg.sf, engine=eng)
    kwargs = {} if labels is None else {"y": labels}
    # the sampler's randn_like (gaussian_diffusion.py:430 / :577) and the loop's own draws come from ONE host stream, in call order
    draw = lambda like: eng.to_device(np.asarray(noise_fn(like.shape), np.float32))
    utils_model.set_randn_like(draw)
    try:
        for i, st in enumerate(steps):
            curr_sigma = dt.reduced[st["t"]]
            t_i = st["t"]
            if cfg.generate_mode == "repaint":                  # main_ddpir.py:355-358
                nr = eng.to_device(np.asarray(noise_fn(shape), np.float32))
                eng._check(lib.dpir_repaint_mix(hnd, x.ptr, _ptr(y), _ptr(mask), C.byref(arr[i]), nr.ptr, B, H, W))
            if dps:                                             # main_ddpir.py:370-373
                x = x.requires_grad_()
                xt, x0 = utils_model.model_fn(x, noise_level=curr_sigma * 255, model_out_type="pred_x_prev_and_start", model_diffusion=model,
                                              diffusion=diffusion, ddim_sample=cfg.ddim_sample, alphas_cumprod=dt.alphas_cumprod, **kwargs)
            else:
                x0 = utils_model.model_fn(x, noise_level=curr_sigma * 255, model_out_type="pred_xstart", model_diffusion=model,
                                          diffusion=diffusion, ddim_sample=cfg.ddim_sample, alphas_cumprod=dt.alphas_cumprod, **kwargs)
                noise_fn(shape)                                 # p_sample's draw (unused for pred_xstart)
            if st["last"]:
                continue
            tau = np.float32(st["tau"])                         # rhos[t_i]
            if cfg.generate_mode == "DPS_y0":                   # main_ddpir.py:434-438
                measurement = 2 * y - 1
                norm_grad, norm = utils_model.grad_and_value(operator=degrade_op, x=x, x_hat=x0, measurement=measurement)
                x = xt - norm_grad * 1.
                x = x.detach_()
                continue
            if cfg.generate_mode == "DPS_yt":                   # main_ddpir.py:439-445
                y_t = dt.sqrt_ac[t_i] * (2 * y - 1) + dt.sqrt_1m_ac[t_i] * draw(y)
                measurement = y_t
                norm_grad, norm = utils_model.grad_and_value(operator=degrade_op, x=xt, x_hat=xt, measurement=measurement)
                x = xt - norm_grad * cfg.lambda_ * norm / tau * 0.35
                x = x.detach_()
                continue
            if cfg.generate_mode != "DiffPIR":
                pass                                            # no data-fidelity step outside DiffPIR mode (main_ddpir.py:385)
            elif not cfg.sub_1_analytic:                        # main_ddpir.py:420-430 (task sr)
                x0 = x0.requires_grad_()
                measurement = 2 * y - 1
                norm_grad, norm = utils_model.grad_and_value(operator=degrade_op, x=x0, x_hat=x0, measurement=measurement)
                x0 = x0 - norm_grad * norm / tau
                x0 = x0.detach_()
            elif cfg.task == "inpaint":
                eng._check(lib.dpir_prox_mask(hnd, x0.ptr, _ptr(y), _ptr(mask), float(tau), cfg.guidance_scale, B, H, W))
            elif cfg.task == "deblur" or cfg.sr_mode == "blur":
                x0_p = eng.empty(shape)
                eng._check(lib.dpir_finalize(hnd, x0.ptr, x0_p.ptr, None, B, H, W))          # x0/2+.5
                x0_p = sr.data_solution(x0_p, FB, FBC, F2B, FBFy, tau, cfg.sf)
                a, b = x0.numpy(), x0_p.numpy() * np.float32(2) - np.float32(1)
                x0.copy_from(a + np.float32(cfg.guidance_scale) * (b - a))
            else:
                eng._check(lib.dpir_prox_ibp(hnd, x0.ptr, _ptr(y), float(tau), cfg.gamma, cfg.inIter, cfg.sf, B, H, W))
            n1 = eng.to_device(np.asarray(noise_fn(shape), np.float32))
            n2 = eng.to_device(np.asarray(noise_fn(shape), np.float32))
            eng._check(lib.dpir_renoise(hnd, x.ptr, x0.ptr, C.byref(arr[i]), n1.ptr, n2.ptr, B, H, W))
    finally:
        utils_model.set_randn_like(None)
    out = eng.empty(shape)
    eng._check(lib.dpir_finalize(hnd, x.ptr, out.ptr, None, B, H, W))
    eng.sync()
    return out


def psnr_batch(a: np.ndarray, b: np.ndarray, max_pixel=2.0, eps=1e-10) -> float:
    """utils/utils_image.py:601-610 on numpy arrays in [-1,1]."""
    mse = np.mean((a.astype(np.float32) - b.astype(np.float32)) ** 2, axis=(1, 2, 3), dtype=np.float32)
    with np.errstate(divide="ignore"):
        v = np.where(mse == 0, np.inf, 20 * np.log10(max_pixel / np.sqrt(mse + np.float32(eps))))
    v = np.where(np.isnan(v), 0.0, v)
    return float(np.mean(v))
