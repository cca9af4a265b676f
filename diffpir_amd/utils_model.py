"""Denoiser plug with the reference's signature (utils/utils_model.py:202-258, 353-387).

    x0 = utils_model.model_fn(x, noise_level=curr_sigma*255, model_out_type='pred_xstart',
                              model_diffusion=model, diffusion=diffusion, ddim_sample=False,
                              alphas_cumprod=alphas_cumprod)

`x` is a device array (diffpir_amd.engine.DeviceArray, or a contiguous torch HIP tensor used as
plumbing); the result is a DeviceArray.  The p_sample randn_like draw of the reference
(gaussian_diffusion.py:430) feeds only the discarded 'sample' output; callers that need the
reference's host RNG stream position advance it themselves (see diffpir_amd/restore.py).
"""
from __future__ import annotations

import argparse

import numpy as np

from .schedule import find_nearest
from .script_util import add_dict_to_argparser


MODEL_OUT_TYPES = ("pred_xstart", "pred_x_prev_and_start", "pred_x_prev", "epsilon", "score")

# The sampler's `th.randn_like(x)` (gaussian_diffusion.py:430, 577).  'pred_xstart' never uses the draw (callers that follow the
# reference's host RNG stream advance it themselves, restore.py); the other output types do.  Default: the engine's Philox stream with
# one stream id per call; parity runs install a host-fed draw with `set_randn_like(fn)`, fn(x: DeviceArray) -> DeviceArray.
_randn = {"fn": None, "calls": 0, "seed": 0}


def set_randn_like(fn=None, seed: int = 0):
    _randn.update(fn=fn, calls=0, seed=int(seed))


def randn_like(x):
    if _randn["fn"] is not None:
        return _randn["fn"](x)
    eng = x.engine
    out = eng.empty(x.shape)
    B, Cc, H, W = x.shape
    _randn["calls"] += 1
    eng._check(eng.lib.dpir_randn(eng.h, out.ptr, _randn["seed"], 1_000_000 + _randn["calls"], 0, B, Cc, H, W))
    return out


def model_fn(x, noise_level, model_diffusion, vec_t=None, model_out_type='pred_xstart',
             diffusion=None, ddim_sample=False, alphas_cumprod=None, **model_kwargs):
    """utils/utils_model.py:207-258.  'pred_xstart' is the DiffPIR analytic path (one fused call); 'pred_x_prev_and_start' (the DPS
    branch, main_ddpir.py:370-373), 'pred_x_prev', 'epsilon' and 'score' go through dpir_p_sample: p_sample with the learned-range
    variance, or ddim_sample(eta=0) when ddim_sample is set."""
    if model_out_type not in MODEL_OUT_TYPES:
        raise ValueError(f"model_out_type={model_out_type!r}: expected one of {MODEL_OUT_TYPES}")
    ac = np.asarray(alphas_cumprod, dtype=np.float32)
    sqrt_ac = np.sqrt(ac, dtype=np.float32)
    sqrt_1m = np.sqrt((np.float32(1.0) - ac).astype(np.float32), dtype=np.float32)
    reduced = (sqrt_1m / sqrt_ac).astype(np.float32)
    if vec_t is None:
        t_step = find_nearest(reduced, noise_level / 255.)
    else:
        vt = np.asarray(vec_t).reshape(-1)
        if not (vt == vt[0]).all():
            raise NotImplementedError("per-sample timesteps: use HipUNetModel.__call__ directly")
        if model_out_type in ("epsilon", "score"):
            raise NameError("t_step is not defined when vec_t is given (utils/utils_model.py:248, 252 read it)")     # as the reference
        t_step = int(vt[0])
    c1 = np.float32(diffusion.sqrt_recip_alphas_cumprod[t_step])
    c2 = np.float32(diffusion.sqrt_recipm1_alphas_cumprod[t_step])
    eng = model_diffusion.engine
    y = model_kwargs.get("y")
    if model_out_type == "pred_xstart":
        # ddim_sample=True (utils_model.py:230-240 -> gaussian_diffusion.py:537-585, eta=0): 'pred_xstart' is the very same
        # p_mean_variance output as in p_sample and one randn_like is consumed either way (tests/golden/model_fn.npz)
        return eng.model_fn_xstart(x, t_step, c1, c2, y)
    from . import _lib
    tab = diffusion.tables
    cf = _lib.PSampleCoef()
    cf.c1, cf.c2 = float(c1), float(c2)
    cf.pc1, cf.pc2, cf.min_log, cf.max_log = [float(v) for v in tab.dps_coef(t_step)]
    cf.ddim = 1 if ddim_sample else 0
    cf.sa_prev, cf.s1m_prev = [float(v) for v in tab.ddim_coef(t_step)]
    xt, x0 = eng.p_sample(x, t_step, cf, randn_like(x), y)
    if model_out_type == "pred_x_prev_and_start":
        return xt, x0
    if model_out_type == "pred_x_prev":
        return xt
    # 'epsilon' / 'score' (utils_model.py:247-255): alphas_cumprod[t] ** 0.5 and (1 - alphas_cumprod[t]) ** 0.5 on the driver's float32 table
    import torch
    a_t = torch.as_tensor(ac)[int(t_step)]
    sa, s1m = float(a_t ** 0.5), float((1 - a_t) ** 0.5)
    out = eng.empty(x.shape)
    eng._check(eng.lib.dpir_eps_from_xstart(eng.h, x.ptr, x0.ptr, sa, s1m, 1 if model_out_type == "score" else 0, out.ptr, out.size))
    return out


def grad_and_value(operator, x, x_hat, measurement):
    """utils/utils_model.py:390-394 -- (d ||measurement - operator(x_hat)|| / d x, the norm) -- for the operator the reference can run it
    with: utils_resizer.Resizer (task sr, main_ddpir.py:294).  `x is x_hat`: the gradient w.r.t. the operator's own argument
    (first-order data step :425, DPS_yt :443).  Otherwise x must be the input and x_hat the pred_xstart output of the LAST
    model_fn(..., 'pred_x_prev_and_start') call (DPS_y0 :436): the gradient runs through the clamp and the denoiser's tape."""
    from .utils_resizer import Resizer
    if not isinstance(operator, Resizer):
        raise NotImplementedError("grad_and_value: the engine differentiates through diffpir_amd.utils_resizer.Resizer (the reference's deblurring "
                                  "degrade_op raises at main_ddpir.py:302 and inpainting has none)")
    eng = x_hat.engine
    return eng.grad_and_value(x is not x_hat, x_hat, measurement, operator.sf)


def create_argparser(model_config):
    """utils/utils_model.py:353-387: DiffPIR's model hyper-parameter defaults."""
    defaults = dict(clip_denoised=True, num_samples=1, batch_size=1, use_ddim=False, model_path='', diffusion_steps=1000,
                    noise_schedule='linear', num_head_channels=64, resblock_updown=True, use_fp16=False,
                    use_scale_shift_norm=True, num_heads=4, num_heads_upsample=-1, use_new_attention_order=False,
                    timestep_respacing="", use_kl=False, predict_xstart=False, rescale_timesteps=False,
                    rescale_learned_sigmas=False, channel_mult="", learn_sigma=True, class_cond=False,
                    use_checkpoint=False, image_size=256, num_channels=128, num_res_blocks=1,
                    attention_resolutions="16", dropout=0.1)
    defaults.update(model_config)
    parser = argparse.ArgumentParser()
    add_dict_to_argparser(parser, defaults)
    return parser
