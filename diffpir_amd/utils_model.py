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


def model_fn(x, noise_level, model_diffusion, vec_t=None, model_out_type='pred_xstart',
             diffusion=None, ddim_sample=False, alphas_cumprod=None, **model_kwargs):
    # ddim_sample=True (utils_model.py:230-240 -> gaussian_diffusion.py:537-585, eta=0): 'pred_xstart' is the very same
    # p_mean_variance output as in p_sample and one randn_like is consumed either way, so the flag does not change this
    # path (pinned against the live reference in tests/golden/model_fn.npz).
    if model_out_type not in ("pred_xstart", "epsilon"):
        raise NotImplementedError(f"model_out_type={model_out_type!r}: only the DiffPIR analytic path "
                                  f"('pred_xstart') is accelerated")
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
        t_step = int(vt[0])
    c1 = np.float32(diffusion.sqrt_recip_alphas_cumprod[t_step])
    c2 = np.float32(diffusion.sqrt_recipm1_alphas_cumprod[t_step])
    eng = model_diffusion.engine
    y = model_kwargs.get("y")
    x0 = eng.model_fn_xstart(x, t_step, c1, c2, y)
    if model_out_type == "pred_xstart":
        return x0
    raise NotImplementedError("model_out_type='epsilon' needs an extra elementwise kernel; not on the DiffPIR path")


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
