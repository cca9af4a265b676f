"""Model / diffusion factory with the reference's surface.

Mirrors guided_diffusion/script_util.py:43-65 (defaults), :130-184 (create_model), :386-424
(create_gaussian_diffusion) and utils/utils_model.py:353-387 (create_argparser defaults), so the
driver lines of main_ddpir.py:219-240

    args = utils_model.create_argparser(model_config).parse_args([])
    model, diffusion = create_model_and_diffusion(**args_to_dict(args, model_and_diffusion_defaults().keys()))
    model.load_state_dict(torch.load(args.model_path, map_location="cpu"))
    model.eval(); model = model.to(device)

run unchanged against the HIP engine.  Only what the sampling path uses is supported; anything else
(training, fp16, new attention order, learned sigma off, respacing) raises.
"""
from __future__ import annotations

import argparse
from typing import Dict, Optional

import numpy as np

from . import _lib
from .engine import Engine, default_engine
from .schedule import DiffusionTables

NUM_CLASSES = 1000


def model_and_diffusion_defaults():
    """script_util.py:43-65 merged with :13-24 (same keys)."""
    return dict(image_size=64, num_channels=128, num_res_blocks=2, num_heads=4, num_heads_upsample=-1,
                num_head_channels=-1, attention_resolutions="16,8", channel_mult="", dropout=0.0,
                class_cond=False, use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=False,
                use_fp16=False, use_new_attention_order=False,
                learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="",
                use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}


class HipUNetModel:
    """Stands in for guided_diffusion.unet.UNetModel on the sampling path (forward only)."""

    def __init__(self, desc: "_lib.UNetDesc", engine: Optional[Engine] = None):
        self.desc = desc
        self._engine = engine
        self.loaded = False
        self.num_classes = desc.num_classes if desc.num_classes > 0 else None

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = default_engine(0)
        return self._engine

    def load_state_dict(self, state_dict: Dict[str, "np.ndarray"], strict: bool = True):
        self.engine.load_unet(self.desc, state_dict)
        self.engine.out_channels = self.desc.out_channels
        self.loaded = True
        return self

    # torch.nn.Module surface used by main_ddpir.py:235-240
    def eval(self):
        return self

    def to(self, device=None):
        return self

    def named_parameters(self):
        return iter(())

    def __call__(self, x, timesteps, y=None):
        """UNetModel.forward (unet.py:634-663): x [B,3,H,W] device array, timesteps [B] ints."""
        if (y is not None) != (self.num_classes is not None):
            raise AssertionError("must specify y if and only if the model is class-conditional")
        return self.engine.unet_forward(x, np.asarray(timesteps, dtype=np.int64), y)


class HipGaussianDiffusion:
    """The slice of SpacedDiffusion/GaussianDiffusion the sampling path touches: the float64 tables
    behind _predict_xstart_from_eps (gaussian_diffusion.py:133-151, 328-333) with the identity timestep
    map of timestep_respacing="" (respace.py:63-128)."""

    def __init__(self, steps=1000):
        self.num_timesteps = steps
        self.tables = DiffusionTables.make(steps)
        self.sqrt_recip_alphas_cumprod = self.tables.sqrt_recip_ac
        self.sqrt_recipm1_alphas_cumprod = self.tables.sqrt_recipm1_ac


def create_model(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False, class_cond=False,
                 use_checkpoint=False, attention_resolutions="16", num_heads=1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0, resblock_updown=False,
                 use_fp16=False, use_new_attention_order=False, engine: Optional[Engine] = None,
                 num_classes: int = NUM_CLASSES):
    """script_util.py:130-184."""
    if not (use_scale_shift_norm and resblock_updown):
        raise NotImplementedError("the engine implements the DiffPIR configuration: use_scale_shift_norm=True, resblock_updown=True")
    if use_fp16 or use_new_attention_order:
        raise NotImplementedError("use_fp16 / use_new_attention_order are not used by DiffPIR (utils_model.py:364,368)")
    if not learn_sigma:
        raise NotImplementedError("learn_sigma=False checkpoints are not used by DiffPIR")
    if num_head_channels != 64:
        raise NotImplementedError("num_head_channels must be 64 (utils_model.py:361)")
    d = _lib.UNetDesc()
    d.image_size = int(image_size)
    d.in_channels = 3
    d.model_channels = int(num_channels)
    d.out_channels = 6 if learn_sigma else 3
    d.num_res_blocks = int(num_res_blocks)
    d.num_head_channels = int(num_head_channels)
    if channel_mult == "":
        d.n_channel_mult = 0
        if image_size not in (512, 256, 128, 64):
            raise ValueError(f"unsupported image size: {image_size}")
    else:
        cm = [float(c) for c in channel_mult.split(",")] if isinstance(channel_mult, str) else [float(c) for c in channel_mult]
        d.n_channel_mult = len(cm)
        for i, c in enumerate(cm):
            d.channel_mult[i] = c
    ads = [image_size // int(r) for r in attention_resolutions.split(",")]
    d.n_attention_ds = len(ads)
    for i, a in enumerate(ads):
        d.attention_ds[i] = a
    d.num_classes = num_classes if class_cond else 0
    return HipUNetModel(d, engine)


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear", use_kl=False,
                              predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False,
                              timestep_respacing=""):
    """script_util.py:386-424."""
    if noise_schedule != "linear" or predict_xstart or rescale_timesteps or timestep_respacing:
        raise NotImplementedError("DiffPIR uses the linear schedule, eps-prediction, no respacing (utils_model.py:353-387)")
    return HipGaussianDiffusion(steps)


def create_model_and_diffusion(image_size, class_cond, learn_sigma, num_channels, num_res_blocks, channel_mult, num_heads,
                               num_head_channels, num_heads_upsample, attention_resolutions, dropout, diffusion_steps,
                               noise_schedule, timestep_respacing, use_kl, predict_xstart, rescale_timesteps,
                               rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm, resblock_updown, use_fp16,
                               use_new_attention_order, engine: Optional[Engine] = None):
    """script_util.py:68-127."""
    model = create_model(image_size, num_channels, num_res_blocks, channel_mult=channel_mult, learn_sigma=learn_sigma,
                         class_cond=class_cond, use_checkpoint=use_checkpoint, attention_resolutions=attention_resolutions,
                         num_heads=num_heads, num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                         use_scale_shift_norm=use_scale_shift_norm, dropout=dropout, resblock_updown=resblock_updown,
                         use_fp16=use_fp16, use_new_attention_order=use_new_attention_order, engine=engine)
    diffusion = create_gaussian_diffusion(steps=diffusion_steps, learn_sigma=learn_sigma, noise_schedule=noise_schedule,
                                          use_kl=use_kl, predict_xstart=predict_xstart, rescale_timesteps=rescale_timesteps,
                                          rescale_learned_sigmas=rescale_learned_sigmas, timestep_respacing=timestep_respacing)
    return model, diffusion


def add_dict_to_argparser(parser, default_dict):
    """script_util.py:427-438."""
    for k, v in default_dict.items():
        v_type = type(v)
        if v is None:
            v_type = str
        elif isinstance(v, bool):
            v_type = str2bool
        parser.add_argument(f"--{k}", default=v, type=v_type)


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")
