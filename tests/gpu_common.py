"""Helpers shared by the -m gpu tests: engine fixture plumbing and oracle-side model loading."""
import numpy as np
import torch

from diffpir_amd import _lib, script_util
from oracle import unet_oracle as uo


def make_model(engine, hp: uo.UNetHP, seed=0, precision=None):
    """Build the engine model through the reference-shaped factory and load the oracle's synthetic weights."""
    if precision is not None:
        engine.set_precision(precision)
    model = script_util.create_model(
        image_size=hp.image_size, num_channels=hp.model_channels, num_res_blocks=hp.num_res_blocks,
        channel_mult=",".join(str(c) for c in hp.channel_mult) if hp.channel_mult else "",
        learn_sigma=True, class_cond=hp.class_cond, attention_resolutions=hp.attention_resolutions,
        num_heads=4, num_head_channels=hp.num_head_channels, num_heads_upsample=-1, use_scale_shift_norm=True,
        dropout=0.1, resblock_updown=True, use_fp16=False, use_new_attention_order=False, engine=engine,
        num_classes=hp.num_classes)
    sd = uo.synth_state_dict(hp, seed)
    model.load_state_dict({k: v.numpy() for k, v in sd.items()})
    return model, sd


def seeded_noise_fn_torch(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda like: torch.randn(like.shape, generator=g, dtype=torch.float32)


def seeded_noise_fn_np(seed):
    """Same stream as seeded_noise_fn_torch, returning numpy (for the engine's host-noise path)."""
    g = torch.Generator().manual_seed(seed)
    return lambda shape: torch.randn(tuple(shape), generator=g, dtype=torch.float32).numpy()


def rel_err(a, b):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
