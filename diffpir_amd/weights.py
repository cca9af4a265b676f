"""Model hyper-parameters, state-dict schema, checkpoint loading and synthetic weights (host side).

The key schema is the reference's: time_embed.{0,2}, input_blocks.N.0.{in_layers.{0,2},
emb_layers.1,out_layers.{0,3},skip_connection}, input_blocks.N.1.{norm,qkv,proj_out}, middle_block.{0,1,2},
output_blocks.N.{0,1,2}, out.{0,2}; conv weights OIHW, qkv [3C,C,1].  `load_checkpoint` is the only place
PyTorch touches the model (torch.load, main_ddpir.py:234).  No checkpoint exists offline, so benchmarks
and tests use `synth_state_dict`: numpy PCG64 streams keyed by (seed, tensor index), unit-gain uniform
convs, with the reference's three zero_module sites (unet.py:210,294,615) given small non-zero values.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np


@dataclass
class ModelHP:
    image_size: int = 256
    model_channels: int = 128
    num_res_blocks: int = 1
    attention_resolutions: str = "16"
    channel_mult: Tuple[float, ...] = ()
    num_head_channels: int = 64
    class_cond: bool = False
    num_classes: int = 1000

    def resolved_channel_mult(self):
        if self.channel_mult:
            return tuple(self.channel_mult)
        return {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[self.image_size]


def model_hp(name) -> ModelHP:
    """main_ddpir.py:219-230 (+ BASELINE config 5's 512 class-conditional model)."""
    if isinstance(name, ModelHP):
        return name
    if name in ("ffhq", "diffusion_ffhq_10m"):
        return ModelHP(256, 128, 1, "16")
    if name in ("imagenet256", "256x256_diffusion_uncond"):
        return ModelHP(256, 256, 2, "8,16,32")
    if name in ("imagenet512", "512x512_diffusion"):
        return ModelHP(512, 256, 2, "8,16,32", class_cond=True)
    if name == "tiny":
        return ModelHP(64, 64, 1, "16,32", (1, 2, 2))
    raise ValueError(f"unknown model {name}")


def create_model_kwargs(hp: ModelHP) -> dict:
    return dict(image_size=hp.image_size, num_channels=hp.model_channels, num_res_blocks=hp.num_res_blocks,
                channel_mult=",".join(str(c) for c in hp.channel_mult) if hp.channel_mult else "",
                learn_sigma=True, class_cond=hp.class_cond, attention_resolutions=hp.attention_resolutions,
                num_heads=4, num_head_channels=hp.num_head_channels, num_heads_upsample=-1,
                use_scale_shift_norm=True, dropout=0.1, resblock_updown=True, use_fp16=False,
                use_new_attention_order=False, num_classes=hp.num_classes)


def state_dict_spec(hp: ModelHP) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) in module-registration order of the reference UNetModel (unet.py:470-616)."""
    mc, ted = hp.model_channels, 4 * hp.model_channels
    cm = hp.resolved_channel_mult()
    att = tuple(hp.image_size // int(r) for r in hp.attention_resolutions.split(","))
    spec = []

    def wb(p, shape, zero=False):
        spec.append((p + ".weight", tuple(shape), "zero_w" if zero else "w"))
        spec.append((p + ".bias", (shape[0],), "zero_b" if zero else "b"))

    def gn(p, c):
        spec.append((p + ".weight", (c,), "gn_w"))
        spec.append((p + ".bias", (c,), "gn_b"))

    def res(p, cin, cout):
        gn(p + ".in_layers.0", cin)
        wb(p + ".in_layers.2", (cout, cin, 3, 3))
        wb(p + ".emb_layers.1", (2 * cout, ted))
        gn(p + ".out_layers.0", cout)
        wb(p + ".out_layers.3", (cout, cout, 3, 3), zero=True)
        if cin != cout:
            wb(p + ".skip_connection", (cout, cin, 1, 1))

    def attn(p, c):
        gn(p + ".norm", c)
        wb(p + ".qkv", (3 * c, c, 1))
        wb(p + ".proj_out", (c, c, 1), zero=True)

    wb("time_embed.0", (ted, mc))
    wb("time_embed.2", (ted, ted))
    if hp.class_cond:
        spec.append(("label_emb.weight", (hp.num_classes, ted), "emb"))
    ch = int(cm[0] * mc)
    wb("input_blocks.0.0", (ch, 3, 3, 3))
    chans, ds, nb = [ch], 1, 1
    for level, mult in enumerate(cm):
        for _ in range(hp.num_res_blocks):
            cout = int(mult * mc)
            res(f"input_blocks.{nb}.0", ch, cout)
            ch = cout
            if ds in att:
                attn(f"input_blocks.{nb}.1", ch)
            chans.append(ch)
            nb += 1
        if level != len(cm) - 1:
            res(f"input_blocks.{nb}.0", ch, ch)
            chans.append(ch)
            nb += 1
            ds *= 2
    res("middle_block.0", ch, ch)
    attn("middle_block.1", ch)
    res("middle_block.2", ch, ch)
    nb = 0
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(hp.num_res_blocks + 1):
            ich = chans.pop()
            cout = int(mc * mult)
            j = 0
            res(f"output_blocks.{nb}.{j}", ch + ich, cout)
            j += 1
            ch = cout
            if ds in att:
                attn(f"output_blocks.{nb}.{j}", ch)
                j += 1
            if level and i == hp.num_res_blocks:
                res(f"output_blocks.{nb}.{j}", ch, ch)
                ds //= 2
            nb += 1
    gn("out.0", ch)
    wb("out.2", (6, int(cm[0] * mc), 3, 3), zero=True)
    return spec


def synth_state_dict(hp, seed: int = 0) -> Dict[str, np.ndarray]:
    hp = model_hp(hp)
    sd = {}
    for idx, (key, shape, kind) in enumerate(state_dict_spec(hp)):
        rng = np.random.default_rng([seed, idx])
        n = int(np.prod(shape))
        u = rng.random(n, dtype=np.float32) * 2.0 - 1.0
        if kind in ("w", "zero_w"):
            bound = math.sqrt(3.0 / int(np.prod(shape[1:])))
            if kind == "zero_w":
                bound *= 0.5
            v = u * np.float32(bound)
        elif kind in ("b", "zero_b"):
            v = u * np.float32(0.05)
        elif kind == "gn_w":
            v = np.float32(1.0) + u * np.float32(0.2)
        elif kind == "gn_b":
            v = u * np.float32(0.1)
        else:
            v = u * np.float32(0.5)
        sd[key] = v.reshape(shape).astype(np.float32)
    return sd


def load_checkpoint(path: str) -> Dict[str, np.ndarray]:
    """torch.load(path, map_location='cpu') -> numpy state dict (the one use of PyTorch on the model path)."""
    import torch
    sd = torch.load(path, map_location="cpu")
    return {k: v.detach().float().numpy() for k, v in sd.items()}
