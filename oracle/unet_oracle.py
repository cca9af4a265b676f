"""Oracle: functional restatement of the guided-diffusion UNet forward (torch-CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows (reference paths relative to /root/reference):
  * topology           guided_diffusion/unet.py:480-616, script_util.py:130-184
  * forward            guided_diffusion/unet.py:634-663
  * ResBlock           guided_diffusion/unet.py:236-256
  * AttentionBlock     guided_diffusion/unet.py:299-305, 337-354 (legacy qkv order)
  * GroupNorm32        guided_diffusion/nn.py:17-19, 93-100 (32 groups, eps 1e-5)
  * timestep_embedding guided_diffusion/nn.py:103-121
  * hyper-parameters   utils/utils_model.py:353-387, main_ddpir.py:219-230

The network is described by a flat "plan" (list of blocks, each a list of layer
records) and a state-dict using the reference's key schema, so a checkpoint
written for the reference loads unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class UNetHP:
    """Hyper-parameters as resolved by utils_model.create_argparser + script_util.create_model."""
    image_size: int = 256
    model_channels: int = 128
    num_res_blocks: int = 1
    attention_resolutions: str = "16"
    channel_mult: Tuple[float, ...] = ()
    num_head_channels: int = 64
    learn_sigma: bool = True
    class_cond: bool = False
    num_classes: int = 1000
    in_channels: int = 3

    def resolved_channel_mult(self):
        # script_util.py:147-160
        if self.channel_mult:
            return tuple(self.channel_mult)
        return {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4),
                128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[self.image_size]

    def attention_ds(self):
        # script_util.py:162-164
        return tuple(self.image_size // int(r) for r in self.attention_resolutions.split(","))

    @property
    def out_channels(self):
        return 6 if self.learn_sigma else 3


def ffhq_hp():      # main_ddpir.py:219-224
    return UNetHP(256, 128, 1, "16")


def imagenet256_hp():  # main_ddpir.py:225-230
    return UNetHP(256, 256, 2, "8,16,32")


def imagenet512_hp():  # BASELINE config 5: 512 class-cond (script_util.py:149-150)
    return UNetHP(512, 256, 2, "8,16,32", class_cond=True)


def tiny_hp(image_size=64, mc=64, nrb=1, attn="16,32", cm=(1, 2, 2), class_cond=False):
    """Small topology with every structural feature (down/up resblocks, channel change,
    skip 1x1, attention on both paths) for fast tests."""
    return UNetHP(image_size, mc, nrb, attn, tuple(cm), 64, True, class_cond, 10)


# ------------------------------------------------------------------ topology walk
def build_plan(hp: UNetHP):
    """Returns (input_blocks, middle, output_blocks): each block is a list of layer tuples
    ('conv', cin, cout) | ('res', cin, cout, mode) | ('attn', ch), mode in {None,'down','up'}.
    unet.py:480-616."""
    cm = hp.resolved_channel_mult()
    mc = hp.model_channels
    att = hp.attention_ds()
    ch = int(cm[0] * mc)
    inp = [[("conv", hp.in_channels, ch)]]
    chans = [ch]
    ds = 1
    for level, mult in enumerate(cm):
        for _ in range(hp.num_res_blocks):
            cout = int(mult * mc)
            layers = [("res", ch, cout, None)]
            ch = cout
            if ds in att:
                layers.append(("attn", ch))
            inp.append(layers)
            chans.append(ch)
        if level != len(cm) - 1:
            inp.append([("res", ch, ch, "down")])
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch, None), ("attn", ch), ("res", ch, ch, None)]
    out = []
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(hp.num_res_blocks + 1):
            ich = chans.pop()
            cout = int(mc * mult)
            layers = [("res", ch + ich, cout, None)]
            ch = cout
            if ds in att:
                layers.append(("attn", ch))
            if level and i == hp.num_res_blocks:
                layers.append(("res", ch, ch, "up"))
                ds //= 2
            out.append(layers)
    return inp, mid, out


def state_dict_spec(hp: UNetHP) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (key, shape, kind) list in the reference's key schema.
    kind in {'w','b','gn_w','gn_b','zero_w','zero_b','emb'} drives the synthetic init."""
    mc = hp.model_channels
    ted = 4 * mc
    spec: List[Tuple[str, Tuple[int, ...], str]] = []

    def lin(p, cin, cout, zero=False):
        spec.append((p + ".weight", (cout, cin), "zero_w" if zero else "w"))
        spec.append((p + ".bias", (cout,), "zero_b" if zero else "b"))

    def conv(p, cin, cout, k, zero=False, one_d=False):
        shape = (cout, cin, k) if one_d else (cout, cin, k, k)
        spec.append((p + ".weight", shape, "zero_w" if zero else "w"))
        spec.append((p + ".bias", (cout,), "zero_b" if zero else "b"))

    def gn(p, c):
        spec.append((p + ".weight", (c,), "gn_w"))
        spec.append((p + ".bias", (c,), "gn_b"))

    def layer(p, rec):
        if rec[0] == "conv":
            conv(p, rec[1], rec[2], 3)
        elif rec[0] == "res":
            _, cin, cout, _mode = rec
            gn(p + ".in_layers.0", cin)
            conv(p + ".in_layers.2", cin, cout, 3)
            lin(p + ".emb_layers.1", ted, 2 * cout)
            gn(p + ".out_layers.0", cout)
            conv(p + ".out_layers.3", cout, cout, 3, zero=True)
            if cin != cout:
                conv(p + ".skip_connection", cin, cout, 1)
        else:
            c = rec[1]
            gn(p + ".norm", c)
            conv(p + ".qkv", c, 3 * c, 1, one_d=True)
            conv(p + ".proj_out", c, c, 1, zero=True, one_d=True)

    lin("time_embed.0", mc, ted)
    lin("time_embed.2", ted, ted)
    if hp.class_cond:
        spec.append(("label_emb.weight", (hp.num_classes, ted), "emb"))
    inp, mid, out = build_plan(hp)
    for i, blk in enumerate(inp):
        for j, rec in enumerate(blk):
            layer(f"input_blocks.{i}.{j}", rec)
    for j, rec in enumerate(mid):
        layer(f"middle_block.{j}", rec)
    for i, blk in enumerate(out):
        for j, rec in enumerate(blk):
            layer(f"output_blocks.{i}.{j}", rec)
    ch0 = int(hp.resolved_channel_mult()[0] * mc)
    gn("out.0", ch0)
    conv("out.2", ch0, hp.out_channels, 3, zero=True)
    return spec


def synth_state_dict(hp: UNetHP, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights (no checkpoint is available offline).
    numpy PCG64 streams keyed by (seed, tensor index): platform independent.  The three
    zero_module sites (unet.py:210,294,615) get small non-zero values, otherwise the
    network output is identically 0 and parity would be vacuous."""
    sd = {}
    for idx, (key, shape, kind) in enumerate(state_dict_spec(hp)):
        rng = np.random.default_rng([seed, idx])
        n = int(np.prod(shape))
        u = rng.random(n, dtype=np.float32) * 2.0 - 1.0
        if kind in ("w", "zero_w"):
            fan_in = int(np.prod(shape[1:]))
            bound = math.sqrt(3.0 / fan_in)          # unit-gain uniform
            if kind == "zero_w":
                bound *= 0.5
            v = u * np.float32(bound)
        elif kind in ("b", "zero_b"):
            v = u * np.float32(0.05)
        elif kind == "gn_w":
            v = np.float32(1.0) + u * np.float32(0.2)
        elif kind == "gn_b":
            v = u * np.float32(0.1)
        else:  # emb
            v = u * np.float32(0.5)
        sd[key] = torch.from_numpy(v.reshape(shape).astype(np.float32))
    return sd


# ------------------------------------------------------------------ forward
def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0):
    """nn.py:103-121: [cos(t f) | sin(t f)], f_i = exp(-ln(max_period) i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    e = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        e = torch.cat([e, torch.zeros_like(e[:, :1])], dim=-1)
    return e


def _gn(sd, p, x):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _resblock(sd, p, rec, x, emb):
    """unet.py:236-256 with use_scale_shift_norm=True, resblock_updown=True, dropout in eval."""
    _, cin, cout, mode = rec
    h = F.silu(_gn(sd, p + ".in_layers.0", x))
    if mode == "down":                       # unet.py:136 avg_pool2d k=2 s=2 on h and x
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    elif mode == "up":                       # unet.py:107 nearest x2 on h and x
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    scale, shift = torch.chunk(e[:, :, None, None], 2, dim=1)
    h = _gn(sd, p + ".out_layers.0", h) * (1 + scale) + shift
    h = F.conv2d(F.silu(h), sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if cin != cout:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def _attention(sd, p, x, head_ch):
    """unet.py:299-305 + QKVAttentionLegacy 337-354: heads split BEFORE q|k|v."""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(sd, p + ".norm", xf), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    nh = c // head_ch
    T = xf.shape[-1]
    q, k, v = qkv.reshape(b * nh, 3 * head_ch, T).split(head_ch, dim=1)
    s = 1.0 / math.sqrt(math.sqrt(head_ch))
    w = torch.einsum("bct,bcs->bts", q * s, k * s)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, c, T)
    h = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + h).reshape(b, c, hh, ww)


def _run_layers(sd, prefix, blk, h, emb, hp, taps):
    for j, rec in enumerate(blk):
        p = f"{prefix}.{j}"
        if rec[0] == "conv":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif rec[0] == "res":
            h = _resblock(sd, p, rec, h, emb)
        else:
            h = _attention(sd, p, h, hp.num_head_channels)
        if taps is not None:
            taps[p] = h
    return h


def unet_forward(sd, hp: UNetHP, x: torch.Tensor, t: torch.Tensor, y: Optional[torch.Tensor] = None,
                 taps: Optional[dict] = None) -> torch.Tensor:
    """unet.py:634-663.  x [B,3,H,W] fp32, t [B] int64 -> [B,out_channels,H,W].
    `taps`, if given, collects every layer's output keyed by its state-dict prefix.
    Runs without autograd unless x requires grad (the DPS modes differentiate through the network, main_ddpir.py:370-373)."""
    with torch.set_grad_enabled(bool(x.requires_grad)):
        return _unet_forward(sd, hp, x, t, y, taps)


def _unet_forward(sd, hp, x, t, y, taps):
    assert (y is not None) == hp.class_cond
    inp, mid, out = build_plan(hp)
    emb = timestep_embedding(t, hp.model_channels)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if hp.class_cond:
        emb = emb + sd["label_emb.weight"][y]
    if taps is not None:
        taps["emb"] = emb
    hs = []
    h = x.float()
    for i, blk in enumerate(inp):
        h = _run_layers(sd, f"input_blocks.{i}", blk, h, emb, hp, taps)
        hs.append(h)
    h = _run_layers(sd, "middle_block", mid, h, emb, hp, taps)
    for i, blk in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(sd, f"output_blocks.{i}", blk, h, emb, hp, taps)
    h = F.silu(_gn(sd, "out.0", h))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


def unet_flops(hp: UNetHP, H: int, W: int) -> float:
    """2*MAC over conv / conv1d / linear / attention matmuls for one image."""
    inp, mid, out = build_plan(hp)
    fl = 0.0
    mc = hp.model_channels
    ted = 4 * mc
    fl += 2 * (mc * ted + ted * ted)
    res = [H, W]

    def layer(rec):
        nonlocal fl
        h, w = res
        if rec[0] == "conv":
            fl += 2 * 9 * rec[1] * rec[2] * h * w
        elif rec[0] == "res":
            _, cin, cout, mode = rec
            if mode == "down":
                res[0] //= 2; res[1] //= 2
            elif mode == "up":
                res[0] *= 2; res[1] *= 2
            h, w = res
            fl += 2 * 9 * cin * cout * h * w + 2 * 9 * cout * cout * h * w + 2 * ted * 2 * cout
            if cin != cout:
                fl += 2 * cin * cout * h * w
        else:
            c = rec[1]
            T = h * w
            fl += 2 * c * 3 * c * T + 2 * c * c * T + 2 * 2 * T * T * c
    for blk in inp:
        for rec in blk:
            layer(rec)
    for rec in mid:
        layer(rec)
    for blk in out:
        for rec in blk:
            layer(rec)
    ch0 = int(hp.resolved_channel_mult()[0] * mc)
    fl += 2 * 9 * ch0 * hp.out_channels * H * W
    return fl
