"""Per-layer diagnostic of the HIP UNet against the oracle (prints every tap's relative error, never asserts).  GPU box only.
    python tools/layer_diag.py [--model tiny|ffhq|imagenet256|imagenet512] [--size 64] [--batch 1] [--precision f16x3]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from oracle import unet_oracle as uo
from tests.gpu_common import make_model, rel_err

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="ffhq"); ap.add_argument("--size", type=int, default=64)
ap.add_argument("--batch", type=int, default=1); ap.add_argument("--precision", default="f16x3")
a = ap.parse_args()
hp = {"tiny": uo.tiny_hp, "ffhq": uo.ffhq_hp, "imagenet256": uo.imagenet256_hp, "imagenet512": uo.imagenet512_hp}[a.model]()
e = diffpir_amd.Engine(0)
e.set_precision(a.precision)
model, sd = make_model(e, hp)
g = torch.Generator().manual_seed(1)
x = torch.randn((a.batch, 3, a.size, a.size), generator=g)
t = torch.randint(0, 1000, (a.batch,), generator=g)
y = torch.arange(a.batch) % 1000 if hp.class_cond else None
taps = {}
ref = uo.unet_forward(sd, hp, x, t, y, taps=taps)
out = e.unet_forward(e.to_device(x.numpy()), t.numpy(), None if y is None else y.numpy()).numpy()
for name, tv in taps.items():
    if name == "emb":
        continue
    print(f"{name:40s} rel err {rel_err(e.read_tap(name).reshape(tv.shape), tv.numpy()):.3e}  shape {tuple(tv.shape)}")
print(f"output rel err {rel_err(out, ref.numpy()):.3e}")
