"""conv6 vs conv4 engines on the same forward: taps of one ResBlock compared element-wise (prints).  GPU box only."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from oracle import unet_oracle as uo
from tests.gpu_common import make_model
hp = uo.ffhq_hp()
B = 2
x = torch.randn((B, 3, 256, 256), generator=torch.Generator().manual_seed(3)).numpy()
t = np.array([999, 37])
res = {}
for impl in ("6", "4"):
    os.environ["DIFFPIR_CONV"] = impl
    e = diffpir_amd.Engine(0)
    e.set_precision("f16x3")
    make_model(e, hp)
    e.unet_forward(e.to_device(x), t).numpy()
    res[impl] = {k: e.read_tap(k) for k in ("output_blocks.9.0", "output_blocks.9.1#h1", "output_blocks.9.1", "output_blocks.7.1#h1", "output_blocks.7.1")}
    e.close()
for k in res["6"]:
    a, b = res["6"][k], res["4"][k]
    n = a.size // (B * 128) if "9" in k else a.size // (B * 256)
    side = int(round(n ** 0.5))
    C = a.size // (B * side * side)
    a = a.reshape(B, C, side, side); b = b.reshape(B, C, side, side)
    d = np.abs(a - b)
    bad = d > 1e-4 * np.abs(b).max()
    print(f"{k}: shape {a.shape} max diff {d.max():.3e} (max |ref| {np.abs(b).max():.3e}) bad {bad.sum()}")
    if bad.any():
        idx = np.argwhere(bad)
        print("   images", np.bincount(idx[:, 0]), " y%8 hist", np.bincount(idx[:, 2] % 8, minlength=8), " x%32 hist", np.bincount(idx[:, 3] % 32, minlength=32))
        print("   c%32 hist", np.bincount(idx[:, 1] % 32, minlength=32))
        print("   first 12:", [tuple(int(v) for v in r) for r in idx[:12]])
        r = idx[0]
        print("   values conv6", a[tuple(r)], "conv4", b[tuple(r)])
