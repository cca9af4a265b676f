"""Per-layer diagnostic at full size (prints, never asserts): FFHQ topology, B=2, 256x256, f16x3.  GPU box only."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from oracle import unet_oracle as uo
from tests.gpu_common import make_model, rel_err
e = diffpir_amd.Engine(0)
e.set_precision("f16x3")
hp = uo.ffhq_hp()
model, sd = make_model(e, hp)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = torch.randn((B, 3, 256, 256), generator=torch.Generator().manual_seed(3))
t = torch.tensor([999, 37, 500][:B])
taps = {}
ref = uo.unet_forward(sd, hp, x, t, taps=taps)
out = e.unet_forward(e.to_device(x.numpy()), t.numpy()).numpy()
print(f"output rel err {rel_err(out, ref.numpy()):.3e}")
for name, tv in taps.items():
    if name == "emb":
        continue
    got = e.read_tap(name).reshape(tv.shape)
    d = np.abs(got - tv.numpy())
    per_img = [float(d[i].max() / (np.abs(tv.numpy()).max() + 1e-30)) for i in range(B)]
    bad = d > 1e-3 * np.abs(tv.numpy()).max()
    where = ""
    if bad.any():
        idx = np.argwhere(bad)
        where = f" bad {bad.sum()} of {bad.size}: n {sorted(set(idx[:,0]))[:4]} c-range {idx[:,1].min()}-{idx[:,1].max()} y {idx[:,2].min()}-{idx[:,2].max()} x {idx[:,3].min()}-{idx[:,3].max()}" if idx.shape[1] == 4 else f" bad {bad.sum()}"
    print(f"   {name:28s} {tuple(tv.shape)!s:22s} rel/img {['%.2e' % v for v in per_img]}{where}")
