"""Batch-composition invariance on the GPU: every image's UNet output / loop result must not depend on which batch it
travels in (beyond fp32 rounding).  Forward: f16x3 vs f32 kernels at B = 1..16 and B-batch vs one-by-one; loop: 16 images in one
batch vs 2 x 8 vs 4 x 4 (device Philox keyed by global image index), both arithmetic modes.  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import restore, synth, script_util, weights

H = 256
hp = weights.model_hp("ffhq")
sd = weights.synth_state_dict(hp, 0)
def mk(prec):
    e = diffpir_amd.Engine(0); e.set_precision(prec)
    m = script_util.create_model(**weights.create_model_kwargs(hp), engine=e); m.load_state_dict(sd)
    return e
ef, eh = mk("f32"), mk("f16x3")
x = np.random.default_rng(0).standard_normal((16, 3, H, H)).astype(np.float32)
def fwd(e, xs):
    o = e.unet_forward(e.to_device(xs), np.full(len(xs), 500)); e.sync(); return o.numpy()
one = np.concatenate([fwd(ef, x[i:i + 1]) for i in range(16)])
sc = np.abs(one).max()
print(f"output scale {sc:.3f}")
for B in (1, 2, 4, 8, 16):
    a, b = fwd(ef, x[:B]), fwd(eh, x[:B])
    print(f"forward B={B:2d}: f32 vs one-by-one f32 {np.abs(a - one[:B]).max() / sc:.2e} | f16x3 vs one-by-one f32 {np.abs(b - one[:B]).max() / sc:.2e}", flush=True)
    if B in (8, 16):
        per = np.abs(b - one[:B]).reshape(B, -1).max(1) / sc
        print("   per image:", " ".join(f"{v:.1e}" for v in per))
for nfe in (2, 6):
    cfg = restore.LoopConfig(task="deblur", iter_num=nfe, lambda_=7.0, zeta=0.3)
    case = synth.make_case("deblur", 16, H, H, seed=100, ksize=61)
    res = {}
    for name, e in (("f32", ef), ("f16x3", eh)):
        for n in (16, 8, 4):
            outs = []
            for l in range(16 // n):
                sl = slice(l * n, (l + 1) * n)
                o = restore.restore_batch(e, cfg, case["y"][sl], k=case["k"][sl], noise_source="device", seed=1234, image_offset=l * n, use_graph=(n != 4))
                outs.append(o.numpy())
            res[(name, n)] = np.concatenate(outs)
    base = res[("f32", 16)]
    for key, v in res.items():
        per = np.abs(v - base).reshape(16, -1).max(1)
        print(f"loop {nfe} NFE {key}: max|diff| vs f32 B=16 {per.max():.3e}   per image: " + " ".join(f"{q:.0e}" for q in per), flush=True)
