"""Does anything change when two engines run at the same time on one GPU?  (a) 30-NFE loops: B=16 on one engine vs 2 x B=8
sequential vs 2 x B=8 concurrent, per arithmetic mode; (b) forwards: two engines forward different halves concurrently, 20 times,
each result compared with the sequential result of the same engine.  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import restore, synth, script_util, weights
H = 256
hp = weights.model_hp("ffhq"); sd = weights.synth_state_dict(hp, 0)
def mk(prec):
    e = diffpir_amd.Engine(0); e.set_precision(prec)
    m = script_util.create_model(**weights.create_model_kwargs(hp), engine=e); m.load_state_dict(sd)
    return e
NFE = int(os.environ.get("NFE", "30"))
cfg = restore.LoopConfig(task="deblur", iter_num=NFE, lambda_=7.0, zeta=0.3)
case = synth.make_case("deblur", 16, H, H, seed=100, ksize=61)
for prec in ("f16x3", "f32"):
    e0, e1 = mk(prec), mk(prec)
    # (b) forwards
    x = np.random.default_rng(0).standard_normal((16, 3, H, H)).astype(np.float32)
    xa, xb = e0.to_device(x[:8]), e1.to_device(x[8:])
    t = np.full(8, 500)
    ra = e0.unet_forward(xa, t); e0.sync(); rb = e1.unet_forward(xb, t); e1.sync()
    ra, rb = ra.numpy(), rb.numpy()
    oa, ob = e0.empty(ra.shape), e1.empty(rb.shape)
    worst = 0.0
    for it in range(20):
        e0.unet_forward(xa, t, out=oa); e1.unet_forward(xb, t, out=ob)
        e0.sync(); e1.sync()
        worst = max(worst, np.abs(oa.numpy() - ra).max(), np.abs(ob.numpy() - rb).max())
    print(f"[{prec}] concurrent forwards vs sequential: max|diff| {worst:.3e}", flush=True)
    # (a) loops
    def loop(e, sl, off, sync=True, keep=None, out=None):
        return restore.restore_batch(e, cfg, case["y"][sl], k=case["k"][sl], noise_source="device", seed=1234, image_offset=off,
                                     use_graph=True, _cache=keep, out_f32=out)
    full = loop(e0, slice(0, 16), 0).numpy()
    seq = np.concatenate([loop(e0, slice(0, 8), 0).numpy(), loop(e1, slice(8, 16), 8).numpy()])
    seq2 = np.concatenate([loop(e0, slice(0, 8), 0).numpy(), loop(e0, slice(8, 16), 8).numpy()])
    k0, k1 = {}, {}
    o0, o1 = e0.empty((8, 3, H, H)), e1.empty((8, 3, H, H))
    loop(e0, slice(0, 8), 0, keep=k0, out=o0); loop(e1, slice(8, 16), 8, keep=k1, out=o1)
    e0.sync(); e1.sync()
    con = np.concatenate([o0.numpy(), o1.numpy()])
    pi = lambda a, b: " ".join(f"{q:.0e}" for q in np.abs(a - b).reshape(16, -1).max(1))
    print(f"[{prec}] {NFE}-NFE loops: 2x8 sequential (two engines) vs B=16: {pi(seq, full)}")
    print(f"[{prec}]               2x8 sequential (one engine)  vs B=16: {pi(seq2, full)}")
    print(f"[{prec}]               2x8 concurrent vs 2x8 sequential:     {pi(con, seq)}", flush=True)
    e0.close(); e1.close()
