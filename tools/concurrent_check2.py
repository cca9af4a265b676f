"""Narrowing tools/concurrent_check.py: which of (graph replay, task, precision, NFE) makes two concurrent engines disagree with
their own sequential results.  GPU box only."""
import os, sys, itertools
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
cases = {"deblur": synth.make_case("deblur", 16, H, H, seed=100, ksize=61), "inpaint": synth.make_case("inpaint", 16, H, H, seed=100)}
for prec in os.environ.get("PRECS", "f16x3,f16x1").split(","):
    e0, e1 = mk(prec), mk(prec)
    for task, graph, nfe in [("deblur", True, 2), ("deblur", True, 8), ("deblur", False, 30), ("inpaint", True, 30), ("deblur", True, 30)]:
        case = cases[task]
        cfg = (restore.LoopConfig(task="deblur", iter_num=nfe, lambda_=7.0, zeta=0.3) if task == "deblur" else
               restore.LoopConfig(task="inpaint", iter_num=nfe, noise_level_img=0.0, lambda_=1.0, zeta=1.0))
        def loop(e, sl, off, keep=None, out=None):
            return restore.restore_batch(e, cfg, case["y"][sl], k=None if case["k"] is None else case["k"][sl],
                                         mask=None if case["mask"] is None else case["mask"][sl], noise_source="device", seed=1234,
                                         image_offset=off, use_graph=graph, _cache=keep, out_f32=out)
        seq = np.concatenate([loop(e0, slice(0, 8), 0).numpy(), loop(e1, slice(8, 16), 8).numpy()])
        cons = []
        for rep in range(2):
            k0, k1 = {}, {}
            o0, o1 = e0.empty((8, 3, H, H)), e1.empty((8, 3, H, H))
            loop(e0, slice(0, 8), 0, keep=k0, out=o0); loop(e1, slice(8, 16), 8, keep=k1, out=o1)
            e0.sync(); e1.sync()
            cons.append(np.concatenate([o0.numpy(), o1.numpy()]))
        pi = lambda a, b: " ".join(f"{q:.0e}" for q in np.abs(a - b).reshape(16, -1).max(1))
        print(f"[{prec}] {task} graph={graph} nfe={nfe}: concurrent vs sequential: {pi(cons[0], seq)} | rep2 vs rep1 max {np.abs(cons[1]-cons[0]).max():.1e}"
              f" | nan {int(np.isnan(cons[0]).sum())}", flush=True)
    e0.close(); e1.close()
