"""Does running two half-batches on two streams overlap the HBM-bound passes of one with the MFMA-bound kernels of the other?  (GPU box only.)
Two engines (own stream, own workspace) each run FFHQ forwards at B = 8 from their own host thread, against one engine at B = 16; DPIR_FUSE_H1 as set by the caller."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import script_util, weights


def make(B):
    e = diffpir_amd.Engine(0)
    e.set_precision("f16x3")
    hp = weights.model_hp("ffhq")
    m = script_util.create_model(**weights.create_model_kwargs(hp), engine=e)
    m.load_state_dict(weights.synth_state_dict(hp, 0))
    x = e.to_device(np.random.default_rng(B).standard_normal((B, 3, 256, 256)).astype(np.float32))
    t = np.full(B, 500)
    out = e.unet_forward(x, t); e.sync()
    return e, x, t, out


def loop(e, x, t, out, n):
    for _ in range(n):
        e.unet_forward(x, t, out=out)
    e.sync()


N = 12
one = make(16)
loop(*one, 3)
t0 = time.perf_counter(); loop(*one, N); t1 = time.perf_counter() - t0
print(f"one engine  B=16: {t1 / N * 1e3:7.2f} ms per 16 images")
for Bh in (8, 16):
    a, b = make(Bh), make(Bh)
    for eng in (a, b):
        loop(*eng, 2)
    ths = [threading.Thread(target=loop, args=(*eng, N)) for eng in (a, b)]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    t2 = time.perf_counter() - t0
    print(f"two engines B={Bh} each, concurrent: {t2 / N * 1e3:7.2f} ms per {2 * Bh} images  ({t2 / N * 1e3 * 16 / (2 * Bh):7.2f} ms per 16 images)")
    t0 = time.perf_counter(); loop(*a, N); t3 = time.perf_counter() - t0
    print(f"   one of them alone B={Bh}: {t3 / N * 1e3:7.2f} ms per {Bh} images")
    a[0].close(); b[0].close()
one[0].close()
