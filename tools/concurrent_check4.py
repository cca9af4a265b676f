"""Which f16-path kernel perturbs the FFT prox of another engine running at the same time?  Thread A: one conv shape in a loop
(dpir_debug_conv_bench), thread B: FFT prox applications compared with their sequential result.  GPU box only."""
import os, sys, threading, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import synth, utils_sisr as sr
H = 256
ea, ef = diffpir_amd.Engine(0), diffpir_amd.Engine(0)
lib = ea.lib
lib.dpir_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_double)]
lib.dpir_debug_conv_bench.restype = C.c_int
case = synth.make_case("deblur", 8, H, H, seed=100, ksize=61)
y, k = ef.to_device(case["y"]), ef.to_device(case["k"])
pre = sr.pre_calculate(y, k, 1, engine=ef)
x0h = (case["gt"] * 2 - 1).astype(np.float32)
b0 = ef.empty(x0h.shape); b0.copy_from(x0h)
ef._check(lib.dpir_prox_fft_apply(ef.h, pre[0].spectra.handle, b0.ptr, 7e-7, 1.0)); ef.sync()
ref = b0.numpy()
bufs = [ef.empty(x0h.shape) for _ in range(60)]
variants = [("conv2 fp32 3x3 128->128 @256", (8, 128, 128, 256, 256, 3, 0, 1, 0)),
            ("conv6 only 128->128 @256", (8, 128, 128, 256, 256, 3, 0, 1, 2)),
            ("act_split+conv6 128->128 @256", (8, 128, 128, 256, 256, 3, 0, 1, 1)),
            ("conv5 1x1 256->128 @256", (8, 256, 128, 256, 256, 1, 0, 1, 1)),
            ("conv6 only 256->256 @64", (8, 256, 256, 64, 64, 3, 0, 1, 2)),
            ("conv6 only 512->512 @16 (split-K)", (8, 512, 512, 16, 16, 3, 0, 1, 2)),
            ("conv6 only 128->6 @256", (8, 128, 6, 256, 256, 3, 0, 1, 2))]
for name, (B, ci, co, h, w, ks, mode, prm, dbg) in variants:
    for b in bufs: b.copy_from(x0h)
    ef.sync()
    ms = C.c_double(0)
    iters = 60 if h == 256 else 600
    th = threading.Thread(target=lambda: lib.dpir_debug_conv_bench(ea.h, B, ci, co, h, w, ks, mode, prm, dbg, iters, C.byref(ms)))
    th.start()
    import time; time.sleep(0.01)
    for b in bufs:
        ef._check(lib.dpir_prox_fft_apply(ef.h, pre[0].spectra.handle, b.ptr, 7e-7, 1.0))
        ef.sync()
    th.join()
    dp = [float(np.abs(b.numpy() - ref).max()) for b in bufs]
    print(f"{name:40s}: conv {ms.value:.3f} ms/launch; FFT prox changed {sum(d > 0 for d in dp)}/60, worst {max(dp):.2e}", flush=True)
