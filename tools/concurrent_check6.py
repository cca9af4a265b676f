"""Size and location of the perturbation the f16 conv kernels cause in a concurrently running FFT prox of another engine:
(i) pre_calculate spectra (forward row + column FFT kernels), (ii) prox at alpha = 1 (well conditioned), (iii) alpha = 7e-7.
GPU box only."""
import os, sys, threading, time, ctypes as C
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
x0h = (case["gt"] * 2 - 1).astype(np.float32)
def spectra():
    p = sr.pre_calculate(y, k, 1, engine=ef)
    return p, [p[0].numpy(), p[2].numpy(), p[3].numpy()]
pre, ref_s = spectra()
def prox(alpha, buf):
    buf.copy_from(x0h)
    ef._check(lib.dpir_prox_fft_apply(ef.h, pre[0].spectra.handle, buf.ptr, alpha, 1.0)); ef.sync()
    return buf.numpy()
buf = ef.empty(x0h.shape)
ref1, ref7 = prox(1.0, buf), prox(7e-7, buf)
conv = (8, 256, 128, 256, 256, 1, 0, 1, 1)          # conv5 1x1 256->128 @256
for rnd in range(2):
    ms = C.c_double(0)
    stop = [False]
    def spin():
        while not stop[0]:
            lib.dpir_debug_conv_bench(ea.h, *conv, 200, C.byref(ms))
    th = threading.Thread(target=spin)
    th.start(); time.sleep(0.05)
    ds = [0.0, 0.0, 0.0]; nch = 0
    for _ in range(10):
        _, s = spectra()
        d = [float(np.abs(a - b).max()) for a, b in zip(s, ref_s)]
        ds = [max(a, b) for a, b in zip(ds, d)]; nch += any(v > 0 for v in d)
    d1 = [float(np.abs(prox(1.0, buf) - ref1).max()) for _ in range(20)]
    d7 = [float(np.abs(prox(7e-7, buf) - ref7).max()) for _ in range(20)]
    alive = th.is_alive()
    stop[0] = True
    th.join()
    print(f"round {rnd}: conv still running at the end: {alive}; spectra changed {nch}/10 (max FB {ds[0]:.2e} F2B {ds[1]:.2e} FBFy {ds[2]:.2e}; "
          f"scale FBFy {np.abs(ref_s[2]).max():.1f}) | prox alpha=1: changed {sum(v > 0 for v in d1)}/20 worst {max(d1):.2e} | "
          f"alpha=7e-7: changed {sum(v > 0 for v in d7)}/20 worst {max(d7):.2e}", flush=True)
