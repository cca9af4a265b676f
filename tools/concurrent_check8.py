"""ALU probe under aggressors (see include/diffpir_debug.h dpir_debug_victim_alu).  GPU box only."""
import os, sys, threading, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
ea, ef = diffpir_amd.Engine(0), diffpir_amd.Engine(0)
lib = ea.lib
lib.dpir_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_double)]
lib.dpir_debug_victim_alu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
for fn in (lib.dpir_debug_victim_fft_pk, lib.dpir_debug_victim_fft_nopk):
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
aggressors = [("none", None), ("conv2 fp32 3x3", (8, 128, 128, 256, 256, 3, 0, 1, 0)), ("conv5 1x1 f16", (8, 256, 128, 256, 256, 1, 0, 1, 1)),
              ("conv6 3x3 f16", (8, 128, 128, 256, 256, 3, 0, 1, 2))]
for name, conv in aggressors:
    stop = [False]; ms = C.c_double(0)
    def spin():
        while not stop[0]:
            lib.dpir_debug_conv_bench(ea.h, *conv, 200, C.byref(ms))
    th = None
    if conv:
        th = threading.Thread(target=spin); th.start(); time.sleep(0.05)
    out = []
    for mode, mn in enumerate(["v_add_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f32"]):
        bad = C.c_ulonglong(0)
        rc = lib.dpir_debug_victim_alu(ef.h, mode, 4096, 20000, 20, C.byref(bad))
        out.append(f"{mn}: {bad.value} bad threads of {4096*64*20}")
    for nm, fn in (("regFFT built with v_pk_*_f32", lib.dpir_debug_victim_fft_pk), ("regFFT built without", lib.dpir_debug_victim_fft_nopk)):
        bad = C.c_ulonglong(0)
        rc = fn(ef.h, 4096, 200, 20, C.byref(bad))
        out.append(f"{nm}: {bad.value} bad threads of {4096*64*20}")
    stop[0] = True
    if th: th.join()
    print(f"aggressor {name:16s}: " + " | ".join(out), flush=True)
