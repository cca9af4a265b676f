"""Is it LDS or registers?  A probe kernel parks a pattern in LDS + registers while conv kernels of another engine run.  GPU box only."""
import os, sys, threading, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
ea, ef = diffpir_amd.Engine(0), diffpir_amd.Engine(0)
lib = ea.lib
lib.dpir_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_double)]
lib.dpir_debug_conv_bench.restype = C.c_int
lib.dpir_debug_victim.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_ulonglong)]
lib.dpir_debug_victim.restype = C.c_int
aggressors = [("none", None), ("conv2 fp32 3x3", (8, 128, 128, 256, 256, 3, 0, 1, 0)), ("conv5 1x1", (8, 256, 128, 256, 256, 1, 0, 1, 1)),
              ("conv6 3x3", (8, 128, 128, 256, 256, 3, 0, 1, 2))]
victims = [(19072, 64, 3072), (36992, 256, 432), (8192, 256, 2048), (65536, 256, 512)]
for name, conv in aggressors:
    stop = [False]; ms = C.c_double(0)
    def spin():
        while not stop[0]:
            lib.dpir_debug_conv_bench(ea.h, *conv, 200, C.byref(ms))
    th = None
    if conv:
        th = threading.Thread(target=spin); th.start(); time.sleep(0.05)
    out = []
    for lds, thr, blocks in victims:
        bad = C.c_ulonglong(0)
        rc = lib.dpir_debug_victim(ef.h, lds, thr, blocks, 2000, 30, C.byref(bad))
        out.append(f"lds={lds} thr={thr}: rc {rc} LDS-bad {bad.value & 0xffffffff} reg-bad {bad.value >> 32}")
    stop[0] = True
    if th: th.join()
    print(f"aggressor {name:16s}: " + " | ".join(out), flush=True)
