"""Do the f16 conv kernels write outside their own buffers?  Engine F fills its buffers and goes idle; engine A then runs one
conv shape; F's buffers are read again.  GPU box only."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
H = 256
ea, ef = diffpir_amd.Engine(0), diffpir_amd.Engine(0)
lib = ea.lib
lib.dpir_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_double)]
lib.dpir_debug_conv_bench.restype = C.c_int
pat = np.random.default_rng(1).standard_normal((8, 3, H, H)).astype(np.float32)
variants = [("conv2 fp32 3x3 128->128 @256", (8, 128, 128, 256, 256, 3, 0, 1, 0)),
            ("conv6 only 128->128 @256", (8, 128, 128, 256, 256, 3, 0, 1, 2)),
            ("conv5 1x1 256->128 @256", (8, 256, 128, 256, 256, 1, 0, 1, 1)),
            ("conv6 only 256->256 @64", (8, 256, 256, 64, 64, 3, 0, 1, 2))]
ms = C.c_double(0)
for name, v in variants:          # allocate A's workspaces first so that F's buffers come after them
    ea._check(lib.dpir_debug_conv_bench(ea.h, *v, 1, C.byref(ms)))
bufs = [ef.empty(pat.shape) for _ in range(60)]
for name, v in variants:
    for b in bufs: b.copy_from(pat)
    ef.sync()
    ea._check(lib.dpir_debug_conv_bench(ea.h, *v, 20, C.byref(ms)))
    ea.sync()
    ch = [int((b.numpy() != pat).sum()) for b in bufs]
    print(f"{name:36s}: idle-engine buffers changed: {sum(c > 0 for c in ch)}/60 buffers, {sum(ch)} elements", flush=True)
