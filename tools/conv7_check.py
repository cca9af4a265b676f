"""conv7 prototype (csrc/conv7_proto.hip, test-only library) against conv6: bit equality of the outputs and back-to-back launch times.
GPU box only.  usage: python tools/conv7_check.py [iters]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import _lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
eng = diffpir_amd.Engine(0)
dbg = _lib.load_debug()
for (B, Cin, Cout, H, W) in [(16, 128, 128, 256, 256), (16, 256, 128, 256, 256), (16, 256, 256, 128, 128), (16, 512, 512, 64, 64), (3, 48, 136, 40, 72)]:
    ms6, ms7, bad, mx = C.c_double(), C.c_double(), C.c_ulonglong(), C.c_float()
    rc = dbg.dpir_debug_conv7_check(eng.h, B, Cin, Cout, H, W, iters, C.byref(ms6), C.byref(ms7), C.byref(bad), C.byref(mx))
    if rc != 0:
        msg = eng.lib.dpir_last_error(eng.h); print(f"B={B} {Cin}->{Cout} {H}x{W}: rc={rc} {msg.decode() if msg else ''}", flush=True)
        continue
    fl = 2.0 * 9 * Cin * Cout * H * W * B
    print(f"B={B} {Cin:4d}->{Cout:4d} {H}x{W}: mismatching elements {bad.value} (max |diff| {mx.value:.3e}) | conv6 {ms6.value * 1e3:8.1f} us "
          f"{fl / ms6.value / 1e9:6.1f} TF/s | conv7 {ms7.value * 1e3:8.1f} us {fl / ms7.value / 1e9:6.1f} TF/s | x{ms6.value / ms7.value:.3f}", flush=True)
