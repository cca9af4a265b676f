"""conv4 vs conv6 on the dominant 3x3 shapes, isolated (act planes prepared once, split weights synthetic).  GPU box only.
usage: python tools/conv_compare.py [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
eng = diffpir_amd.Engine(0)
fn = eng.lib.dpir_debug_conv_bench
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_double)]
def run(B, Cin, Cout, H, dbg, iters=10):
    ms = C.c_double()
    rc = fn(eng.h, B, Cin, Cout, H, H, 3, 0, 1, dbg, iters, C.byref(ms))
    assert rc == 0, eng.lib.dpir_last_error(eng.h)
    return ms.value, 2.0 * 9 * Cin * Cout * H * H * B / (ms.value * 1e-3) / 1e12
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [(128, 128, 256), (256, 128, 256), (256, 256, 128), (128, 128, 128), (384, 128, 128), (256, 256, 64), (512, 256, 64), (512, 512, 32), (256, 256, 32),
          (512, 512, 16), (1024, 512, 16), (512, 512, 8), (128, 6, 256), (3, 128, 256)]
if os.environ.get("CMP_IMAGENET"):
    shapes = [(256, 256, 256), (512, 256, 256), (256, 256, 128), (512, 512, 128), (512, 512, 64), (1024, 512, 64), (1024, 1024, 32), (1024, 1024, 16), (1024, 1024, 8)]
tot4 = tot6 = 0.0
for (Cin, Cout, H) in shapes:
    r = []
    for rep in range(2):                       # interleaved repeats
        m4, t4 = run(B, Cin, Cout, H, 64 | 128 | 256)
        m6, t6 = run(B, Cin, Cout, H, 64 | 128 | 256 | 4096)
        r.append((m4, t4, m6, t6))
    m4, t4, m6, t6 = min(r, key=lambda q: q[0])[0:2] + min(r, key=lambda q: q[2])[2:4]
    tot4 += m4; tot6 += m6
    print(f"{Cin:5d}->{Cout:4d} @{H:3d}^2 B={B}: conv4 {m4*1e3:8.1f} us {t4:6.1f} TF-eq | conv6 {m6*1e3:8.1f} us {t6:6.1f} TF-eq | conv6/conv4 time {m6/m4:5.2f}")
print(f"sum: conv4 {tot4:.3f} ms, conv6 {tot6:.3f} ms")
