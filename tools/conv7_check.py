"""conv7 (csrc/conv7.hip) against conv6 on the same operands: bit equality of the outputs AND of the fused GroupNorm sums for every
residual form, then back-to-back launch times.  GPU box only.  usage: python tools/conv7_check.py [iters]
(tests/test_gpu_ops.py runs `run()` on three small cases in the -m gpu suite.)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (B, Cin, Cout, H, W, residual forms): -1 none, 0 same shape, 1 half resolution (nearest up), 2 double resolution (2x2 pooled)
CASES = [(16, 128, 128, 256, 256, (-1, 0, 1, 2)), (16, 256, 128, 256, 256, (0,)), (16, 256, 256, 128, 128, (0, 1, 2)),
         (16, 512, 512, 64, 64, (0,)), (3, 48, 128, 40, 72, (-1, 0, 1, 2)), (1, 6, 128, 256, 256, (-1,)), (2, 1024, 256, 32, 32, (0,))]


def run(iters=10, cases=CASES, engine=None):
    """Number of differing elements over all cases (a failed call counts as one)."""
    import diffpir_amd
    from diffpir_amd import _lib
    eng = engine if engine is not None else diffpir_amd.Engine(0)
    dbg = _lib.load_debug()
    bad_total = 0
    for (B, Cin, Cout, H, W, modes) in cases:
        for mode in modes:
            ms6, ms7, bad, mx = C.c_double(), C.c_double(), C.c_ulonglong(), C.c_float()
            rc = dbg.dpir_debug_conv7_check(eng.h, B, Cin, Cout, H, W, mode, iters, C.byref(ms6), C.byref(ms7), C.byref(bad), C.byref(mx))
            if rc != 0:
                msg = eng.lib.dpir_last_error(eng.h)
                print(f"B={B} {Cin}->{Cout} {H}x{W} res_mode {mode}: rc={rc} {msg.decode() if msg else ''}", flush=True)
                bad_total += 1
                continue
            bad_total += bad.value
            fl = 2.0 * 9 * Cin * Cout * H * W * B
            print(f"B={B:2d} {Cin:4d}->{Cout:4d} {H}x{W} res_mode {mode:2d}: mismatching out/stat elements {bad.value} (max |diff| {mx.value:.3e}) | conv6 {ms6.value * 1e3:8.1f} us "
                  f"{fl / ms6.value / 1e9:6.1f} TF/s | conv7 {ms7.value * 1e3:8.1f} us {fl / ms7.value / 1e9:6.1f} TF/s | x{ms6.value / ms7.value:.3f}", flush=True)
    return bad_total


if __name__ == "__main__":
    n = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
    print("CONV7 CHECK", "OK" if n == 0 else f"FAILED ({n})", flush=True)
    sys.exit(0 if n == 0 else 1)
