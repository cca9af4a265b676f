"""conv7 (csrc/conv7.hip, the 3x3 kernel of the f16 modes) against conv6 (csrc/conv6.hip, kept for two launch classes of the 8 x 32
geometry) on the same operands: bit equality of outputs + fused GroupNorm sums (whole K) or of the split-K partial slabs, then
back-to-back launch times.  Round 4 ran it once more widely (all three geometries, while conv6 still had them): profiles/r04/conv7x_check.log.
GPU box only.  usage: python tools/conv7_check.py [iters]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (B, Cin, Cout, H, W, res_mode, x1, split, scaled) -- W >= 32: the geometry conv6 is still built for
CASES = [
    (16, 128, 128, 256, 256, 0, 0, 0, 0), (16, 256, 128, 256, 256, 0, 0, 0, 0), (16, 256, 256, 128, 128, 2, 0, 0, 0), (16, 256, 256, 128, 128, 1, 0, 0, 0),
    (16, 512, 512, 64, 64, -1, 0, 0, 0), (3, 48, 128, 40, 72, 1, 0, 0, 0), (2, 16, 128, 32, 32, 0, 0, 0, 0), (1, 1024, 128, 32, 64, -1, 0, 0, 0),
    # split-K, f16x1, the dgrad scale, a last co-block with an idle co-half (Cout = 6) and a partially filled one (200)
    (16, 512, 512, 32, 32, 0, 0, 1, 0), (16, 128, 128, 256, 256, 0, 1, 0, 0), (16, 128, 128, 256, 256, -1, 0, 0, 1),
    (16, 128, 6, 256, 256, -1, 0, 0, 0), (4, 64, 200, 64, 64, 0, 0, 0, 0), (5, 96, 128, 36, 44, 2, 1, 1, 1),
    # at most 32 output channels: conv7's NARROW variant (one co-tile, the four waves share the pixels)
    (16, 128, 3, 256, 256, -1, 0, 0, 1), (3, 48, 24, 40, 72, 0, 1, 0, 0), (2, 512, 32, 32, 32, 1, 0, 1, 0),
]


def run(iters=5, cases=CASES, engine=None):
    import diffpir_amd
    from diffpir_amd import _lib
    eng = engine if engine is not None else diffpir_amd.Engine(0)
    dbg = _lib.load_debug()
    bad_total = 0
    for (B, Cin, Cout, H, W, mode, x1, split, scaled) in cases:
        ms6, ms7, bad, mx, ks = C.c_double(), C.c_double(), C.c_ulonglong(), C.c_float(), C.c_int()
        rc = dbg.dpir_debug_conv7_check(eng.h, B, Cin, Cout, H, W, mode, x1, split, scaled, iters,
                                        C.byref(ms6), C.byref(ms7), C.byref(bad), C.byref(mx), C.byref(ks))
        tag = f"B={B:2d} {Cin:4d}->{Cout:4d} {H}x{W} res {mode:2d} x1 {x1} split {split} scaled {scaled}"
        if rc != 0:
            msg = eng.lib.dpir_last_error(eng.h)
            print(f"{tag}: rc={rc} {msg.decode() if msg else ''}", flush=True)
            bad_total += 1
            continue
        bad_total += bad.value
        fl = 2.0 * 9 * Cin * Cout * H * W * B
        print(f"{tag}: ksplit {ks.value:2d}, mismatching elements {bad.value} (max |diff| {mx.value:.3e}) | conv6 {ms6.value * 1e3:8.1f} us | "
              f"conv7 {ms7.value * 1e3:8.1f} us {fl / ms7.value / 1e9:6.1f} TF/s | x{ms6.value / ms7.value:.3f}", flush=True)
    return bad_total


if __name__ == "__main__":
    n = run(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
    print("CONV7 CHECK", "OK" if n == 0 else f"FAILED ({n})", flush=True)
    sys.exit(0 if n == 0 else 1)
