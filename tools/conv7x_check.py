"""conv7x (csrc/conv7x.hip, test-only library: conv7 generalised to all of conv6's cases) against conv6 on the same operands: bit
equality of outputs + fused GroupNorm sums (whole K) or of the split-K partial slabs, then back-to-back launch times.
Written at the end of round 3 with no GPU time left -- FIRST thing to run in the next round.  GPU box only.
usage: python tools/conv7x_check.py [iters]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (B, Cin, Cout, H, W, res_mode, x1, split, scaled)
CASES = [
    # geometry 0 (W >= 32): what conv7 already does, through the generalised kernel
    (16, 128, 128, 256, 256, 0, 0, 0, 0), (16, 256, 256, 128, 128, 2, 0, 0, 0), (3, 48, 128, 40, 72, 1, 0, 0, 0),
    # ... plus split-K, f16x1, the dgrad scale, a partial last co-block with an idle co-half (Cout = 6) and a partially filled one (200)
    (16, 512, 512, 32, 32, 0, 0, 1, 0), (16, 128, 128, 256, 256, 0, 1, 0, 0), (16, 128, 128, 256, 256, -1, 0, 0, 1),
    (16, 128, 6, 256, 256, -1, 0, 0, 0), (4, 64, 200, 64, 64, 0, 0, 0, 0),
    # geometry 1 (16 <= W < 32)
    (16, 512, 512, 16, 16, 0, 0, 0, 0), (16, 512, 512, 16, 16, 0, 0, 1, 0), (16, 1024, 512, 16, 16, 1, 0, 1, 0), (5, 96, 128, 24, 20, 2, 1, 0, 0),
    # geometry 2 (8 <= W < 16: four images per tile)
    (16, 512, 512, 8, 8, 0, 0, 0, 0), (16, 512, 512, 8, 8, 0, 0, 1, 0), (7, 1024, 512, 8, 8, 1, 0, 1, 0), (6, 64, 128, 12, 8, -1, 1, 0, 1),
]


def run(iters=5, cases=CASES, engine=None):
    import diffpir_amd
    from diffpir_amd import _lib
    eng = engine if engine is not None else diffpir_amd.Engine(0)
    dbg = _lib.load_debug()
    bad_total = 0
    for (B, Cin, Cout, H, W, mode, x1, split, scaled) in cases:
        ms6, ms7, bad, mx, ks = C.c_double(), C.c_double(), C.c_ulonglong(), C.c_float(), C.c_int()
        rc = dbg.dpir_debug_conv7x_check(eng.h, B, Cin, Cout, H, W, mode, x1, split, scaled, iters,
                                         C.byref(ms6), C.byref(ms7), C.byref(bad), C.byref(mx), C.byref(ks))
        tag = f"B={B:2d} {Cin:4d}->{Cout:4d} {H}x{W} res {mode:2d} x1 {x1} split {split} scaled {scaled}"
        if rc != 0:
            msg = eng.lib.dpir_last_error(eng.h)
            print(f"{tag}: rc={rc} {msg.decode() if msg else ''}", flush=True)
            bad_total += 1
            continue
        bad_total += bad.value
        print(f"{tag}: ksplit {ks.value:2d}, mismatching elements {bad.value} (max |diff| {mx.value:.3e}) | conv6 {ms6.value * 1e3:8.1f} us | "
              f"conv7x {ms7.value * 1e3:8.1f} us | x{ms6.value / ms7.value:.3f}", flush=True)
    return bad_total


if __name__ == "__main__":
    n = run(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
    print("CONV7X CHECK", "OK" if n == 0 else f"FAILED ({n})", flush=True)
    sys.exit(0 if n == 0 else 1)
