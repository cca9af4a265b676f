"""A MEASURED lower bound for a W-direction Winograd F(2,3) version of conv7 (round-5 review item 4: "measure it instead of pricing it a fourth time").

F(2,3) along W turns the 3 x 3 convolution of a pair of output pixels into FOUR independent 3 x 1 (vertical-tap) problems on transformed inputs V_i and
transformed weights U_i (m_i = sum_c,ky U_i V_i; y0 = m0 + m1 + m2, y1 = m1 - m2 - m3): 4 x 3 = 12 MFMA products per pair instead of 18.  On conv7's
tiling each of the four problems is a GEMM with K = 3 Cin over W / 2 "pixels" per row.  Its MFMA count and its weight stream are EXACTLY those of a plain
9-tap conv7 launch with Cin / 3 input channels on a half-width image -- which the product kernel can run today.  So, without writing the Winograd kernel:

    T_winograd  >=  4 x T_conv7(Cin' = Cin / 3 rounded up to 16, Cout, H, W / 2) x (3 Cin) / (9 Cin')

and that bound still OMITS everything Winograd adds on top: the input transform (a second, twice as large set of operand planes written by the
producer), an activation fill three times larger than the proxy's (all Cin channels of V_i, not Cin / 3), the inverse transform and the four-way
exchange in the epilogue.  If the bound is not well below the direct kernel's time there is nothing to win.   GPU box only.
usage: python tools/winograd_bound.py [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
e = diffpir_amd.Engine(0); e.set_precision("f16x3")
dbg = _lib.load_debug()
ms = C.c_double(0)


def t(cin, cout, h, w, iters=8):
    rc = dbg.dpir_debug_conv_bench(e.h, B, cin, cout, h, w, 3, 0, 0, 2, iters, C.byref(ms))
    if rc != 0:
        raise SystemExit(f"conv bench rc {rc}: {e.lib.dpir_last_error(e.h)}")
    return ms.value * 1e3


t(128, 128, 128, 128, 200)       # clock ramp
print(f"Winograd F(2,3)-along-W bound on conv7's tiling, B = {B}, f16x3 (us per launch)")
for cin, cout, h in ((128, 128, 256), (256, 128, 256), (256, 256, 128), (128, 128, 128), (256, 256, 64)):
    cinp = -(-(-(-cin // 3)) // 16) * 16
    direct = min(t(cin, cout, h, h) for _ in range(3))
    proxy = min(t(cinp, cout, h, h // 2) for _ in range(3))
    corr = (3.0 * cin) / (9.0 * cinp)
    bound = 4 * proxy * corr
    print(f"  3x3 {cin:4d} -> {cout:4d} @ {h:3d}^2: direct conv7 {direct:8.1f} us | proxy conv7({cinp} -> {cout} @ {h} x {h // 2}) {proxy:7.1f} us x 4 x {corr:.3f} = "
          f"{bound:8.1f} us = {bound / direct:.2f} x direct  (ideal MFMA ratio 0.67)", flush=True)
e.close()
