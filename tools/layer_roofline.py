"""Per-shape roofline of the FFHQ forward's convolutions at the benched batch: every distinct 3x3 / 1x1 problem
is launched back to back on synthetic operands through the product's own dispatch (launch_conv6 -> conv7 / conv6, launch_conv5;
dpir_debug_conv_bench of libdiffpir_dbg.so, f16x3 operand-split path), and its rate is held against the 833 TF/s-eq peak.  Back-to-back
launches of ONE shape are the kernel's best case (warm L2, no neighbours); the sum over the forward's launch counts is printed next to
the 3x3-class time the bench measures inside the network.  GPU box only.  DPIR_SPLIT_TARGET / DPIR_SPLIT_BELOW (csrc/conv6.hip) change the
split-K rule of the low-resolution shapes for an A/B.   usage: python tools/layer_roofline.py [B] [first N shapes]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import _lib

# (ks, Cin, Cout, H, count per forward)  -- SURVEY.md A.1
SHAPES = [(3, 128, 128, 256, 6), (3, 256, 128, 256, 2), (3, 256, 256, 128, 2), (3, 128, 128, 128, 6), (3, 256, 256, 64, 5), (3, 384, 128, 128, 1),
          (3, 512, 512, 32, 2), (3, 512, 256, 64, 1), (3, 256, 128, 128, 1), (3, 256, 256, 32, 6), (3, 384, 256, 64, 1), (3, 512, 512, 16, 5),
          (3, 768, 256, 32, 1), (3, 512, 512, 8, 10), (3, 128, 128, 64, 2), (3, 128, 256, 64, 1), (3, 1024, 512, 16, 1), (3, 512, 256, 32, 1),
          (3, 768, 512, 16, 1), (3, 1024, 512, 8, 2), (3, 128, 6, 256, 1), (3, 256, 256, 16, 2), (3, 256, 512, 16, 1),
          (1, 256, 128, 256, 2), (1, 384, 128, 128, 1), (1, 512, 256, 64, 1), (1, 256, 128, 128, 1), (1, 384, 256, 64, 1), (1, 768, 256, 32, 1),
          (1, 128, 256, 64, 1), (1, 1024, 512, 16, 1), (1, 512, 256, 32, 1), (1, 768, 512, 16, 1), (1, 1024, 512, 8, 2), (1, 256, 512, 16, 1)]
PREC = os.environ.get("DIFFPIR_PRECISION", "f16x3")
PEAK = 2500.0 if PREC == "f16x1" else 833.3      # f16x1: one MFMA per product against the dense f16 peak; f16x3: three per product


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    shapes = SHAPES[:int(sys.argv[2])] if len(sys.argv) > 2 else SHAPES
    e = diffpir_amd.Engine(0); e.set_precision(PREC)
    dbg = _lib.load_debug()
    ms = C.c_double(0)
    tot = {1: 0.0, 3: 0.0}; fl = {1: 0.0, 3: 0.0}
    # warm-up: the first launches of a process run ~15 % slower (clock ramp; measured: the same shape 411 us first, 347 us ten configs later,
    # profiles/r04/conv7_128_128_same_work_different_image_shapes.log) -- keep the chip busy for ~50 ms before the first figure
    dbg.dpir_debug_conv_bench(e.h, B, 128, 128, 128, 128, 3, 0, 0, 2, 200 if B <= 16 else 80, C.byref(ms))
    print(f"FFHQ conv shapes at B = {B}, {PREC}, back-to-back launches after a warm-up (us per launch, TF/s-eq, fraction of {PEAK})")
    for ks, cin, cout, h, cnt in shapes:
        flops = 2.0 * cin * cout * ks * ks * h * h * B
        iters = 5 if flops > 2e11 else 20
        rc = dbg.dpir_debug_conv_bench(e.h, B, cin, cout, h, h, ks, 0, 0, 2, iters, C.byref(ms))
        if rc != 0:
            print(f"  {ks}x{ks} {cin:5d} -> {cout:4d} @ {h:3d}^2: rc {rc} ({e.lib.dpir_last_error(e.h)})"); continue
        tf = flops / (ms.value * 1e-3) / 1e12
        tot[ks] += cnt * ms.value; fl[ks] += cnt * flops
        print(f"  {ks}x{ks} {cin:5d} -> {cout:4d} @ {h:3d}^2  x{cnt:2d}: {ms.value * 1e3:8.1f} us  {tf:6.1f} TF/s-eq  {tf / PEAK:5.3f}   ({cnt * ms.value:6.3f} ms per forward)", flush=True)
    for ks in (3, 1):
        if tot[ks] == 0: continue
        print(f"sum {ks}x{ks}: {tot[ks]:.2f} ms per forward, {fl[ks] / (tot[ks] * 1e-3) / 1e12:.1f} TF/s-eq = {fl[ks] / (tot[ks] * 1e-3) / 1e12 / PEAK:.3f} of {PEAK}")
    e.close()


if __name__ == "__main__":
    main()
