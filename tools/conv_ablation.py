"""Isolated timing + ablation of the conv kernel on the dominant layer shapes (GPU box only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
eng = diffpir_amd.Engine(0)
fn = eng.lib.dpir_debug_conv_bench
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_double)]
def run(B, Cin, Cout, H, ks=3, mode=0, prm=1, dbg=0, iters=5):
    ms = C.c_double()
    rc = fn(eng.h, B, Cin, Cout, H, H, ks, mode, prm, dbg, iters, C.byref(ms))
    assert rc == 0, eng.lib.dpir_last_error(eng.h)
    fl = 2.0 * ks * ks * Cin * Cout * H * H * B
    return ms.value, fl / (ms.value * 1e-3) / 1e12
B = int(os.environ.get("ABL_B", "16"))
for name, (Cin, Cout, H, ks) in {"c3 128->128@256": (128, 128, 256, 3), "c3 256->128@256": (256, 128, 256, 3),
                                 "c3 256->256@64": (256, 256, 64, 3), "c3 512->512@16": (512, 512, 16, 3), "c3 512->512@8": (512, 512, 8, 3),
                                 "c1 256->128@256": (256, 128, 256, 1), "c1 384->128@128": (384, 128, 128, 1), "c1 768->256@64": (768, 256, 64, 1), "c1 512->1536@16": (512, 1536, 16, 1)}.items():
    print(f"== {name} B={B}")
    only = os.environ.get("ABL_ONLY")
    rows1 = (("fp32 conv2 prm=null", 0, 0), ("conv5 prm=null", 64 | 128, 0), ("fp32 conv2 with prm", 0, 1), ("conv5 with prm", 64 | 128, 1))
    for label, dbg, prm in rows1 if ks == 1 else (("full", 0, 1), ("conv4 + act_split", 64 | 128, 1), ("conv4 alone", 64 | 128 | 256, 1),
                            ("conv4 no weight DMA", 64 | 128 | 256 | 1024, 1), ("conv4 no activation DMA", 64 | 128 | 256 | 2048, 1), ("conv4 no weight DMA, no epilogue", 64 | 128 | 256 | 1024 | 16, 1), ("conv4 no act DMA, no epilogue", 64 | 128 | 256 | 2048 | 16, 1), ("conv4 MFMA only", 64 | 128 | 256 | 4 | 16, 1), ("conv4 no MFMA", 64 | 128 | 256 | 1, 1),
                            ("conv4 no epilogue", 64 | 128 | 256 | 16, 1), ("conv4 no DMA", 64 | 128 | 256 | 4, 1), ("f16x3 full", 64, 1), ("f16x3 no transform", 64 | 2, 1), ("f16x3 no MFMA", 64 | 1, 1),
                            ("f16x3 MFMA only", 64 | 4 | 8 | 2 | 16, 1), ("f16x3 no loads", 64 | 4, 1), ("f16x3 no epilogue", 64 | 16, 1), ("no prologue transform", 2, 1), ("prm=null", 0, 0), ("no MFMA", 1, 1),
                            ("no global loads", 4, 1), ("no LDS stores", 8, 1), ("no epilogue", 16, 1),
                            ("MFMA only (4|8|2|16)", 4 | 8 | 2 | 16, 1), ("no MFMA, no loads", 1 | 4, 1),
                            ("no MFMA no loads no lds", 1 | 4 | 8, 1)):
        if only and only not in label:
            continue
        ms, tf = run(B, Cin, Cout, H, ks, 0, prm, dbg)
        print(f"   {label:48s} {ms*1e3:9.1f} us   {tf:7.1f} TF-equivalent")
