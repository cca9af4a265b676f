"""The wave-per-transform prox kernels (csrc/fft4.hip, mode 'wave') against the two-pass kernels (csrc/fft2.hip, mode 'launches'): spectra and
dpir_prox_fft_apply results over repeated calls (run-to-run mismatches counted), and us per apply of both (per-apply events, graph, eager).  GPU box only.
usage: python tools/prox_modes_check.py [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import synth, utils_sisr as sr

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = diffpir_amd.Engine(0)
label = os.environ.get("RUN_LABEL", "")
for (B, H, sf) in ((16, 256, 1), (64, 256, 1), (1, 256, 1), (5, 256, 1), (16, 256, 4), (32, 256, 4), (32, 256, 2), (8, 512, 4), (3, 512, 1)):
    rng = np.random.default_rng(B * 7 + sf)
    y = eng.to_device(rng.random((B, 3, H // sf, H // sf)).astype(np.float32))
    kk = rng.random((B, 1, 25, 25)).astype(np.float32); kk /= kk.sum(axis=(2, 3), keepdims=True)
    kd = eng.to_device(kk)
    x0h = (rng.random((B, 3, H, H)).astype(np.float32) * 2 - 1)
    res = {}
    modes = [m for m in os.environ.get("PROX_MODES", "launches,wave").split(",") if not (m.startswith("wave") and H != 256)]
    for mode in modes:
        eng.set_prox_launch(mode)
        pre = sr.pre_calculate(y, kd, sf)
        h = pre[0].spectra.handle
        if mode == modes[0]:
            spec0 = [pre[i].numpy() for i in (0, 2, 3)]
        else:
            spec = [pre[i].numpy() for i in (0, 2, 3)]
            print("   spectra FB/F2B/FBFy max|diff| vs", modes[0], [float(np.abs(a - b).max()) for a, b in zip(spec0, spec)], "scale", [float(np.abs(a).max()) for a in spec0])
        outs = []
        bad = 0
        for r in range(reps):
            x0 = eng.to_device(x0h)
            eng._check(eng.lib.dpir_prox_fft_apply(eng.h, h, x0.ptr, 0.02 + 0.01 * (r % 3), 1.0 if r % 2 else 0.7))
            o = x0.numpy()
            if r < 6:
                outs.append(o)
            elif not np.array_equal(o, outs[r % 6]):
                bad += 1
        # timing: back-to-back applies in place
        x0 = eng.to_device(x0h)
        for _ in range(3):
            eng._check(eng.lib.dpir_prox_fft_apply(eng.h, h, x0.ptr, 0.05, 1.0))
        eng.sync()
        n = 50
        eng.prof_enable(True); eng.prof_reset()
        for _ in range(n):
            eng._check(eng.lib.dpir_prox_fft_apply(eng.h, h, x0.ptr, 0.05, 1.0))
        eng.sync()
        ms, cnt = eng.prof_read()["fft_prox"]
        eng.prof_enable(False)
        import ctypes as C
        us_g, us_e = C.c_float(), C.c_float()
        eng._check(eng.lib.dpir_prox_fft_apply_timed(eng.h, h, x0.ptr, 0.05, 1.0, 40, 1, C.byref(us_g)))
        eng._check(eng.lib.dpir_prox_fft_apply_timed(eng.h, h, x0.ptr, 0.05, 1.0, 40, 0, C.byref(us_e)))
        res[mode] = (outs, ms / n * 1e3, bad, us_g.value, us_e.value)
    half = H // 2 + 1
    alg = B * (3 * H * H * 4 * 2 + 3 * H * half * 8 + H * half * 4 + (H * half * 8 if sf > 1 else 0))
    line = f"{label} B={B} {H}^2 sf={sf}:"
    for m in modes:
        us = res[m][1]
        dmax = max(float(np.abs(a - b).max()) for a, b in zip(res[modes[0]][0], res[m][0]))
        line += f" | {m} evt/apply {us:6.1f} graph {res[m][3]:6.1f} us ({alg/res[m][3]/8e6*100:4.1f}%) eager {res[m][4]:6.1f} diff-vs-{modes[0]} {dmax:.1e} flaky {res[m][2]}"
    print(line, flush=True)
eng.close()
