"""Two engines (two HIP queues) on ONE GPU: does anything one engine runs disturb the other?  One parametrised probe (round 2
grew eight single-purpose scripts while the packed-fp32 erratum of DESIGN.md section 4 was tracked down).  GPU box only.

    python tools/concurrency_probe.py loops    [--precisions f16x3,f16x1,f32] [--cases deblur:1:8,inpaint:1:30,...]   (task:graph:nfe)
    python tools/concurrency_probe.py forwards [--precisions ...]          concurrent UNet forwards vs each engine's sequential result
    python tools/concurrency_probe.py fft      [--precisions ...]          FFT prox of engine B under UNet forwards of engine A (+ the reverse)
    python tools/concurrency_probe.py victim   [--aggressor conv6|conv5|conv2]   parked LDS / register patterns, ALU chains and the
                                               register-FFT probe (built with / without v_pk_*_f32) beside an aggressor convolution
Every comparison is bitwise against the same engine's sequential result; the prints are the result, nothing asserts
(tests/test_gpu_concurrency.py holds the assertions that guard the fix)."""
import argparse, ctypes as C, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import restore, synth, script_util, weights, utils_sisr as sr, _lib

H = 256
hp = weights.model_hp("ffhq")
sd = weights.synth_state_dict(hp, 0)


def mk(prec):
    e = diffpir_amd.Engine(0); e.set_precision(prec)
    m = script_util.create_model(**weights.create_model_kwargs(hp), engine=e); m.load_state_dict(sd)
    return e


def per_image(a, b):
    return " ".join(f"{q:.0e}" for q in np.abs(a - b).reshape(a.shape[0], -1).max(1))


def cmd_loops(args):
    cases = {"deblur": synth.make_case("deblur", 16, H, H, seed=100, ksize=61), "inpaint": synth.make_case("inpaint", 16, H, H, seed=100)}
    for prec in args.precisions.split(","):
        e0, e1 = mk(prec), mk(prec)
        for spec in args.cases.split(","):
            task, graph, nfe = spec.split(":"); graph, nfe = bool(int(graph)), int(nfe)
            case = cases[task]
            cfg = (restore.LoopConfig(task="deblur", iter_num=nfe, lambda_=7.0, zeta=0.3) if task == "deblur" else
                   restore.LoopConfig(task="inpaint", iter_num=nfe, noise_level_img=0.0, lambda_=1.0, zeta=1.0))

            def loop(e, sl, off, keep=None, out=None):
                return restore.restore_batch(e, cfg, case["y"][sl], k=None if case["k"] is None else case["k"][sl],
                                             mask=None if case["mask"] is None else case["mask"][sl], noise_source="device", seed=1234,
                                             image_offset=off, use_graph=graph, _cache=keep, out_f32=out)
            seq = np.concatenate([loop(e0, slice(0, 8), 0).numpy(), loop(e1, slice(8, 16), 8).numpy()])
            k0, k1 = {}, {}
            o0, o1 = e0.empty((8, 3, H, H)), e1.empty((8, 3, H, H))
            loop(e0, slice(0, 8), 0, keep=k0, out=o0); loop(e1, slice(8, 16), 8, keep=k1, out=o1)      # second enqueued while the first runs
            e0.sync(); e1.sync()
            con = np.concatenate([o0.numpy(), o1.numpy()])
            print(f"[{prec}] {task} graph={graph} nfe={nfe}: concurrent vs sequential per image: {per_image(con, seq)} | nan {int(np.isnan(con).sum())}", flush=True)
        e0.close(); e1.close()


def cmd_forwards(args):
    """--engines N (default 2): N engines, 8 images each.  With the fused GroupNorm hop of conv7 (Conv6Emit) an (image, co-block) wait set is
    up to 256 of the chip's 512 resident workgroups: two engines' sets fit side by side, three are not covered by that argument -- the
    probe reports either bitwise equality or the engine's own "waited too long" error (never a hang: the wait is bounded)."""
    n = args.engines
    x = np.random.default_rng(0).standard_normal((8 * n, 3, H, H)).astype(np.float32)
    t = np.full(8, 500)
    for prec in args.precisions.split(","):
        es = [mk(prec) for _ in range(n)]
        xs = [e.to_device(x[8 * i:8 * i + 8]) for i, e in enumerate(es)]
        refs = []
        for e, xi in zip(es, xs):
            r = e.unet_forward(xi, t); e.sync(); refs.append(r.numpy())
        outs = [e.empty(refs[0].shape) for e in es]
        worst, errors = 0.0, 0
        rounds = 20 if n <= 2 else 6
        for _ in range(rounds):
            try:
                for e, xi, o in zip(es, xs, outs): e.unet_forward(xi, t, out=o)
                for e in es: e.sync()
                worst = max([worst] + [float(np.abs(o.numpy() - r).max()) for o, r in zip(outs, refs)])
            except diffpir_amd.EngineError as ex:
                errors += 1
                print(f"[{prec}] engine error under concurrency: {str(ex)[:160]}", flush=True)
                for e in es:
                    try: e.sync()
                    except diffpir_amd.EngineError: pass
        print(f"[{prec}] {n} engines, concurrent forwards vs sequential: max|diff| {worst:.3e}, rounds with an engine error {errors}/{rounds}", flush=True)
        for e in es: e.close()


def cmd_fft(args):
    case = synth.make_case("deblur", 8, H, H, seed=100, ksize=61)
    x = np.random.default_rng(0).standard_normal((8, 3, H, H)).astype(np.float32)
    for prec in args.precisions.split(","):
        eu, ef = mk(prec), mk(prec)                     # eu: UNet forwards, ef: FFT prox only
        xu, t = eu.to_device(x), np.full(8, 500)
        ru = eu.unet_forward(xu, t); eu.sync(); ru_np = ru.numpy()
        ou = eu.empty(ru_np.shape)
        y, k = ef.to_device(case["y"]), ef.to_device(case["k"])
        pre = sr.pre_calculate(y, k, 1, engine=ef)
        x0h = (case["gt"] * 2 - 1).astype(np.float32)
        b0 = ef.empty(x0h.shape); b0.copy_from(x0h)
        ef._check(ef.lib.dpir_prox_fft_apply(ef.h, pre[0].spectra.handle, b0.ptr, 7e-7, 1.0)); ef.sync()
        ref_p = b0.numpy()
        bufs = [ef.empty(x0h.shape) for _ in range(40)]
        for b in bufs: b.copy_from(x0h)
        ef.sync()
        for _ in range(3): eu.unet_forward(xu, t, out=ou)
        for b in bufs: ef._check(ef.lib.dpir_prox_fft_apply(ef.h, pre[0].spectra.handle, b.ptr, 7e-7, 1.0))
        eu.sync(); ef.sync()
        dp = [float(np.abs(b.numpy() - ref_p).max()) for b in bufs]
        print(f"[{prec}] UNet under concurrent FFT: max|diff| {np.abs(ou.numpy() - ru_np).max():.3e};  FFT prox under concurrent UNet: "
              f"worst {max(dp):.3e}, #changed {sum(d > 0 for d in dp)}/40 (prox output scale {np.abs(ref_p).max():.2f})", flush=True)
        eu.close(); ef.close()


AGGRESSORS = {"conv6": (8, 128, 128, 256, 256, 3, 0, 1, 2), "conv5": (8, 256, 128, 256, 256, 1, 0, 0, 2), "conv2": (8, 128, 128, 256, 256, 3, 0, 1, 0)}


def cmd_victim(args):
    dbg = _lib.load_debug()
    ea, ef = diffpir_amd.Engine(0), diffpir_amd.Engine(0)
    conv = AGGRESSORS[args.aggressor]
    stop, ms = [False], C.c_double(0)

    def spin():
        while not stop[0]:
            dbg.dpir_debug_conv_bench(ea.h, *conv, 200, C.byref(ms))
    th = threading.Thread(target=spin); th.start()
    try:
        time.sleep(0.05)
        bad = C.c_ulonglong(0)
        for lds, thr, blocks in ((64 * 1024, 256, 512), (8 * 1024, 64, 4096)):
            dbg.dpir_debug_victim(ef.h, lds, thr, blocks, 2000, 30, C.byref(bad))
            print(f"[{args.aggressor}] parked LDS {lds} B x {blocks} blocks: LDS mismatches {bad.value & 0xffffffff}, register mismatches {bad.value >> 32}", flush=True)
        for mode, nm in enumerate(("v_add_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f32")):
            dbg.dpir_debug_victim_alu(ef.h, mode, 4096, 20000, 20, C.byref(bad))
            print(f"[{args.aggressor}] dependent {nm} chains: wrong threads {bad.value}", flush=True)
        for nm, fn in (("built with v_pk_*_f32", dbg.dpir_debug_victim_fft_pk), ("built without", dbg.dpir_debug_victim_fft_nopk)):
            fn(ef.h, 4096, 200, 20, C.byref(bad))
            print(f"[{args.aggressor}] register FFT {nm}: non-reproducible threads {bad.value} of {4096 * 64 * 20}", flush=True)
    finally:
        stop[0] = True; th.join(); ea.close(); ef.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["loops", "forwards", "fft", "victim"])
    ap.add_argument("--precisions", default="f16x3,f32")
    ap.add_argument("--cases", default="deblur:1:2,deblur:1:8,deblur:0:30,inpaint:1:30,deblur:1:30")
    ap.add_argument("--aggressor", default="conv6", choices=sorted(AGGRESSORS))
    ap.add_argument("--engines", type=int, default=2, help="forwards: number of engines sharing the GPU")
    a = ap.parse_args()
    {"loops": cmd_loops, "forwards": cmd_forwards, "fft": cmd_fft, "victim": cmd_victim}[a.what](a)
