"""bench.py -- restored images/sec @100 NFE, 256x256 (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE full restoration (100 NFE: init -> 100 x (UNet -> FFT prox -> re-noise) -> u8 output) of one batch of
synthetic inputs per GPU.  Workload (--config): BASELINE configs[1] PER GPU at every N (c2: FFHQ topology, 256x256 Gaussian deblur, 61x61 PSF,
B = 16 per GPU) -- weak scaling with the per-GPU work fixed, so the N = 1, 2, 4, 8 values of one sweep are comparable; --config c4 is BASELINE's
multi-GPU configs[3] (FFHQ topology, motion deblur, B = 32 per GPU = 256 over 8 GPUs) and --config c5 configs[4] (512x512 class-conditional topology, x4
SISR, B = 8 per GPU = 64 over 8).  Inputs (y, k) are resident in HBM before
the timed region; the loop is a replayed per-step hipGraph with device-side Philox noise; with N > 1 the batch is sharded by image
(weak scaling, no data-path collective) and the u8 results are all-gathered over RCCL -- bound through the C ABI
(dpir_allgather_results), engine-owned buffers, engine stream -- inside the timed region (diffpir_amd.dist, the function the
YAML driver uses).  Weights are synthetic.

Rank 0 adds to the ONE JSON line (single-GPU runs; each part can be switched off):
  roofline       the dominant kernel class (3x3 convolutions): algorithmic FLOPs over the HIP-event duration of every
                 launch of the class in an instrumented pass (events on the engine stream), against the dense MFMA peak of
                 the arithmetic used; plus the WHOLE UNet step against the same peak (`unet_step_frac`, the north-star figure);
  roofline_prox  the FFT data-fidelity step (3 launches) against its algorithmic HBM bytes (2.50 MB / image);
  config_c3      BASELINE configs[2]: ImageNet-256 topology, x4 SISR (bicubic PSF), B = 32, 100 NFE -- one timed restoration;
  alt_precision  configs[1] once more in the exact-fp32-MFMA mode, with the output difference between the two modes;
  cpu_baseline   the oracle (torch-CPU restatement of the reference path) on the host cores: BASELINE configs[0] in full
                 (FFHQ inpainting, 20 NFE, B = 1) with the UNet / prox / re-noise split; kind "port".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL; must be in the environment before the HIP runtime starts
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16X3_TFLOPS = 2500.0 / 3   # dense f16 MFMA peak / 3 MFMAs per fp32-equivalent product
PEAK_HBM_TBS = 8.0               # HBM3E spec
PROX_BYTES_PER_IMAGE = {1: 2_497_536, 4: 2_761_728}     # sf = 1 / sf = 4
# HBM-side bytes per launch of the roofline kernel classes from the committed PMC passes (bench.py cannot run rocprofv3 on
# itself): profiles/pmc_traffic.json is written by tools/pmc_traffic.py from profiles/r03/*_pmc_{FETCH,WRITE}_SIZE.txt (commands:
# tools/gpu_prof_round.sh; FETCH_SIZE doubled per the gfx950 note).
PMC_TRAFFIC = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))) if os.path.exists(os.path.join(ROOT, "profiles", "pmc_traffic.json")) else {}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=["c2", "c4", "c5"],
                    help="BASELINE workload per GPU: c2 = configs[1] (default at every --gpus: fixed per-GPU work), c4 = configs[3] (32 per GPU), c5 = configs[4]; "
                         "--batch / --size / --task / --model override single fields")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (weak scaling)")
    ap.add_argument("--nfe", type=int, default=100)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--task", default=None, choices=["deblur", "inpaint", "sr"])
    ap.add_argument("--blur", default=None, choices=["gaussian", "motion"])
    ap.add_argument("--model", default=None, choices=["ffhq", "imagenet256", "imagenet512"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c3", action="store_true", help="skip the BASELINE configs[2] object")
    ap.add_argument("--c3-batch", type=int, default=32)
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"],
                    help="arithmetic of the conv GEMMs for the headline value: f16x3 = fp32 operands split into f16 hi+lo, three f16 MFMAs "
                         "per product, fp32 accumulate (fp32-equivalent results, see DESIGN.md); f32 = v_mfma_f32_32x32x2_f32")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary measurement in the other precision mode")
    ap.add_argument("--cpu-threads", type=int, default=32, help="torch CPU threads for the oracle (all 256 host threads "
                    "oversubscribe MKL-DNN at B=1: 79 s/NFE measured vs ~1 s/NFE at 32)")
    a = ap.parse_args()
    preset = {"c2": dict(model="ffhq", task="deblur", blur="gaussian", batch=16, size=256),
              "c4": dict(model="ffhq", task="deblur", blur="motion", batch=32, size=256),
              "c5": dict(model="imagenet512", task="sr", blur="gaussian", batch=8, size=512)}[a.config or "c2"]
    a.preset = a.config or "c2"
    a.exact_preset = all(getattr(a, kk) in (None, v) for kk, v in preset.items()) and a.nfe == 100
    for kk, v in preset.items():
        if getattr(a, kk) is None:
            setattr(a, kk, v)
    return a


def make_problem(restore, synth, task, B, H, nfe, seed, blur="gaussian"):
    if task == "deblur":
        return (restore.LoopConfig(task="deblur", iter_num=nfe, lambda_=7.0, zeta=0.3),
                synth.make_case("deblur", B, H, H, seed=seed, ksize=61, blur=blur))
    if task == "inpaint":
        return (restore.LoopConfig(task="inpaint", iter_num=nfe, noise_level_img=0.0, lambda_=1.0, zeta=1.0),
                synth.make_case("inpaint", B, H, H, seed=seed))
    return restore.LoopConfig(task="sr", iter_num=nfe, lambda_=6.0, zeta=0.25, sf=4), synth.make_case("sr", B, H, H, seed=seed, sf=4)


def conv_roofline(eng, B, H, precision, model_name):
    """Instrumented pass: HIP events around every launch of the 3x3 convolution class, 3 forwards of the same batch."""
    x = eng.to_device(np.random.default_rng(0).standard_normal((B, 3, H, H)).astype(np.float32))
    t = np.full(B, 500)
    lab = np.arange(B) % 1000 if model_name == "imagenet512" else None
    o6 = eng.unet_forward(x, t, lab)
    for _ in range(2):
        eng.unet_forward(x, t, lab, out=o6)
    eng.sync()
    n_wall = 5
    t0 = time.perf_counter()
    for _ in range(n_wall):
        eng.unet_forward(x, t, lab, out=o6)
    eng.sync()
    wall_ms = (time.perf_counter() - t0) / n_wall * 1e3           # un-instrumented forward (eager launches)
    eng.prof_enable(True)
    eng.prof_reset()
    n_pass = 3
    for _ in range(n_pass):
        eng.unet_forward(x, t, lab, out=o6)
    eng.sync()
    prof = eng.prof_read()
    eng.prof_enable(False)
    ms, cnt = prof["conv3x3"]
    fl = eng.unet_flops(H, H, 0) * B * n_pass
    achieved = fl / (ms * 1e-3) / 1e12
    peak = {"f32": PEAK_FP32_MFMA_TFLOPS, "f16x1": 2500.0}.get(precision, PEAK_F16X3_TFLOPS)
    kern = ("conv2_mfma_kernel<3x3> (v_mfma_f32_32x32x2_f32, exact fp32)" if precision == "f32" else
            "conv6_mfma_kernel<3x3, X1> (one v_mfma_f32_32x32x16_f16 per product, hi planes only; two workgroups per CU)" if precision == "f16x1" else
            "3x3 class: conv7_mfma_kernel<geometry, f16x3> (64 co x 128 px per wave, weights straight into registers; the plane-emitting variant of every "
            "ResBlock's conv1 also carries GroupNorm statistics, the per-image wait, FiLM + SiLU and the f16 split of conv2's operands -- that work and "
            "wait are inside this class's time since round 4, which is why the class fraction fell while the UNet step rose; + conv6_mfma_kernel for the "
            "split-K launches of the 8x32 geometry; + conv8_fused_kernel, the 128 -> 6 output layer with GroupNorm + SiLU + split in its LDS fill, "
            "v_mfma_f32_16x16x32_f16, since round 5) "
            "(3 x v_mfma_f32_32x32x16_f16 per fp32-equivalent product; operands pre-split by act_split*_kernel; two workgroups per CU)")
    step_fl = eng.unet_flops(H, H) * B
    tr = PMC_TRAFFIC.get(f"{model_name}_B{B}_{H}_{precision}")
    return {"bound": "mfma", "kernel": kern, "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None if tr is None else tr["bytes_per_launch"],
            "traffic_source": None if tr is None else tr["source"],
            "launches": int(cnt), "avg_launch_ms": round(ms / max(cnt, 1), 4), "flops_per_launch_avg": fl / max(cnt, 1),
            "unet_forward_ms": round(wall_ms, 3), "unet_step_tflops": round(step_fl / wall_ms / 1e9, 2),
            "unet_step_frac": round(step_fl / wall_ms / 1e9 / peak, 4),
            "unet_step_note": "whole UNet forward (every kernel and launch gap, eager launches, no event instrumentation): "
                              "algorithmic FLOPs of the step / wall time / the same MFMA peak -- the north-star 'UNet step' figure",
            "class_ms_per_forward_instrumented": {kk: round(v[0] / n_pass, 3) for kk, v in prof.items() if v[1]}}


def prox_roofline(eng, case, B, H, sf):
    """dpir_prox_fft_apply (x0 -> data-fidelity step -> x0) timed with HIP events on the engine stream.  `us_per_apply` (what `achieved` uses) is
    dpir_prox_fft_apply_timed's graph mode: 60 applies captured as ONE hipGraph, one event before and one after its replay -- device time per apply with
    the launch boundaries between its kernels, which is how dpir_run_loop runs the step (the kernel durations of the rocprofv3 trace under profiles/ add
    up to the same figure).  `us_per_apply_event_pairs` is the round 1-5 method (an event pair around EVERY eager apply: +2.5 us of record overhead per
    apply), kept so that the series stays comparable."""
    import ctypes as C
    from diffpir_amd import utils_sisr as sr
    y, k = eng.to_device(case["y"]), eng.to_device(case["k"])
    pre = sr.pre_calculate(y, k, sf)
    x0 = eng.to_device(case["gt"] * 2 - 1)
    h = pre[0].spectra.handle
    for _ in range(5):
        eng._check(eng.lib.dpir_prox_fft_apply(eng.h, h, x0.ptr, 0.05, 1.0))
    eng.sync()
    usg = C.c_float()
    best = 1e30
    for _ in range(3):
        eng._check(eng.lib.dpir_prox_fft_apply_timed(eng.h, h, x0.ptr, 0.05, 1.0, 60, 1, C.byref(usg)))
        best = min(best, usg.value)
    n = 100
    eng.prof_enable(True)
    eng.prof_reset()
    for _ in range(n):
        eng._check(eng.lib.dpir_prox_fft_apply(eng.h, h, x0.ptr, 0.05, 1.0))
    eng.sync()
    ms, cnt = eng.prof_read()["fft_prox"]
    eng.prof_enable(False)
    us_ev = ms / n * 1e3
    us = best
    byts = PROX_BYTES_PER_IMAGE[sf] * B * (H * H) / 65536
    ach = byts / (us * 1e-6) / 1e12
    tr = PMC_TRAFFIC.get(f"fftprox_sf{sf}_B{B}_{H}")        # committed PMC passes: sum of the apply kernels' fabric-side bytes per launch
    traffic = None if tr is None else int(sum(v["bytes_per_launch"] for v in tr["kernels"].values()))
    wave = H in (256, 512) and sf in (1, 2, 4)
    kern = (f"rfft4_rows + cfft4_cols(solve) + irfft4_rows (fft4.hip: one wave per {H}-point transform, column-major half spectrum" if wave else
            "rfft_rows + cfft_cols(solve) + irfft_rows (fft2.hip two-pass register kernels, row-major half spectrum" if H == 64 and sf in (1, 2, 4) else "fft.hip c2c path (")
    return {"bound": "hbm", "kernel": kern + ("" if sf == 1 else f", alias-grouped slots, sf = {sf}") + ")",
            "achieved": round(ach, 4), "peak": PEAK_HBM_TBS, "unit": "TB/s", "frac": round(ach / PEAK_HBM_TBS, 4),
            "us_per_apply": round(us, 2), "us_per_apply_event_pairs": round(us_ev, 2), "frac_event_pairs": round(byts / (us_ev * 1e-6) / 1e12 / PEAK_HBM_TBS, 4),
            "timing": "60 applies as one captured hipGraph between two events (best of 3 replays)",
            "algorithmic_bytes": int(byts), "batch": B, "sf": sf, "launches_per_apply": 3,
            "traffic": traffic, "traffic_source": None if tr is None else tr["source"]}


def cpu_baseline_c1(weights, threads):
    """BASELINE configs[0] in full on the host: FFHQ 256x256 box inpainting, 20 NFE, B = 1, through the oracle (a torch-CPU
    restatement of the reference loop that reproduces live-reference outputs bit for bit, tests/test_oracle_golden.py)."""
    import torch as th
    from oracle import unet_oracle as uo, diffpir_oracle as do
    from diffpir_amd import synth
    th.set_num_threads(max(1, min(threads, os.cpu_count())))
    hp = uo.ffhq_hp()
    sd = {kk: th.from_numpy(v) for kk, v in weights.synth_state_dict("ffhq", 0).items()}
    case = synth.make_case("inpaint", 1, 256, 256, seed=42)
    split = {"unet": 0.0, "prox": 0.0, "renoise": 0.0}

    def timed(mod, name, key):
        f = getattr(mod, name)

        def w(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            split[key] += time.perf_counter() - t0
            return r
        setattr(mod, name, w)
        return f
    saved = [(uo, "unet_forward", timed(uo, "unet_forward", "unet")), (do, "prox_mask", timed(do, "prox_mask", "prox")),
             (do, "renoise", timed(do, "renoise", "renoise"))]
    try:
        g = th.Generator().manual_seed(0)
        nf = lambda like: th.randn(like.shape, generator=g, dtype=th.float32)
        ocfg = do.LoopConfig("inpaint", 20, 0.0, 1.0, 1.0)
        t0 = time.perf_counter()
        do.restore(sd, hp, ocfg, th.from_numpy(case["y"]), mask=th.from_numpy(case["mask"]).float(), noise_fn=nf)
        total = time.perf_counter() - t0
    finally:
        for mod, name, f in saved:
            setattr(mod, name, f)
    return {"value": round(1.0 / total, 6), "unit": "images/s @20 NFE (configs[0])", "cores": th.get_num_threads(),
            "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle (torch-CPU fp32 restatement of the reference loop; reproduces live-reference outputs bit for bit), BASELINE "
                      f"configs[0] in full: FFHQ topology, 256x256 box inpainting, 20 NFE, B=1: {total:.1f} s",
            "seconds_per_nfe": {k: round(v / 20, 4) for k, v in split.items()},
            "equivalent_images_per_s_at_100_nfe": round(1.0 / (total * 5), 6)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_dist.py runs the N = 2 launch line on a ONE-GPU box): all ranks on one device, gloo instead of RCCL
    # (RCCL refuses two ranks on one GPU).  Never set by the driver.
    if "DIFFPIR_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["DIFFPIR_BENCH_DEVICE"])
    backend = os.environ.get("DIFFPIR_BENCH_BACKEND", "rccl")      # rccl = the C ABI's RCCL binding (default); gloo only in the one-GPU tests
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch
    import diffpir_amd
    from diffpir_amd import restore, synth, script_util, weights, dist as ddist
    torch.cuda.set_device(local_rank)
    ddist.init(backend)         # a no-op at WORLD_SIZE == 1; "rccl": the communicator is created by attach(engine) below

    def load(model_name, precision):
        e = diffpir_amd.Engine(local_rank)
        e.set_precision(precision)
        hp = weights.model_hp(model_name)
        m = script_util.create_model(**weights.create_model_kwargs(hp), engine=e)
        m.load_state_dict(weights.synth_state_dict(hp, 0))
        return e

    eng = load(args.model, args.precision)
    ddist.attach(eng)           # ncclCommInitRank on this engine's device (unique id exchanged over MASTER_ADDR:MASTER_PORT)
    B, H = args.batch, args.size
    cfg, case = make_problem(restore, synth, args.task, B, H, args.nfe, 100 + rank, args.blur)
    labels = (np.arange(B) + rank * B) % 1000 if args.model == "imagenet512" else None
    y = eng.to_device(case["y"])
    k = None if case["k"] is None else eng.to_device(case["k"])
    mask = None if case["mask"] is None else eng.to_device(case["mask"])
    out_f32 = eng.empty((B, 3, H, H))
    out_u8 = eng.empty((B, H, H, 3), np.uint8)       # engine-owned send buffer of the one collective
    keep = {}

    def one_step():
        # weak scaling: every rank restores its own B images (global indices [rank*B, (rank+1)*B)), then the ONE collective of
        # the path: all-gather of the uint8 results
        restore.restore_batch(eng, cfg, y, k=k, mask=mask, labels=labels, noise_source="device", seed=1234, image_offset=rank * B,
                              use_graph=not args.no_graph, out_f32=out_f32, out_u8=out_u8, _cache=keep)
        eng.sync()
        ddist.all_gather_results(out_u8, B * world, rank, world, engine=eng)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        ddist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0, device=None if backend == "gloo" else f"cuda:{local_rank}")
    coll_name = ddist.collective_name()
    ranks_info = ddist.gather_rank_info(eng, local_rank)     # every rank's device / RCCL version, moved by the same collective as the results
    value = B * world * args.steps / elapsed
    headline_out = out_f32.numpy()           # before the instrumented passes below reuse the buffer

    extras = rank == 0 and world == 1
    roofline = conv_roofline(eng, B, H, args.precision, args.model) if rank == 0 else None
    prox = None
    if extras and args.task in ("deblur", "sr"):
        prox = prox_roofline(eng, case, B, H, cfg.sf)
        if B != 64 and args.task == "deblur":
            rng = np.random.default_rng(7)          # timing only: any y / PSF of the right shapes
            c64 = {"y": rng.random((64, 3, H, H), dtype=np.float32), "gt": rng.random((64, 3, H, H), dtype=np.float32),
                   "k": np.repeat(synth.gaussian_psf(61, 3.0)[None, None], 64, 0)}
            prox["at_batch_64"] = {kk: v for kk, v in prox_roofline(eng, c64, 64, H, 1).items() if kk in ("achieved", "frac", "us_per_apply", "us_per_apply_event_pairs", "frac_event_pairs")}
            # configs[4]'s data step alone (the "HBM-bound FFT prox stress"): 512 x 512, sf = 4, its per-GPU shard of 8 images; timing only
            c5 = {"y": rng.random((8, 3, 128, 128), dtype=np.float32), "gt": rng.random((8, 3, 512, 512), dtype=np.float32),
                  "k": np.repeat(synth.gaussian_psf(25, 2.0)[None, None], 8, 0)}
            prox["c5_512_sf4_batch_8"] = {kk: v for kk, v in prox_roofline(eng, c5, 8, 512, 4).items()
                                          if kk in ("kernel", "achieved", "frac", "us_per_apply", "us_per_apply_event_pairs", "frac_event_pairs", "algorithmic_bytes")}

    # ---- the fused data step of the loop (eps -> x0 prologue + FFT prox + re-noise/Philox epilogue: 3 launches per step)
    if extras and prox is not None and args.task == "deblur":
        def fft_class_ms(nfe):
            c = restore.LoopConfig(task="deblur", iter_num=nfe, lambda_=7.0, zeta=0.3)
            eng.prof_enable(True); eng.prof_reset()
            restore.restore_batch(eng, c, y, k=k, noise_source="device", seed=1234, use_graph=False, out_f32=out_f32)
            eng.sync()
            ms = eng.prof_read()["fft_prox"][0]
            eng.prof_enable(False)
            return ms
        fft_class_ms(4)
        us = (fft_class_ms(24) - fft_class_ms(4)) / 20 * 1e3          # the difference removes pre_calculate (same prof class)
        fb = B * (4 * 3 * H * H * 4 + 3 * H * (H // 2 + 1) * 8 + H * (H // 2 + 1) * 4)     # x, eps, x (again), x' + FBFy + F2B
        prox["fused_data_step"] = {"what": "one loop step's data side: x0 = clamp(c1 x - c2 eps) in the row-FFT prologue, spectral solve, "
                                           "re-noise + Philox in the inverse row-FFT epilogue; x0 / noise never touch HBM",
                                   "us_per_step": round(us, 2), "algorithmic_bytes": int(fb), "achieved": round(fb / (us * 1e-6) / 1e12, 4),
                                   "frac": round(fb / (us * 1e-6) / 1e12 / PEAK_HBM_TBS, 4), "launches_per_step": 3}
        if roofline is not None and world == 1:
            # The UNet step AS THE HOT PATH RUNS IT (captured step graph, hoisted FiLM table): the timed headline loop's wall time per NFE minus the
            # measured data step.  Everything else of a batch (init, pre_calculate, finalize, graph launches) stays inside the figure, so it is
            # an upper bound of the forward's time; `unet_forward_ms` above is the same forward launched eagerly through dpir_unet_forward
            # (per-call time embedding + FiLM projection, ~800 host launches), the figure rounds 1-4 reported.
            per_nfe_ms = elapsed / args.steps / args.nfe * 1e3
            in_loop = per_nfe_ms - us * 1e-3 * (args.nfe - 1) / args.nfe
            step_fl = eng.unet_flops(H, H) * B
            roofline["unet_step_in_loop_ms"] = round(in_loop, 3)
            roofline["unet_step_in_loop_frac"] = round(step_fl / in_loop / 1e9 / roofline["peak"], 4)
            roofline["unet_step_in_loop_note"] = ("timed headline loop (hipGraph replay, 100 NFE) per NFE minus the measured fused data step; init / "
                                                  "pre_calculate / finalize of the batch are NOT subtracted")

    # ---- the steps either side of the loop (device degradation synthesis, device metrics) and the YAML driver's
    # end-to-end rate degrade -> 100-NFE loop -> metrics on the headline batch
    f1 = None
    if extras and args.task == "deblur":
        from diffpir_amd import degrade as dgr
        gt_d = eng.to_device(np.ascontiguousarray((np.clip(case["gt"], 0, 1) * 255).round().astype(np.uint8).transpose(0, 2, 3, 1)))
        yd, _ops = dgr.degrade(eng, "deblur", gt_d, k=k, noise_level_img=cfg.noise_level_img, seed=5)
        dgr.metrics(eng, out_f32, gt_d)
        eng.sync()
        reps = 5
        ta = time.perf_counter()
        for _ in range(reps):
            dgr.degrade(eng, "deblur", gt_d, k=k, noise_level_img=cfg.noise_level_img, seed=5, out=yd)
        eng.sync()
        t_deg = (time.perf_counter() - ta) / reps
        ta = time.perf_counter()
        for _ in range(reps):
            dgr.metrics(eng, out_f32, gt_d)
        t_met = (time.perf_counter() - ta) / reps
        ta = time.perf_counter()
        dgr.degrade(eng, "deblur", gt_d, k=k, noise_level_img=cfg.noise_level_img, seed=5, out=yd)
        restore.restore_batch(eng, cfg, yd, k=k, noise_source="device", seed=1234, use_graph=not args.no_graph, out_f32=out_f32, out_u8=out_u8, _cache=keep)
        dgr.metrics(eng, out_f32, gt_d)
        eng.sync()
        t_e2e = time.perf_counter() - ta
        f1 = {"what": "dpir_degrade (61x61 wrap-around blur of the uint8 ground truth in fp64 + AWGN) and dpir_metrics (PSNR / PSNR-Y incl. the "
                      "D2H of the per-image sums) on the headline batch; end_to_end = degrade -> loop -> metrics, the YAML driver's per-batch work",
              "degrade_ms": round(t_deg * 1e3, 3), "metrics_ms": round(t_met * 1e3, 3),
              "end_to_end_images_per_s": round(B / t_e2e, 4), "end_to_end_ms": round(t_e2e * 1e3, 1)}

    # ---- the gradient-based mode (DPS_y0, x4 SISR): forward + p_sample + Resizer^T + UNet backward per NFE
    dps = None
    if extras and not args.no_alt and args.model == "ffhq":
        eg = diffpir_amd.Engine(local_rank)
        eg.set_precision(args.precision)
        eg.enable_grad()
        mg = script_util.create_model(**weights.create_model_kwargs(weights.model_hp("ffhq")), engine=eg)
        mg.load_state_dict(weights.synth_state_dict(weights.model_hp("ffhq"), 0))
        peak_d = PEAK_FP32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16X3_TFLOPS
        fl_img = eg.unet_flops(256, 256)                      # one forward; the input-gradient pass runs the same contractions transposed
        nfe_d = 6

        def dps_at(Bd):
            cd = synth.make_case("sr", Bd, 256, 256, seed=400, sf=4)
            cfgd = restore.LoopConfig(task="sr", iter_num=nfe_d, lambda_=6.0, zeta=0.25, sf=4, sr_mode="cubic", generate_mode="DPS_y0")
            yd_ = eg.to_device(cd["y"])
            restore.restore_batch(eg, cfgd, yd_, noise_source="device", seed=1)          # allocations
            ta = time.perf_counter()
            od = restore.restore_batch(eg, cfgd, yd_, noise_source="device", seed=1)
            eg.sync()
            td = (time.perf_counter() - ta) / (nfe_d - 1)
            ach = 2.0 * fl_img * Bd / td / 1e12
            return {"batch": Bd, "ms_per_nfe": round(td * 1e3, 2), "images_per_s_at_100_nfe": round(Bd / (td * 100), 4),
                    "finite": bool(np.isfinite(od.numpy()).all()),
                    "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak_d, 1), "unit": "TFLOP/s", "frac": round(ach / peak_d, 4),
                                 "flops_per_nfe": 2.0 * fl_img * Bd}}
        d8, d16 = dps_at(8), dps_at(16)
        dps = {"what": "generate_mode DPS_y0 (main_ddpir.py:370-373, 434-438), FFHQ topology, x4 SISR 64^2 -> 256^2: per NFE one UNet forward "
                       f"({args.precision}), p_sample, residual norm, Resizer^T and one UNet input-gradient pass (dgrad on the same MFMA kernels as the forward: "
                       "conv7 with the flipped / transposed weight pack and a run-time power-of-two scale on dY); eager launches.  roofline: the "
                       "algorithmic FLOPs of forward + input-gradient pass (2 x the forward's: every contraction runs once each way) over the whole NFE "
                       "(every kernel and launch gap) against the same MFMA peak as the headline",
               "nfe": nfe_d, **{k: v for k, v in d8.items()}, "at_batch_16": d16}
        eg.close()

    # ---- secondary measurements in the other arithmetic modes (same inputs, weights, device noise and graph path)
    def other_mode(other):
        eng2 = load(args.model, other)
        y2 = eng2.to_device(case["y"]); k2 = None if case["k"] is None else eng2.to_device(case["k"])
        m2 = None if case["mask"] is None else eng2.to_device(case["mask"])
        o2 = eng2.empty((B, 3, H, H)); keep2 = {}

        def step2():
            restore.restore_batch(eng2, cfg, y2, k=k2, mask=m2, labels=labels, noise_source="device", seed=1234, image_offset=0,
                                  use_graph=not args.no_graph, out_f32=o2, _cache=keep2)
            eng2.sync()
        step2()
        ta = time.perf_counter()
        step2()
        tb = time.perf_counter() - ta
        a_out, b_out = headline_out, o2.numpy()
        gt = case["gt"] * 2 - 1
        pa, pb = restore.psnr_batch(a_out * 2 - 1, gt), restore.psnr_batch(b_out * 2 - 1, gt)
        rf2 = conv_roofline(eng2, B, H, other, args.model)
        eng2.close()
        return {"precision": other, "value": round(B / tb, 4), "unit": "images/s", "ms_per_step": round(tb * 1e3, 2),
                "max_abs_diff_vs_headline_output": float(np.abs(a_out - b_out).max()),
                "psnr_headline_dB": round(pa, 5), "psnr_alt_dB": round(pb, 5), "abs_dpsnr_dB": round(abs(pa - pb), 6),
                "roofline_frac": rf2["frac"], "unet_step_frac": rf2["unet_step_frac"], "peak": rf2["peak"]}
    alt = reduced = None
    if extras and not args.no_alt:
        alt = other_mode("f16x3" if args.precision == "f32" else "f32")
        alt["note"] = ("exact-fp32 MFMA kernels on the same inputs, weights and device noise; parity of BOTH modes with the "
                       "reference is what tests/ -m gpu assert -- this entry only shows that the two modes agree")
        # the opt-in reduced-precision mode (f16 operands, ONE MFMA per product, fp32 accumulate, fp32 GroupNorm /
        # softmax -- the reference's own use_fp16 recipe).  NOT the headline: it does not meet the fp32 parity bar layer by layer
        # (8.7e-4 relative per forward); its quality contract is the measured PSNR change, reported here.
        reduced = other_mode("f16x1")
        reduced["note"] = ("opt-in engine_precision: f16x1 (set_precision('f16x1')); roofline fractions against the dense f16 MFMA peak; "
                           "quality contract = abs_dpsnr_dB against the fp32-equivalent headline run on the same inputs")
    eng_closed = False

    # ---- BASELINE configs[2]: ImageNet-256 topology, x4 SISR, B = 32, 100 NFE
    c3 = None
    if extras and not args.no_c3 and args.preset == "c2" and (args.model, args.task) == ("ffhq", "deblur"):
        eng.close(); eng_closed = True                       # free the FFHQ workspace before the 40 GB ImageNet one
        e3 = load("imagenet256", args.precision)
        B3 = args.c3_batch
        cfg3, case3 = make_problem(restore, synth, "sr", B3, 256, args.nfe, 300)
        y3, k3 = e3.to_device(case3["y"]), e3.to_device(case3["k"])
        o3 = e3.empty((B3, 3, 256, 256)); keep3 = {}
        warm = restore.LoopConfig(task="sr", iter_num=3, lambda_=6.0, zeta=0.25, sf=4)      # captures both step graphs
        restore.restore_batch(e3, warm, y3, k=k3, noise_source="device", seed=1, use_graph=not args.no_graph, out_f32=o3, _cache=keep3)
        e3.sync()
        ta = time.perf_counter()
        restore.restore_batch(e3, cfg3, y3, k=k3, noise_source="device", seed=1234, use_graph=not args.no_graph, out_f32=o3, _cache=keep3)
        e3.sync()
        tb = time.perf_counter() - ta
        fin = bool(np.isfinite(o3.numpy()).all())
        rf3 = conv_roofline(e3, B3, 256, args.precision, "imagenet256")
        c3 = {"workload": f"configs[2]: imagenet256 topology 256x256 sr x4 (bicubic PSF, FFT prox), {args.nfe} NFE, batch {B3}, device Philox noise, "
                          f"hipGraph={'off' if args.no_graph else 'on'}", "value": round(B3 / tb, 4), "unit": "images/s", "steps": 1,
              "ms_per_step": round(tb * 1e3, 1), "finite": fin, "roofline": rf3,
              "roofline_prox": prox_roofline(e3, case3, B3, 256, 4)}
        e3.close()

    cpu = cpu_baseline_c1(weights, args.cpu_threads) if extras and not args.no_cpu_baseline else None

    if rank == 0:
        names = {"c2": "configs[1]" if world == 1 else f"configs[1] per GPU (weak scaling: batch {B} on each of {world} GPUs)", "c4": f"configs[3] (batch {B} per GPU; BASELINE quotes 256 images over 8 GPUs)",
                 "c5": f"configs[4] (batch {B} per GPU; BASELINE quotes 64 images over 8 GPUs)"}
        cfg_tag = (names[args.preset] if args.exact_preset else
                   "configs[2] topology/task (reduced batch or NFE)" if (args.model, args.task) == ("imagenet256", "sr") else
                   f"BASELINE {names[args.preset].split(' ')[0]} family, non-default flags")
        psf = {"deblur": "61x61 motion PSF (seeded random walk)" if args.blur == "motion" else "61x61 Gaussian PSF",
               "sr": "x4, bicubic PSF, FFT prox", "inpaint": "box mask"}[args.task]
        line = {"metric": f"restored images/sec @{args.nfe} NFE, {H}x{H}", "value": round(value, 4), "unit": "images/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None,
                "dtype": "f32" if args.precision == "f32" else "f32 (GEMMs as 3 x f16 MFMA on hi/lo-split fp32 operands, fp32 accumulate; exact-fp32-MFMA mode in alt_precision)",
                "data": "synthetic",
                "config": {"workload": f"{cfg_tag}: {args.model} topology {H}x{H} {args.task} ({psf}), {args.nfe} NFE, "
                                       f"batch {B}/GPU, device Philox noise, hipGraph={'off' if args.no_graph else 'on'}",
                           "global_batch": B * world, "nfe": args.nfe, "sharding": f"images x{world}, all_gather(u8) of results",
                           "collective": coll_name, "ranks": ranks_info},
                "roofline": roofline, "roofline_prox": prox, "degrade_metrics": f1, "dps_y0": dps, "cpu_baseline": cpu, "config_c3": c3, "alt_precision": alt, "reduced_precision": reduced}
        print(json.dumps(line))
    ddist.shutdown()
    if not eng_closed:
        eng.close()


if __name__ == "__main__":
    main()
