"""bench.py -- restored images/sec @100 NFE, 256x256 (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE full restoration (100 NFE: init -> 100 x (UNet -> FFT prox -> re-noise) -> u8 output) of one
batch of synthetic 256x256 inputs per GPU -- BASELINE config[1]: FFHQ topology, Gaussian deblur (61x61 PSF),
B=16 per GPU.  Inputs (y, k) are resident in HBM before the timed region; the loop is a replayed hipGraph with
device-side Philox noise; with N>1 the batch is sharded by image (weak scaling, no data-path collective) and the
u8 results are all-gathered over RCCL inside the timed region.  Weights are synthetic (no checkpoint offline).

The JSON line also carries
  roofline     -- the dominant kernel (3x3 implicit-GEMM conv on fp32 MFMA): algorithmic FLOPs / launch over the
                  HIP-event duration of every launch of that kernel class in an instrumented pass of the same
                  workload on the engine stream, against the 157.3 TF/s fp32-MFMA peak;
  cpu_baseline -- the oracle (CPU restatement of the reference path, torch-CPU fp32) timed on this host's cores on
                  a bounded sample (NFE steps at B=1), extrapolated to 100 NFE.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16X3_TFLOPS = 2500.0 / 3   # dense f16 MFMA peak / 3 MFMAs per fp32-equivalent product
# HBM-side bytes per launch of the roofline kernel, from the committed PMC passes (bench.py cannot run rocprofv3 on itself):
# f16x3: conv4_mfma_kernel (2 x 314570.6 + 155571.1) KB; f32: conv2/conv kernels (2 x 283 + 132) MB (profiles/r01/README.md)
PMC_TRAFFIC_BYTES_PER_LAUNCH = {"f16x3": int((2 * 314570.6 + 155571.1) * 1024), "f32": int((2 * 283 + 132) * 1e6)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (weak scaling)")
    ap.add_argument("--nfe", type=int, default=100)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--task", default="deblur", choices=["deblur", "inpaint", "sr"])
    ap.add_argument("--model", default="ffhq", choices=["ffhq", "imagenet256"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"],
                    help="arithmetic of the conv GEMMs for the headline value: f16x3 = fp32 operands split into f16 hi+lo, three f16 MFMAs "
                         "per product, fp32 accumulate (fp32-equivalent results, see DESIGN.md); f32 = v_mfma_f32_32x32x2_f32")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary measurement in the other precision mode")
    ap.add_argument("--cpu-nfe", type=int, default=6, help="NFE steps of the CPU oracle sample (B=1)")
    ap.add_argument("--cpu-threads", type=int, default=32, help="torch CPU threads for the oracle sample (all 256 host\n                    threads oversubscribe MKL-DNN at B=1: 79 s/NFE measured vs ~1-2 s/NFE at 32)")
    return ap.parse_args()


def synth_weights(model_name, seed=0):
    """Deterministic synthetic state dict in the reference key schema (product-side generator)."""
    from diffpir_amd import weights
    return weights.synth_state_dict(model_name, seed)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch
    import diffpir_amd
    from diffpir_amd import restore, synth, script_util, weights, dist as ddist
    torch.cuda.set_device(local_rank)
    ddist.init("nccl")          # RCCL over xGMI; a no-op at WORLD_SIZE == 1

    eng = diffpir_amd.Engine(local_rank)
    eng.set_precision(args.precision)
    B, H = args.batch, args.size
    hp = weights.model_hp(args.model)
    sd_np = weights.synth_state_dict(hp, 0)
    model = script_util.create_model(**weights.create_model_kwargs(hp), engine=eng)
    model.load_state_dict(sd_np)

    if args.task == "deblur":
        cfg = restore.LoopConfig(task="deblur", iter_num=args.nfe, lambda_=7.0, zeta=0.3)
        case = synth.make_case("deblur", B, H, H, seed=100 + rank, ksize=61)
    elif args.task == "inpaint":
        cfg = restore.LoopConfig(task="inpaint", iter_num=args.nfe, noise_level_img=0.0, lambda_=1.0, zeta=1.0)
        case = synth.make_case("inpaint", B, H, H, seed=100 + rank)
    else:
        cfg = restore.LoopConfig(task="sr", iter_num=args.nfe, lambda_=6.0, zeta=0.25, sf=4)
        case = synth.make_case("sr", B, H, H, seed=100 + rank, sf=4)
    y = eng.to_device(case["y"])
    k = None if case["k"] is None else eng.to_device(case["k"])
    mask = None if case["mask"] is None else eng.to_device(case["mask"])
    out_f32 = eng.empty((B, 3, H, H))
    out_u8 = torch.empty((B, H, H, 3), dtype=torch.uint8, device=f"cuda:{local_rank}")     # plumbing for RCCL
    keep = {}

    def one_step():
        # weak scaling: every rank restores its own B images (global indices [rank*B, (rank+1)*B)), then the ONE collective of
        # the path: all-gather of the uint8 results (diffpir_amd.dist, the same function the YAML driver uses)
        restore.restore_batch(eng, cfg, y, k=k, mask=mask, noise_source="device", seed=1234, image_offset=rank * B,
                              use_graph=not args.no_graph, out_f32=out_f32, out_u8=out_u8, _cache=keep)
        eng.sync()
        ddist.all_gather_results(out_u8, B * world, rank, world)

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
    elapsed = time.perf_counter() - t0
    elapsed = ddist.max_over_ranks(elapsed, device=f"cuda:{local_rank}")
    images = B * world * args.steps
    value = images / elapsed

    # ---- roofline of the dominant kernel, instrumented pass on rank 0 (same batch, same weights)
    roofline = None
    if rank == 0:
        x = eng.to_device(np.random.default_rng(0).standard_normal((B, 3, H, H)).astype(np.float32))
        t = np.full(B, 500)
        o6 = eng.unet_forward(x, t)
        eng.sync()
        eng.prof_enable(True)
        eng.prof_reset()
        n_pass = 3
        for _ in range(n_pass):
            eng.unet_forward(x, t, out=o6)
        eng.sync()
        prof = eng.prof_read()
        eng.prof_enable(False)
        ms, cnt = prof["conv3x3"]
        fl = eng.unet_flops(H, H, 0) * B * n_pass           # conv3x3 FLOPs of the instrumented passes
        achieved = fl / (ms * 1e-3) / 1e12
        peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16X3_TFLOPS
        kern = ("conv2_mfma_kernel<3x3> (v_mfma_f32_32x32x2_f32, exact fp32)" if args.precision == "f32"
                else "conv4_mfma_kernel<3x3> (3 x v_mfma_f32_32x32x16_f16 per fp32-equivalent product; operands pre-split by act_split4_kernel)")
        roofline = {"bound": "mfma", "kernel": kern,
                    "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": PMC_TRAFFIC_BYTES_PER_LAUNCH.get(args.precision) if (B, H, args.model) == (16, 256, "ffhq") else None,
                    "traffic_source": "profiles/r01/pmc_{FETCH,WRITE}_SIZE_prof_forward_*.txt: separate rocprofv3 --pmc passes over tools/prof_forward.py "
                                      "(same model, batch and kernels), FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, KB -> bytes, "
                                      "average over this kernel's launches of a forward",
                    "launches": int(cnt), "avg_launch_ms": round(ms / max(cnt, 1), 4),
                    "flops_per_launch_avg": fl / max(cnt, 1),
                    "unet_forward_ms": round(prof["unet_forward"][0] / n_pass, 3),
                    "unet_tflops": round(eng.unet_flops(H, H) * B * n_pass / (prof["unet_forward"][0] * 1e-3) / 1e12, 3),
                    "class_ms_per_forward": {kk: round(v[0] / n_pass, 3) for kk, v in prof.items() if v[1]}}

    # ---- secondary measurement in the other precision mode (same inputs, same graph path), rank 0, single GPU
    alt = None
    if rank == 0 and world == 1 and not args.no_alt:
        other = "f16x3" if args.precision == "f32" else "f32"
        eng2 = diffpir_amd.Engine(local_rank)
        eng2.set_precision(other)
        model2 = script_util.create_model(**weights.create_model_kwargs(hp), engine=eng2)
        model2.load_state_dict(sd_np)
        y2 = eng2.to_device(case["y"]); k2 = None if case["k"] is None else eng2.to_device(case["k"])
        m2 = None if case["mask"] is None else eng2.to_device(case["mask"])
        o2 = eng2.empty((B, 3, H, H)); keep2 = {}
        def step2():
            restore.restore_batch(eng2, cfg, y2, k=k2, mask=m2, noise_source="device", seed=1234, image_offset=0,
                                  use_graph=not args.no_graph, out_f32=o2, _cache=keep2)
            eng2.sync()
        step2()
        ta = time.perf_counter()
        step2()
        tb = time.perf_counter() - ta
        a_out, b_out = out_f32.numpy(), o2.numpy()
        gt = case["gt"] * 2 - 1
        alt = {"precision": other, "value": round(B / tb, 4), "unit": "images/s", "ms_per_step": round(tb * 1e3, 2),
               "max_abs_diff_vs_headline_output": float(np.abs(a_out - b_out).max()),
               "psnr_headline_dB": round(restore.psnr_batch(a_out * 2 - 1, gt), 5),
               "psnr_alt_dB": round(restore.psnr_batch(b_out * 2 - 1, gt), 5),
               "note": "f16x3 = operand-split f16 MFMA (x = hi + lo, 3 MFMAs per product, fp32 accumulate): per-layer error vs the "
                       "oracle equals the exact-fp32 kernels' (3e-6), see DESIGN.md"}
        eng2.close()

    # ---- CPU baseline: the oracle on the host cores, bounded sample, rank 0 only
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        import torch as th
        from oracle import unet_oracle as uo, diffpir_oracle as do
        ohp = uo.ffhq_hp() if args.model == "ffhq" else uo.imagenet256_hp()
        sd = {kk: th.from_numpy(v) for kk, v in sd_np.items()}
        nfe = max(2, args.cpu_nfe)
        ocfg = do.LoopConfig(cfg.task, nfe, cfg.noise_level_img, cfg.lambda_, cfg.zeta, sf=cfg.sf)
        g = th.Generator().manual_seed(0)
        nf = lambda like: th.randn(like.shape, generator=g, dtype=th.float32)
        yy = th.from_numpy(case["y"][:1])
        kk_ = None if case["k"] is None else th.from_numpy(case["k"][:1])
        mm = None if case["mask"] is None else th.from_numpy(case["mask"][:1]).float()
        th.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count())))
        tc = time.perf_counter()
        do.restore(sd, ohp, ocfg, yy, k=kk_, mask=mm, noise_fn=nf)
        tcpu = time.perf_counter() - tc
        per_nfe = tcpu / nfe
        cpu = {"value": round(1.0 / (per_nfe * args.nfe), 6), "unit": "images/s", "cores": th.get_num_threads(),
               "host_cores": os.cpu_count(), "kind": "port",
               "sample": f"oracle (torch-CPU fp32 restatement of the reference loop), B=1, {nfe} NFE "
               f"at {H}x{H} in {tcpu:.1f} s, extrapolated linearly to {args.nfe} NFE"}

    if rank == 0:
        cfg_tag = ("configs[1]" if (args.model, args.task, B, args.nfe, H) == ("ffhq", "deblur", 16, 100, 256) else
                   "configs[2] topology/task (reduced batch or NFE)" if (args.model, args.task) == ("imagenet256", "sr") else
                   "BASELINE configs[1] family, non-default flags")
        line = {"metric": f"restored images/sec @{args.nfe} NFE, {H}x{H}", "value": round(value, 4), "unit": "images/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "f32 (GEMMs as 3 x f16 MFMA on hi/lo-split fp32 operands, fp32 accumulate; exact-fp32-MFMA mode in alt_precision)",
                "data": "synthetic",
                "config": {"workload": f"{cfg_tag}: {args.model} topology {H}x{H} {args.task} "
                                       f"({'61x61 Gaussian PSF' if args.task == 'deblur' else args.task}), {args.nfe} NFE, "
                                       f"batch {B}/GPU, device Philox noise, hipGraph={'off' if args.no_graph else 'on'}",
                           "global_batch": B * world, "nfe": args.nfe, "sharding": f"images x{world}, all_gather(u8) of results"},
                "roofline": roofline, "cpu_baseline": cpu, "alt_precision": alt}
        print(json.dumps(line))
    ddist.shutdown()
    eng.close()


if __name__ == "__main__":
    main()
