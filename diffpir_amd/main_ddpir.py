"""YAML-driven batched driver with the reference's config surface (main_ddpir.py:127-169, 172-599), on the engine.

    python -m diffpir_amd.main_ddpir --opt <config.yaml> [--synthetic N] [--device 0] [--num-gpus G]

--num-gpus G (or the YAML key `num_gpus`) re-launches the driver as G processes, one per GPU (torch.distributed.run); every batch
is then block-partitioned over the ranks (dist.restore_sharded) and one all-gather of the uint8 results ends it.

Accepts the reference's configs/*.yaml unchanged.  What it keeps from the reference driver: the derived fields
(noise_level_img/255, sigma = max(0.001, .), kernel_std), the per-task lambda/zeta sweeps (main_ddpir.py:548-580),
the per-image degradation recipe (main_ddpir.py:46-117) and the batch PSNR log line.  What it does differently:
the loop body is one engine call (restore.restore_batch), images are read / written with PIL when available (cv2
is not installed here), LPIPS is skipped (package absent; `calc_LPIPS` is honoured as "not available"), and with no
dataset / checkpoint on disk it falls back to synthetic ground truth and synthetic weights, saying so in the log.
"""
from __future__ import annotations

import argparse
import glob
import logging
import os
import time

import numpy as np
import yaml

from . import restore, synth, script_util, utils_model, weights, degrade as dgr
from .engine import Engine
from .utils_inpaint import mask_generator

log = logging.getLogger("diffpir_amd")


class Config:
    def __init__(self, dictionary):
        for k, v in dictionary.items():
            setattr(self, k, Config(v) if isinstance(v, dict) else v)

    def get(self, k, default=None):
        return getattr(self, k, default)


def parse_config(path):
    with open(path, "r") as f:
        config = Config(yaml.safe_load(f))
    config.opt = path
    config.noise_level_img = config.noise_level_img / 255.0              # main_ddpir.py:138
    config.noise_level_model = config.noise_level_img                    # :140
    config.sigma = max(0.001, config.noise_level_img)                    # :141
    if config.task == "deblur":
        config.kernel_std = 3.0 if config.blur_mode == "Gaussian" else 0.5   # :151
    if config.task == "inpaint":
        assert config.generate_mode in ["DiffPIR", "repaint", "vanilla"]
    for k, d in dict(sr_mode="blur", inIter=1, gamma=0.01, sf=1).items():
        if not hasattr(config, k):
            setattr(config, k, d)
    return config


def loop_config(config, lambda_, zeta) -> restore.LoopConfig:
    return restore.LoopConfig(task=config.task, iter_num=config.iter_num, noise_level_img=config.noise_level_img,
                              lambda_=lambda_, zeta=zeta, eta=config.eta, guidance_scale=config.guidance_scale,
                              sf=config.sf, sr_mode=config.sr_mode, inIter=config.inIter, gamma=config.gamma,
                              skip_type=config.skip_type, num_train_timesteps=config.num_train_timesteps,
                              beta_start=config.beta_start, beta_end=config.beta_end, generate_mode=config.generate_mode,
                              model_output_type=config.model_output_type, sub_1_analytic=config.sub_1_analytic,
                              ddim_sample=config.ddim_sample, iter_num_U=config.iter_num_U,
                              noise_init_img=config.get("noise_init_img", "max"),
                              skip_noise_model_t=bool(config.get("skip_noise_model_t", False)))


def sweeps(config):
    """main_ddpir.py:548-580."""
    if config.task == "sr":
        return [(config.lambda_ * i, config.zeta) for i in range(2, 13)]
    if config.task == "deblur":
        return [(config.lambda_ * 7, config.zeta * 3)]
    return [(config.lambda_ * 1, config.zeta * 1)]


def load_images(config, n_synth):
    """uint8 [N,H,W,3] (what util.imread_uint returns, main_ddpir.py:84) + names.  testsets/<testset_name>/*.png via PIL when
    present, else synthetic."""
    d = os.path.join(config.get("cwd", "") or "", "testsets", config.testset_name)
    paths = sorted(glob.glob(os.path.join(d, "*.png")) + glob.glob(os.path.join(d, "*.jpg")))
    if paths and not n_synth:
        try:
            from PIL import Image
            imgs = [np.asarray(Image.open(p).convert("RGB"), np.uint8) for p in paths]
            return np.stack(imgs), [os.path.basename(p) for p in paths]
        except Exception as ex:   # pragma: no cover
            log.warning("cannot read %s (%s); using synthetic images", d, ex)
    n = n_synth or 16
    log.info("no dataset on disk: using %d synthetic 256x256 ground-truth images", n)
    f = synth.smooth_images(n, 256, 256, 42)
    return np.ascontiguousarray((f * 255.0).round().astype(np.uint8).transpose(0, 2, 3, 1)), [f"synth_{i:04d}.png" for i in range(n)]


_MAT_CACHE = {}


def load_reference_kernels(config, name):
    """kernels/<name>.mat of the reference tree (main_ddpir.py:54, 71: hdf5storage.loadmat(...)['kernels']) -> object array [1, n] of 2-D
    float arrays, or None when the file is not under <cwd>/kernels.  The two v5 files are read with scipy; Levin09.mat is v7.3 (HDF5):
    h5py if importable, else a `<name>.npz` beside it (tools/convert_mat_v73.py writes it under an interpreter that has h5py)."""
    path = os.path.join(config.get("cwd", "") or "", "kernels", name + ".mat")
    if path in _MAT_CACHE:
        return _MAT_CACHE[path]
    cell = None
    if os.path.exists(path):
        try:
            import scipy.io
            cell = scipy.io.loadmat(path)["kernels"]
        except NotImplementedError:
            try:
                import h5py
                with h5py.File(path, "r") as f:
                    refs = f["kernels"]
                    ks = [np.array(f[refs[i, 0]]).T for i in range(refs.shape[0])]     # MATLAB is column-major: h5py sees the transpose
            except ImportError:
                side = os.path.splitext(path)[0] + ".npz"
                if not os.path.exists(side):
                    raise RuntimeError(f"{path} is a MATLAB v7.3 (HDF5) file and h5py is not importable here: convert it once with "
                                       f"`<python with h5py> tools/convert_mat_v73.py {path}` (writes {side})")
                z = np.load(side)
                ks = [z[f"k{i}"] for i in range(len(z.files))]
            cell = np.empty((1, len(ks)), dtype=object)
            for i, kk in enumerate(ks):
                cell[0, i] = kk
    _MAT_CACHE[path] = cell
    return cell


def make_operators(config, n, idx0, H, W):
    """Per-image PSFs / masks of CustomDataset.__getitem__ (main_ddpir.py:52-110): a few hundred scalar operations per image
    under the numpy RNG, kept on the host so that they stay bit-identical to the reference's.  Returns (k, mask) numpy or None."""
    k = mask = None
    if config.task == "deblur" and not config.get("use_DIY_kernel", True):
        # main_ddpir.py:69-72: k_index = 0 of kernels/Levin09.mat, float32, the same PSF for every image
        cell = load_reference_kernels(config, "Levin09")
        if cell is None:
            raise FileNotFoundError("use_DIY_kernel: false needs kernels/Levin09.mat under `cwd` (main_ddpir.py:71); it is not there")
        k = np.broadcast_to(cell[0, 0].astype(np.float32), (n, 1) + cell[0, 0].shape).copy()
    elif config.task == "deblur":
        ks = []
        for b in range(n):
            np.random.seed(seed=(idx0 + b) * 10)                            # main_ddpir.py:59
            if config.blur_mode == "Gaussian":
                ks.append(synth.gaussian_psf(config.kernel_size, config.kernel_std * np.abs(np.random.rand() * 2 + 1)))
            else:
                ks.append(synth.motion_psf(config.kernel_size, seed=idx0 + b))
        k = np.stack(ks)[:, None].astype(np.float32)
    elif config.task == "sr":
        # main_ddpir.py:53-56: kernels_bicubicx234.mat[0, sf - 2] (index 2 for sf >= 5)
        cell = load_reference_kernels(config, "kernels_bicubicx234")
        if cell is not None:
            kk = cell[0, config.sf - 2 if config.sf < 5 else 2].astype(np.float64).astype(np.float32)
        else:
            if config.sf != 4:
                raise FileNotFoundError(f"task sr, sf = {config.sf}: kernels/kernels_bicubicx234.mat is not under `cwd` and the built-in stand-in exists for sf = 4 only")
            if not _MAT_CACHE.get("warned"):
                log.warning("kernels/kernels_bicubicx234.mat not found under cwd=%r: using the ANALYTIC x4 bicubic PSF stand-in (differs from the "
                            "reference's file by up to 1.9e-3 per tap, 3 %% of the peak) -- results will not match a reference run", config.get("cwd", ""))
                _MAT_CACHE["warned"] = True
            kk = synth.bicubic_psf_x4()
        k = np.broadcast_to(kk, (n, 1) + kk.shape).copy()
    elif config.get("load_mask", False):
        # main_ddpir.py:103-104: ONE mask image for every test image, `imread_uint(mask_path, n_channels).astype(bool)`
        mask = load_mask_image(config, H, W)
        mask = np.broadcast_to(mask, (n,) + mask.shape[1:]).copy()
    else:
        gen = mask_generator(config.mask_type, config.mask_len_range, config.mask_prob_range)
        mask = np.concatenate([gen((1, 3, H, W)) for _ in range(n)], 0)
    return k, mask


def load_mask_image(config, H, W):
    """`load_mask: true` (main_ddpir.py:103-104): util.imread_uint(mask_path, n_channels) -> astype(bool): any non-zero byte keeps the pixel.
    Returns uint8 {0, 1} [1, 3, H, W].  A relative mask_path is looked up under `cwd` like the reference's other inputs; a mask whose size differs
    from the images is an error (the reference would fail at `img_H * mask`)."""
    path = config.get("mask_path", "")
    if not path:
        raise ValueError("load_mask: true needs mask_path (main_ddpir.py:104)")
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(config.get("cwd", ""), path)
    if path.endswith(".npy"):
        m = np.load(path)
    else:
        from PIL import Image
        m = np.asarray(Image.open(path).convert("RGB"), np.uint8)          # imread_uint(n_channels=3): grey images are replicated (utils_image.py:190-199)
    if m.ndim == 2:
        m = np.repeat(m[:, :, None], 3, axis=2)
    if m.shape[:2] != (H, W) or m.shape[2] != 3:
        raise ValueError(f"mask {path}: shape {m.shape}, the test images are {H} x {W} x 3 (main_ddpir.py:109 multiplies them elementwise)")
    return np.ascontiguousarray((m != 0).astype(np.uint8).transpose(2, 0, 1)[None])


def main(argv=None):
    import sys
    ap = argparse.ArgumentParser()
    ap.add_argument("--opt", type=str, required=True, help="Path to option YAML file.")
    ap.add_argument("--synthetic", type=int, default=0, help="use N synthetic images instead of the testset")
    ap.add_argument("--device", type=int, default=None, help="HIP ordinal (default: LOCAL_RANK, else 0)")
    ap.add_argument("--max-sweeps", type=int, default=0)
    ap.add_argument("--num-gpus", type=int, default=0, help="shard every batch over this many GPUs (one process per GPU)")
    ap.add_argument("--dist-backend", default="rccl", help="result collective: rccl (the C ABI's ncclAllGather binding, default) | nccl | gloo (torch.distributed)")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    config = parse_config(args.opt)
    num_gpus = args.num_gpus or int(config.get("num_gpus", 1) or 1)
    if num_gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU: re-launch under torch.distributed.run (rendezvous on 127.0.0.1)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={num_gpus}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29541"), "-m", "diffpir_amd.main_ddpir"] + list(argv if argv is not None else sys.argv[1:])
        return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    from . import dist as ddist
    rank, local_rank, world = ddist.init(args.dist_backend)
    np.random.seed(config.seed)
    eng = Engine(args.device if args.device is not None else local_rank)
    eng.set_precision(str(config.get("engine_precision", "f16x3")))      # before load_state_dict: selects the weight packing
    if config.generate_mode == "DPS_y0":
        eng.enable_grad()                  # main_ddpir.py:236-239: DPS_y0 is the one mode that differentiates through the network
    ddist.attach(eng)                      # rccl: ncclCommInitRank on this engine's device

    model_config = dict(model_path=os.path.join(config.get("cwd", "") or "", "model_zoo", config.model_name + ".pt"),
                        num_channels=128, num_res_blocks=1, attention_resolutions="16") \
        if config.model_name == "diffusion_ffhq_10m" else \
        dict(model_path=os.path.join(config.get("cwd", "") or "", "model_zoo", config.model_name + ".pt"),
             num_channels=256, num_res_blocks=2, attention_resolutions="8,16,32")
    margs = utils_model.create_argparser(model_config).parse_args([])
    model, diffusion = script_util.create_model_and_diffusion(
        **script_util.args_to_dict(margs, script_util.model_and_diffusion_defaults().keys()), engine=eng)
    if os.path.exists(margs.model_path):
        model.load_state_dict(weights.load_checkpoint(margs.model_path))
    else:
        log.info("checkpoint %s not found: using synthetic weights (results are not meaningful images)", margs.model_path)
        model.load_state_dict(weights.synth_state_dict("ffhq" if config.model_name == "diffusion_ffhq_10m" else "imagenet256"))

    imgs, names = load_images(config, args.synthetic)
    use_graph = bool(config.get("engine_graph", True))
    noise = config.get("engine_noise", "device")
    host_gen = None
    if noise == "host":
        # ONE generator for the whole run, like the reference's global torch RNG (main_ddpir.py:131): successive batches
        # continue the stream instead of replaying it
        import torch
        host_gen = torch.Generator().manual_seed(config.seed)
    results = []
    cache = {}
    import torch
    # every batch's sharding is checked BEFORE any work starts (a ragged last batch smaller than the world used to raise after all earlier
    # batches had been computed, losing the run)
    for i0 in range(0, len(imgs), config.batch_size):
        ddist.check_dps_sharding(eng, loop_config(config, *sweeps(config)[0]), min(config.batch_size, len(imgs) - i0), world)
    for si, (lambda_, zeta) in enumerate(sweeps(config)):
        if args.max_sweeps and si >= args.max_sweeps:
            break
        cfg = loop_config(config, lambda_, zeta)
        if rank == 0:
            log.info("eta:%s, zeta:%s, lambda:%s, guidance_scale:%s", config.eta, zeta, lambda_, config.guidance_scale)
        psnrs, psnrs_y, t0, n = [], [], time.time(), 0
        for i0 in range(0, len(imgs), config.batch_size):
            gt = imgs[i0:i0 + config.batch_size]                           # uint8 [n,H,W,3]
            n_b, H, W = gt.shape[0], gt.shape[1], gt.shape[2]
            k_all, mask_all = make_operators(config, n_b, i0, H, W)      # every rank draws the same operators (seeded numpy)
            lo, hi = ddist.shard_range(n_b, rank, world)                 # this rank's images of the batch
            per_img = np.zeros((0, 2))
            u8_local = eng.empty((hi - lo, H, W, 3), np.uint8)          # engine-owned result buffer (send side of the all-gather)
            drawn = None
            dps_mode = cfg.generate_mode in ("DPS_y0", "DPS_yt")
            ddist.check_dps_sharding(eng, cfg, n_b, world)
            dps_nf = None
            if noise == "host" and dps_mode:
                # the DPS modes draw in their own order (init, then per step the sampler's draw [and the y_t draw]); restore_batch pulls
                # them through noise_fn.  The GLOBAL batch's tensor is drawn every time and this rank keeps its image rows, so results do
                # not depend on the number of ranks; a rank with an empty shard still advances the shared generator.
                dps_nf = lambda shape: torch.randn((n_b,) + tuple(shape[1:]), generator=host_gen).numpy()[lo:hi]
                if hi == lo:
                    _, steps, _ = restore._steps(cfg)
                    for shp in restore.dps_host_noise_shapes(cfg, steps, 0, H, W):
                        dps_nf(shp)
            elif noise == "host":
                # EVERY rank draws the global batch's noise, also a rank whose shard is empty (ragged last batch with fewer images
                # than ranks): the shared generator must advance identically everywhere or later batches depend on the world size
                _, steps, _ = restore._steps(cfg)
                nf = lambda shape: torch.randn(tuple(shape), generator=host_gen).numpy()
                full = restore.draw_host_noise(nf, steps, (n_b, 3, H, W), cfg.eta != 0, repaint=cfg.generate_mode == "repaint")
                drawn = [None if a is None else (np.ascontiguousarray(a[lo:hi]) if a.ndim == 4 else np.ascontiguousarray(a[:, lo:hi])) for a in full]
            if hi > lo:
                sl = slice(lo, hi)
                # degradation on the device (dpir_degrade): blur / down-sampling / masking + AWGN, device Philox noise keyed by the
                # global image index
                y, ops = dgr.degrade(eng, config.task, gt[sl], k=None if k_all is None else k_all[sl], mask=None if mask_all is None else mask_all[sl],
                                     noise_level_img=config.noise_level_img, sf=config.sf, sr_mode=config.sr_mode, seed=config.seed + 1,
                                     image_offset=i0 + lo)
                out_f32 = restore.restore_batch(eng, cfg, y, k=ops.get("k"), mask=None if dps_mode else ops.get("mask"), noise_source=noise,
                                                predrawn=drawn, noise_fn=dps_nf, seed=config.seed, image_offset=i0 + lo, use_graph=use_graph, out_u8=u8_local, _cache=cache,
                                                skip_dead_final_eval=bool(config.get("engine_skip_dead_final_eval", False)))
                psnr_i, psnr_y_i = dgr.metrics(eng, out_f32, ops["gt"])            # dpir_metrics: per-image PSNR / PSNR-Y
                per_img = np.stack([psnr_i, psnr_y_i], 1).astype(np.float64)
            u8 = ddist.all_gather_results(u8_local, n_b, rank, world, engine=eng)           # the one collective of the path
            allm = ddist.all_gather_results(per_img.reshape(-1, 2), n_b, rank, world, engine=eng)      # per-image PSNR rows (host numpy)
            p, p_y = float(np.mean(allm[:, 0])), float(np.mean(allm[:, 1]))
            psnrs.append(p * n_b)
            psnrs_y.append(p_y * n_b)
            n += n_b
            if rank == 0:
                log.info("batch%4d--> PSNR: %.4fdB", i0 // config.batch_size + 1, p)
        dt = time.time() - t0
        if rank == 0:
            log.info("-----------> Average PSNR(RGB) of (%s) scale factor: (%d), sigma: (%.3f): %.4f dB  [%.2f images/s on %d GPU(s)]",
                     config.testset_name, config.sf, config.noise_level_model, sum(psnrs) / n, n / dt, world)
            log.info("-----------> Average PSNR(Y) of (%s) scale factor: (%d), sigma: (%.3f): %.4f dB",
                     config.testset_name, config.sf, config.noise_level_model, sum(psnrs_y) / n)
        results.append(sum(psnrs) / n)
    ddist.shutdown()
    eng.close()
    return results


if __name__ == "__main__":
    main()
