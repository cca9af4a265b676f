"""Image sharding across the GPUs of a node (one process per GPU) and the single result collective.

The restoration path has no cross-image coupling: a batch is block-partitioned over ranks, every
rank runs its own loop, and the only exchange is one all-gather of the results (RCCL over xGMI with backend
"nccl"; "gloo" on CPU for tests).  Device noise is keyed by the GLOBAL image index (`image_offset`), so outputs
do not depend on the number of ranks.
"""
from __future__ import annotations

import os
from typing import Tuple


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_images: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; the first (n_images % world) ranks get one extra image."""
    base, rem = divmod(n_images, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init(backend: str = "nccl"):
    """One process per GPU (launched by torch.distributed.run): joins the process group when WORLD_SIZE > 1.
    backend "nccl" is RCCL over xGMI on ROCm; "gloo" (host staging) is what the single-GPU / CPU tests use."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    # DIFFPIR_FORCE_DIST=1 joins the group at WORLD_SIZE == 1 too, so that the RCCL code path (init, barrier, all_gather,
    # all_reduce) can be exercised on a single-GPU box exactly as the multi-GPU launch runs it (tests/test_gpu_dist.py)
    if (world > 1 or os.environ.get("DIFFPIR_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of one scalar (the bench's elapsed time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def init_rccl(engine, rank: int, world: int, port: int = None):
    """Bind the result collective straight to RCCL through the C ABI (dpir_comm_init / dpir_allgather_results): rank 0 creates
    the 128-byte ncclUniqueId and ships it to the other ranks over a TCP socket on MASTER_ADDR (stdlib; rendezvous plumbing
    only), then every rank joins the communicator on its engine's device.  Selected with DIFFPIR_COLLECTIVE=rccl."""
    import ctypes as C
    import socket
    import time
    buf = C.create_string_buffer(128)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = port or int(os.environ.get("MASTER_PORT", "29533")) + 17
    if rank == 0:
        engine._check(engine.lib.dpir_comm_unique_id(buf))
        if world > 1:
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            for _ in range(world - 1):
                c, _a = srv.accept()
                c.sendall(buf.raw)
                c.close()
            srv.close()
    else:
        for attempt in range(600):
            try:
                c = socket.create_connection((addr, port), timeout=5)
                break
            except OSError:
                time.sleep(0.1)
        else:
            raise RuntimeError("RCCL rendezvous: rank 0 is not listening")
        data = b""
        while len(data) < 128:
            chunk = c.recv(128 - len(data))
            if not chunk:
                raise RuntimeError("RCCL rendezvous: short read of the unique id")
            data += chunk
        c.close()
        buf.raw = data
    engine._check(engine.lib.dpir_comm_init(engine.h, world, rank, buf))
    engine.rccl = True


def all_gather_results(local, n_images: int, rank: int, world: int, engine=None):
    """local: torch tensor [n_local, ...] (uint8 NHWC results).  Returns [n_images, ...] on every rank.
    Shards may differ by one image, so each is padded to the largest shard for the collective.
    engine with an RCCL communicator (init_rccl): ncclAllGather through the C ABI on the engine stream; otherwise
    torch.distributed (backend nccl == RCCL, or gloo in the single-GPU / CPU tests)."""
    import torch
    import torch.distributed as dist
    if engine is not None and getattr(engine, "rccl", False):
        sizes = [shard_range(n_images, r, world) for r in range(world)]
        mx = max(hi - lo for lo, hi in sizes)
        pad = local
        if local.shape[0] < mx:
            pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], 0)
        pad = pad.contiguous()
        recv = torch.empty((world,) + tuple(pad.shape), dtype=pad.dtype, device=pad.device)
        torch.cuda.current_stream(pad.device).synchronize()          # `pad` may have been produced on torch's stream
        engine._check(engine.lib.dpir_allgather_results(engine.h, pad.data_ptr(), recv.data_ptr(), pad.numel() * pad.element_size()))
        engine.sync()
        return torch.cat([recv[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local
    sizes = [shard_range(n_images, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], 0)
    pad = pad.contiguous()
    if dist.get_backend() == "gloo" and pad.is_cuda:       # gloo stages through the host (tests on one GPU)
        pad = pad.cpu()
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)
    return out.to(local.device)


def restore_sharded(engine, cfg, y, k=None, mask=None, labels=None, *, rank: int, world: int, image_offset: int = 0, seed: int = 0,
                    use_graph: bool = True, noise_source: str = "device", host_noise=None, cache: dict = None,
                    skip_dead_final_eval: bool = False):
    """One GLOBAL batch restored by `world` ranks: rank r runs images [lo, hi) of the batch on its own GPU (its own engine,
    its own replayed graph, no collective inside the loop) and ONE all-gather of the uint8 results ends the batch
    (SURVEY.md 8e).  y / k / mask / labels are the global batch (numpy, host); every rank slices its block.

    Device noise is keyed by the global image index (image_offset + lo + b), so the gathered result is independent of
    `world`.  Host noise (parity mode) must be pre-drawn for the GLOBAL batch: host_noise = (init, n1, n2[, nrp]) as returned
    by restore.draw_host_noise for the global shape; each rank uploads its image slice.
    Returns (uint8 [n_images, H, W, 3] on every rank as a torch tensor on this rank's device, local fp32 DeviceArray)."""
    import numpy as np
    import torch
    from . import restore
    n = y.shape[0]
    lo, hi = shard_range(n, rank, world)
    sl = slice(lo, hi)
    H, W = y.shape[2] * cfg.sf, y.shape[3] * cfg.sf
    out_u8 = torch.empty((hi - lo, H, W, 3), dtype=torch.uint8, device=f"cuda:{engine.device}")
    if hi > lo:
        kw = dict(k=None if k is None else np.ascontiguousarray(k[sl]), mask=None if mask is None else np.ascontiguousarray(mask[sl]),
                  labels=None if labels is None else np.asarray(labels)[sl], seed=seed, image_offset=image_offset + lo,
                  use_graph=use_graph, out_u8=out_u8, _cache=cache, skip_dead_final_eval=skip_dead_final_eval)
        if noise_source == "host":
            drawn = [None if a is None else (np.ascontiguousarray(a[sl]) if a.ndim == 4 else np.ascontiguousarray(a[:, sl])) for a in host_noise]
            out_f32 = restore.restore_batch(engine, cfg, np.ascontiguousarray(y[sl]), noise_source="host", predrawn=drawn, **kw)
        else:
            out_f32 = restore.restore_batch(engine, cfg, np.ascontiguousarray(y[sl]), noise_source="device", **kw)
        engine.sync()
    else:
        out_f32 = None
    return all_gather_results(out_u8, n, rank, world, engine=engine), out_f32
