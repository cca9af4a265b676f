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
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def all_gather_results(local, n_images: int, rank: int, world: int):
    """local: torch tensor [n_local, ...] (uint8 NHWC results).  Returns [n_images, ...] on every rank.
    Shards may differ by one image, so each is padded to the largest shard for the collective."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    sizes = [shard_range(n_images, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], 0)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)
