"""CPU, world_size=2, gloo: the N>1 path of the bench (image sharding + one all-gather of results).
The per-image work is replaced by a deterministic function of the GLOBAL image index (exactly how the engine
keys its device noise), so the test checks that the gathered result equals the single-process result for even
and ragged splits."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffpir_amd import dist as ddist


def fake_restore(lo, hi):
    out = np.empty((hi - lo, 4, 4, 3), np.uint8)
    for i in range(lo, hi):
        out[i - lo] = np.random.default_rng(i).integers(0, 256, (4, 4, 3), dtype=np.uint8)
    return torch.from_numpy(out)


def _worker(rank, world, port, n_images, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ddist.shard_range(n_images, rank, world)
    full = ddist.all_gather_results(fake_restore(lo, hi), n_images, rank, world)
    ret[rank] = full.numpy().tobytes()
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_images", [8, 7])
def test_sharded_results_equal_single_process(n_images):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_images, ret), nprocs=world, join=True)
    ref = fake_restore(0, n_images).numpy().tobytes()
    assert ret[0] == ref and ret[1] == ref


def _worker_api(rank, world, port, ret):
    """The module-level flow bench.py / the YAML driver use: init -> attach -> barrier / max_over_ranks / gather of host rows -> shutdown."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      DIFFPIR_COLLECTIVE="gloo")
    r, _, w = ddist.init()                      # DIFFPIR_COLLECTIVE overrides the default (rccl needs a GPU engine)
    ddist.attach(None)
    ddist.barrier()
    mx = ddist.max_over_ranks(1.0 + rank)
    lo, hi = ddist.shard_range(5, r, w)
    rows = np.stack([np.arange(lo, hi, dtype=np.float64), -np.arange(lo, hi, dtype=np.float64)], 1)
    allm = ddist.all_gather_results(rows, 5, r, w)
    ret[rank] = (mx, allm.tolist(), ddist.collective_name())
    ddist.shutdown()


def test_module_level_flow_on_gloo_world_2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_api, args=(world, _free_port(), ret), nprocs=world, join=True)
    want = [[float(i), -float(i)] for i in range(5)]
    for rnk in range(world):
        mx, rows, name = ret[rnk]
        assert mx == 2.0 and rows == want and "gloo" in name


def test_shard_ranges_partition_the_batch():
    for n in (1, 7, 16, 256):
        for w in (1, 2, 4, 8):
            r = [ddist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
