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


@pytest.mark.parametrize("world,n_images", [(2, 8), (2, 7), (4, 7), (4, 3)])
def test_sharded_results_equal_single_process(world, n_images):
    """Even and ragged splits at world 2, and world 4 with a ragged 7-image batch (shards 2, 2, 2, 1) and with FEWER images than ranks
    (3 over 4: one rank's shard is empty and it still has to join the collective)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_images, ret), nprocs=world, join=True)
    ref = fake_restore(0, n_images).numpy().tobytes()
    assert all(ret[r] == ref for r in range(world))


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


class _FakeLib:
    """Stands in for libdiffpir_hip.so's two communicator entry points: records what dpir_comm_init receives."""

    def __init__(self, fail_id):
        self.fail_id, self.joined = fail_id, None

    def dpir_comm_unique_id(self, buf):
        if self.fail_id:
            return -5
        buf.raw = bytes(range(128))
        return 0

    def dpir_comm_init(self, h, world, rank, buf):
        self.joined = (world, rank, bytes(buf.raw))
        return 0

    def dpir_last_error(self, h):
        return b"librccl.so cannot be loaded"


class _FakeEngine:
    h = None

    def __init__(self, fail_id=False):
        self.lib = _FakeLib(fail_id)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"engine error {rc}")


@pytest.mark.parametrize("rank0_fails", [False, True])
def test_rccl_rendezvous_is_collective_and_bounded(rank0_fails, monkeypatch):
    """dist.init_rccl (the TCP exchange of the ncclUniqueId) at world = 3 with in-process fake engines: every rank receives rank 0's
    id -- or, when rank 0 cannot create one, every rank raises TOGETHER (the fallback to torch.distributed must be collective, round-3
    advisor finding) instead of spinning on a dead port; a stray connection that does not introduce itself is ignored."""
    import threading
    import time
    port = _free_port()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("DIFFPIR_RENDEZVOUS_TIMEOUT", "20")
    engines = [_FakeEngine(fail_id=rank0_fails and r == 0) for r in range(3)]
    errors = {}

    def run(r):
        try:
            ddist.init_rccl(engines[r], r, 3, port=port)
        except Exception as ex:
            errors[r] = ex
    ths = [threading.Thread(target=run, args=(r,)) for r in range(3)]
    t0 = time.monotonic()
    ths[0].start()
    time.sleep(0.2)
    stray = socket.create_connection(("127.0.0.1", port), timeout=5)       # says nothing useful
    stray.sendall(b"GET / HT")
    stray.close()
    for t in ths[1:]:
        t.start()
    for t in ths:
        t.join(30)
    assert not any(t.is_alive() for t in ths) and time.monotonic() - t0 < 15
    if rank0_fails:
        assert sorted(errors) == [0, 1, 2]
        assert all(e.lib.joined is None for e in engines)
    else:
        assert not errors, errors
        assert [e.lib.joined[:2] for e in engines] == [(3, 0), (3, 1), (3, 2)]
        assert all(e.lib.joined[2] == bytes(range(128)) for e in engines)


def test_rccl_rendezvous_times_out_when_rank0_never_listens(monkeypatch):
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("DIFFPIR_RENDEZVOUS_TIMEOUT", "1")
    with pytest.raises(RuntimeError, match="not listening"):
        ddist.init_rccl(_FakeEngine(), 1, 2, port=_free_port())


def test_dps_y0_sharding_needs_the_in_loop_allreduce():
    """DPS_y0 couples the images of a batch through the batch-wide norm (round-3 advisor finding): sharded runs are refused unless the
    engine carries the C ABI's RCCL communicator (the loop all-reduces the squared sums) and every rank has images."""
    from diffpir_amd import restore

    class E:
        rccl = False
    cfg = restore.LoopConfig(task="sr", sf=4, generate_mode="DPS_y0")
    ddist.check_dps_sharding(E(), cfg, 8, 1)                                   # one rank: nothing to exchange
    ddist.check_dps_sharding(E(), restore.LoopConfig(task="sr", sf=4, generate_mode="DPS_yt"), 8, 2)      # the norm cancels in DPS_yt
    with pytest.raises(NotImplementedError, match="rccl"):
        ddist.check_dps_sharding(E(), cfg, 8, 2)
    E.rccl = True
    ddist.check_dps_sharding(E(), cfg, 8, 2)
    with pytest.raises(NotImplementedError, match="every rank"):
        ddist.check_dps_sharding(E(), cfg, 1, 2)
    # DPS_yt with the communicator attached all-reduces the same sums: an empty shard would hang the other ranks
    yt = restore.LoopConfig(task="sr", sf=4, generate_mode="DPS_yt")
    ddist.check_dps_sharding(E(), yt, 8, 2)
    with pytest.raises(NotImplementedError, match="every rank"):
        ddist.check_dps_sharding(E(), yt, 1, 2)
    E.rccl = False
    ddist.check_dps_sharding(E(), yt, 1, 2)                                    # torch.distributed fallbacks: no in-loop collective
