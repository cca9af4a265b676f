"""Image sharding across the GPUs of a node (one process per GPU) and the single result collective.

The restoration path has no cross-image coupling: a batch is block-partitioned over ranks, every
rank runs its own loop, and the only exchange is one all-gather of the results.  Device noise is keyed by the GLOBAL
image index (`image_offset`), so outputs do not depend on the number of ranks.

Collective back-ends (`init(collective)`, overridden by DIFFPIR_COLLECTIVE):
  * "rccl"  (default, the product path): RCCL over xGMI bound through the C ABI (`dpir_comm_init`, `dpir_allgather_results`,
    `dpir_comm_barrier`, `dpir_comm_allreduce_max`; csrc/comm.cpp dlopens librccl).  Buffers are engine-owned DeviceArrays, the
    collective runs on the engine stream, and the rendezvous is one TCP exchange of the 128-byte ncclUniqueId on
    MASTER_ADDR:MASTER_PORT (the variables the one-process-per-GPU launcher exports).  No torch in the data path.
  * "nccl" / "gloo": torch.distributed -- kept as the rendezvous fallback and for the tests that run several ranks on ONE GPU or
    on CPU (RCCL refuses two ranks on one device).
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np

_state = {"mode": "single", "engine": None, "rank": 0, "world": 1}


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_images: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; the first (n_images % world) ranks get one extra image."""
    base, rem = divmod(n_images, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init(collective: str = "rccl"):
    """One process per GPU (launched by torch.distributed.run or any launcher that exports RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, local_rank, world).  With the default "rccl" collective the communicator is
    created by `attach(engine)` once the rank's engine exists; "nccl" / "gloo" join a torch.distributed process group here."""
    rank, local_rank, world = env_rank_world()
    collective = os.environ.get("DIFFPIR_COLLECTIVE", collective)
    if collective == "torch":
        collective = "nccl"
    # DIFFPIR_FORCE_DIST=1 runs the collective code path at WORLD_SIZE == 1 too, exactly as the multi-GPU launch runs it
    active = world > 1 or os.environ.get("DIFFPIR_FORCE_DIST") == "1"
    _state.update(mode="single", engine=None, rank=rank, world=world)
    if not active:
        return rank, local_rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if collective == "rccl":
        _state["mode"] = "rccl-pending"
    elif collective in ("nccl", "gloo"):
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(collective, rank=rank, world_size=world)
        _state["mode"] = "torch"
    else:
        raise ValueError(f"unknown collective {collective!r} (rccl | nccl | gloo)")
    return rank, local_rank, world


def attach(engine):
    """Bind this rank's engine to the group: creates the RCCL communicator on the engine's device (rccl mode)."""
    if _state["mode"] == "rccl-pending":
        try:
            init_rccl(engine, _state["rank"], _state["world"])
            _state.update(mode="rccl", engine=engine)
        except Exception as ex:                      # loud, not silent: the run continues on the rendezvous fallback
            import sys
            import torch.distributed as dist
            print(f"[diffpir_amd.dist] rank {_state['rank']}: RCCL through the C ABI failed ({ex}); falling back to torch.distributed/nccl",
                  file=sys.stderr, flush=True)
            if not dist.is_initialized():
                dist.init_process_group("nccl", rank=_state["rank"], world_size=_state["world"])
            _state.update(mode="torch", engine=engine)
    elif _state["mode"] == "torch":
        _state["engine"] = engine


def collective_name() -> str:
    if _state["mode"] == "torch":
        import torch.distributed as dist
        return f"torch.distributed/{dist.get_backend()}"
    return {"rccl": "RCCL via the C ABI (dpir_allgather_results)", "rccl-pending": "RCCL via the C ABI (not attached)"}.get(_state["mode"], "none")


def barrier():
    if _state["mode"] == "rccl":
        e = _state["engine"]
        e._check(e.lib.dpir_comm_barrier(e.h))
    elif _state["mode"] == "torch":
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of one scalar (the bench's elapsed time)."""
    if _state["mode"] == "rccl":
        import ctypes as C
        e = _state["engine"]
        v = C.c_double(float(value))
        e._check(e.lib.dpir_comm_allreduce_max(e.h, C.byref(v)))
        return float(v.value)
    if _state["mode"] == "torch":
        import torch
        import torch.distributed as dist
        if dist.get_backend() == "gloo":
            device = None
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value


def rank_info(engine, local_rank: int) -> dict:
    """What this rank runs on: device ordinal / name / PCI bus id (dpir_device_info), the librccl version the C ABI bound
    (dpir_comm_version), pid and collective -- gathered by `gather_rank_info` into the bench line so that a multi-GPU run diagnoses itself."""
    import ctypes as C
    buf = C.create_string_buffer(200)
    info = buf.value.decode() if engine.lib.dpir_device_info(engine.h, buf, 200) == 0 else "?"
    v = C.c_int(0)
    engine.lib.dpir_comm_version(C.byref(v))
    return {"rank": _state["rank"], "world": _state["world"], "local_rank": local_rank, "device": info, "rccl_version": int(v.value),
            "pid": os.getpid(), "collective": collective_name()}


def gather_rank_info(engine, local_rank: int) -> list:
    """Every rank's rank_info() on every rank: one all-gather of a fixed 512-byte JSON record per rank over the SAME collective as the
    results (so the record itself proves the collective moved bytes between the listed devices)."""
    import json
    rec = json.dumps(rank_info(engine, local_rank)).encode()[:511]
    world = _state["world"]
    if _state["mode"] == "single" or world == 1:
        return [json.loads(rec)]
    try:
        send = engine.to_device(np.frombuffer(rec.ljust(512, b"\0"), np.uint8).reshape(1, 512).copy())
        allr = all_gather_results(send, world, _state["rank"], world, engine=engine)
        allr = allr if isinstance(allr, np.ndarray) else allr.numpy()
        return [json.loads(bytes(allr[r]).rstrip(b"\0").decode()) for r in range(world)]
    except Exception as ex:          # diagnostics must never cost the run its result line
        return [json.loads(rec), {"gather_error": repr(ex)[:200]}]


def shutdown():
    if _state["mode"] == "rccl":
        barrier()
        e = _state["engine"]
        e.lib.dpir_comm_destroy(e.h)
        e.rccl = False
    elif _state["mode"] == "torch":
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    _state.update(mode="single", engine=None)


_SENTINEL = b"\xff" * 128          # rank 0 could not create a unique id: every rank falls back together


def init_rccl(engine, rank: int, world: int, port: int = None):
    """Rank 0 creates the 128-byte ncclUniqueId and ships it to the other ranks over a TCP socket on MASTER_ADDR (stdlib;
    rendezvous plumbing only), then every rank joins the communicator on its engine's device (dpir_comm_init).

    The fallback decision is COLLECTIVE: rank 0 always opens the listener and answers every rank with either the id or a failure
    sentinel, so a missing librccl on rank 0 sends all ranks to torch.distributed together instead of leaving the others spinning
    on a dead port.  Both sides are bounded by DIFFPIR_RENDEZVOUS_TIMEOUT seconds (default 300: ranks reach this point after loading
    the same checkpoint, but the 2 GB topologies take minutes to pack).  A peer introduces itself (b"DPIR" + rank) before it is
    answered; anything else is dropped."""
    import ctypes as C
    import socket
    import struct
    import time
    buf = C.create_string_buffer(128)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = port or int(os.environ.get("MASTER_PORT", "29533")) + 17
    deadline = time.monotonic() + float(os.environ.get("DIFFPIR_RENDEZVOUS_TIMEOUT", "300"))
    if rank == 0:
        failure = None
        try:
            engine._check(engine.lib.dpir_comm_unique_id(buf))
        except Exception as ex:
            failure = ex
        if world > 1:
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            served = set()
            try:
                while len(served) < world - 1:
                    srv.settimeout(max(0.1, deadline - time.monotonic()))
                    try:
                        c, _a = srv.accept()
                    except socket.timeout:
                        raise RuntimeError(f"RCCL rendezvous: only ranks {sorted(served)} of {world - 1} peers connected before the timeout")
                    c.settimeout(10)
                    try:
                        hello = c.recv(8)
                        if len(hello) == 8 and hello[:4] == b"DPIR":
                            peer = struct.unpack("<i", hello[4:])[0]
                            if 0 < peer < world and peer not in served:
                                c.sendall(_SENTINEL if failure is not None else buf.raw)
                                served.add(peer)
                    except OSError:
                        pass
                    finally:
                        c.close()
            finally:
                srv.close()
        if failure is not None:
            raise failure
    else:
        c = None
        while time.monotonic() < deadline:     # rank 0 may still be packing weights
            try:
                c = socket.create_connection((addr, port), timeout=5)
                break
            except OSError:
                time.sleep(0.1)
        if c is None:
            raise RuntimeError("RCCL rendezvous: rank 0 is not listening")
        c.settimeout(max(1.0, deadline - time.monotonic()))
        c.sendall(b"DPIR" + struct.pack("<i", rank))
        data = b""
        while len(data) < 128:
            chunk = c.recv(128 - len(data))
            if not chunk:
                raise RuntimeError("RCCL rendezvous: short read of the unique id")
            data += chunk
        c.close()
        if data == _SENTINEL:
            raise RuntimeError("RCCL rendezvous: rank 0 could not create a unique id (librccl missing?)")
        buf.raw = data
    engine._check(engine.lib.dpir_comm_init(engine.h, world, rank, buf))
    engine.rccl = True


def _gather_rccl(engine, local, n_images: int, rank: int, world: int):
    """ncclAllGather through the C ABI on the engine stream.  `local`: DeviceArray [n_local, ...] (any dtype, moved as bytes) or a
    host numpy array (small per-image metrics: staged through engine-owned device words).  Returns the same kind, [n_images, ...]."""
    from .engine import DeviceArray
    host = not isinstance(local, DeviceArray)
    if host:
        a = np.ascontiguousarray(local)
        dev = engine.empty(a.shape, a.dtype)
        if a.size:
            dev.copy_from(a)
        local = dev
    sizes = [shard_range(n_images, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    item = tuple(local.shape[1:])
    row_bytes = int(np.prod(item, dtype=np.int64)) * local.dtype.itemsize
    send = local
    if local.shape[0] < mx:                    # ragged shards: pad to the largest shard for the collective
        send = engine.empty((mx,) + item, local.dtype)
        if local.shape[0]:
            engine._check(engine.lib.dpir_d2d(engine.h, send.ptr, local.ptr, local.shape[0] * row_bytes))
    recv = engine.empty((world * mx,) + item, local.dtype)
    engine._check(engine.lib.dpir_allgather_results(engine.h, send.ptr, recv.ptr, mx * row_bytes))
    if all(hi - lo == mx for lo, hi in sizes):
        out = recv                               # recv is already [n_images, ...]
    else:
        out = engine.empty((n_images,) + item, local.dtype)
        for r, (lo, hi) in enumerate(sizes):
            if hi > lo:
                engine._check(engine.lib.dpir_d2d(engine.h, out.ptr + lo * row_bytes, recv.ptr + r * mx * row_bytes, (hi - lo) * row_bytes))
    engine.sync()
    return out.numpy() if host else out


def all_gather_results(local, n_images: int, rank: int, world: int, engine=None):
    """local: [n_local, ...] results of this rank (uint8 NHWC images, or per-image metric rows).  Returns [n_images, ...] on
    every rank.  Shards may differ by one image; each is padded to the largest shard for the collective.

    Engine with an RCCL communicator (attach / init_rccl): DeviceArray in -> DeviceArray out (engine-owned buffers,
    ncclAllGather on the engine stream), host numpy in -> numpy out.  Otherwise torch.distributed (torch tensors; DeviceArrays
    and numpy arrays are staged through the host -- the several-ranks-on-one-GPU / CPU test configuration)."""
    from .engine import DeviceArray
    if engine is not None and getattr(engine, "rccl", False):
        if hasattr(local, "data_ptr"):           # a torch tensor from an older caller: stage through the host
            return _gather_rccl(engine, local.cpu().numpy(), n_images, rank, world)
        return _gather_rccl(engine, local, n_images, rank, world)
    import torch
    import torch.distributed as dist
    kind = "torch"
    if isinstance(local, DeviceArray):
        kind, eng0 = "device", local.engine
        local = torch.from_numpy(local.numpy() if local.shape[0] else np.zeros(local.shape, local.dtype))
    elif isinstance(local, np.ndarray):
        kind, local = "numpy", torch.from_numpy(np.ascontiguousarray(local))

    def back(t):
        if kind == "device":
            return eng0.to_device(t.cpu().numpy())
        return t.cpu().numpy() if kind == "numpy" else t
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return back(local)
    sizes = [shard_range(n_images, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], 0)
    pad = pad.contiguous()
    if dist.get_backend() == "gloo" and pad.is_cuda:       # gloo stages through the host (tests on one GPU)
        pad = pad.cpu()
    elif dist.get_backend() == "nccl" and not pad.is_cuda:  # a process group created with "nccl" has no CPU backend
        pad = pad.cuda()
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)
    return back(out.to(local.device))


def check_dps_sharding(engine, cfg, n_images: int, world: int):
    """generate_mode DPS_y0 is the ONE mode that couples the images of a batch: grad_and_value takes the norm over the whole batch
    (utils/utils_model.py:392) and the update `x = xt - norm_grad` (main_ddpir.py:437) has no `* norm` factor that would cancel it.
    Sharded, every rank's loop therefore all-reduces the squared residual sums over the communicator (csrc/api.hip `dps_norm`) -- which
    needs the C ABI's RCCL communicator on the engine and every rank inside the loop (no empty shard).

    DPS_yt multiplies the norm back in (main_ddpir.py:444), so it is correct without any exchange -- but with a communicator attached
    its loop all-reduces the same sums (the whole batch's norm, as the reference has it: results do not depend on the sharding even in
    the last bit), so an empty shard would leave the other ranks waiting in that collective: refused as well."""
    mode = getattr(cfg, "generate_mode", "")
    if world <= 1 or mode not in ("DPS_y0", "DPS_yt"):
        return
    rccl = getattr(engine, "rccl", False)
    if mode == "DPS_y0" and not rccl:
        raise NotImplementedError("generate_mode DPS_y0 on several GPUs needs the default 'rccl' collective: the batch-wide residual norm is all-reduced "
                                  "inside the loop through the engine's communicator (torch.distributed fallbacks cannot reach into the loop)")
    if rccl and n_images < world:
        raise NotImplementedError(f"generate_mode {mode}: a batch of {n_images} images cannot be sharded over {world} ranks (every rank must run the loop: "
                                  "its all-reduce of the batch norm is collective)")


def restore_sharded(engine, cfg, y, k=None, mask=None, labels=None, *, rank: int, world: int, image_offset: int = 0, seed: int = 0,
                    use_graph: bool = True, noise_source: str = "device", host_noise=None, cache: dict = None,
                    skip_dead_final_eval: bool = False):
    """One GLOBAL batch restored by `world` ranks: rank r runs images [lo, hi) of the batch on its own GPU (its own engine,
    its own replayed graph, no collective inside the loop) and ONE all-gather of the uint8 results ends the batch
    (SURVEY.md 8e).  y / k / mask / labels are the global batch (numpy, host); every rank slices its block.

    Device noise is keyed by the global image index (image_offset + lo + b), so the gathered result is independent of
    `world`.  Host noise (parity mode) must be pre-drawn for the GLOBAL batch: host_noise = (init, n1, n2[, nrp]) as returned
    by restore.draw_host_noise for the global shape; each rank uploads its image slice.
    Returns (uint8 [n_images, H, W, 3] on every rank as an engine-owned DeviceArray, local fp32 DeviceArray)."""
    from . import restore
    n = y.shape[0]
    check_dps_sharding(engine, cfg, n, world)
    lo, hi = shard_range(n, rank, world)
    sl = slice(lo, hi)
    H, W = y.shape[2] * cfg.sf, y.shape[3] * cfg.sf
    out_u8 = engine.empty((hi - lo, H, W, 3), np.uint8)
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
