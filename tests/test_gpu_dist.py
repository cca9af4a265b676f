"""-m gpu: the multi-GPU product path on ONE GPU.  Two processes (one engine each, both on device 0) restore a B=4 batch
through diffpir_amd.dist.restore_sharded -- the function the YAML driver and bench.py call -- with the result all-gather on the
`gloo` backend; the gathered uint8 batch must equal the single-process result bit for bit (device noise is keyed by the global
image index, so sharding cannot change an image).  Covers even and ragged splits and the host-noise (parity) slicing."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(n):
    from diffpir_amd import synth
    return synth.make_case("deblur", n, 32, 32, seed=31, ksize=9)


def _run(rank, world, port, n, noise, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch
    import diffpir_amd
    from diffpir_amd import dist as ddist, restore
    from oracle import unet_oracle as uo
    from tests.gpu_common import make_model, seeded_noise_fn_np
    r, _, w = ddist.init("gloo")
    eng = diffpir_amd.Engine(0)
    ddist.attach(eng)
    make_model(eng, uo.tiny_hp())
    cfg = restore.LoopConfig(task="deblur", iter_num=5, lambda_=7.0, zeta=0.3, eta=0.5 if noise == "host" else 0.0)
    case = _case(n)
    drawn = None
    if noise == "host":                  # every rank draws the GLOBAL noise in the reference's order and keeps its image slice
        _, steps, _ = restore._steps(cfg)
        drawn = restore.draw_host_noise(seeded_noise_fn_np(77), steps, (n, 3, 32, 32), True)
    u8, _ = ddist.restore_sharded(eng, cfg, case["y"], k=case["k"], rank=r, world=w, image_offset=100, seed=9, use_graph=True,
                                  noise_source=noise, host_noise=drawn)
    ret[rank] = u8.numpy().tobytes()                # engine-owned DeviceArray
    ddist.shutdown()
    eng.close()


@pytest.mark.parametrize("n,noise", [(4, "device"), (3, "device"), (4, "host")])
def test_two_ranks_on_one_gpu_equal_one_rank(n, noise):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    one, two = mgr.dict(), mgr.dict()
    mp.spawn(_run, args=(1, _free_port(), n, noise, one), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), n, noise, two), nprocs=2, join=True)
    assert len(one[0]) == n * 32 * 32 * 3
    assert two[0] == one[0] and two[1] == one[0]


def test_rccl_collectives_through_the_c_abi_world_1():
    """dpir_comm_* / dpir_allgather_results / dpir_comm_barrier / dpir_comm_allreduce_max bind RCCL from librccl.so (dlopen).  One
    GPU admits one rank per communicator, so this checks the binding, the stream ordering, the engine-owned buffers and the
    padding / slicing logic at world = 1 -- through dist.init / attach, the calls the multi-GPU launch makes (the DEFAULT
    collective: no torch.distributed process group exists in this process)."""
    import diffpir_amd
    from diffpir_amd import dist as ddist
    from diffpir_amd.engine import DeviceArray
    os.environ.update(DIFFPIR_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.pop("DIFFPIR_COLLECTIVE", None)
    eng = diffpir_amd.Engine(0)
    try:
        r, _, w = ddist.init()
        assert (r, w) == (0, 1)
        ddist.attach(eng)
        assert "C ABI" in ddist.collective_name()
        import torch.distributed as tdist
        assert not tdist.is_initialized()
        host = np.random.default_rng(0).integers(0, 256, (3, 8, 8, 3), dtype=np.uint8)
        local = eng.to_device(host)
        out = ddist.all_gather_results(local, 3, 0, 1, engine=eng)
        assert isinstance(out, DeviceArray) and out.ptr != local.ptr and np.array_equal(out.numpy(), host)
        rows = np.arange(6, dtype=np.float64).reshape(3, 2)               # per-image metric rows: host numpy in, numpy out
        assert np.array_equal(ddist.all_gather_results(rows, 3, 0, 1, engine=eng), rows)
        ddist.barrier()
        assert ddist.max_over_ranks(3.25) == 3.25
        ddist.shutdown()
    finally:
        os.environ.pop("DIFFPIR_FORCE_DIST", None)
        eng.close()


@pytest.mark.parametrize("collective", ["rccl", "nccl"])
def test_yaml_driver_with_a_process_group_world_1(tmp_path, collective):
    """`python -m diffpir_amd.main_ddpir` under DIFFPIR_FORCE_DIST=1: the driver's multi-GPU code path (sharding, the uint8
    all-gather, the per-image PSNR gather -- a HOST float64 array, which a torch "nccl" group cannot move without staging it on
    the device) at world = 1, on the default C-ABI collective and on torch.distributed's nccl backend."""
    import subprocess
    import sys
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "engine_example.yaml")))
    cfg.update(iter_num=3, batch_size=2, task="deblur")
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, DIFFPIR_FORCE_DIST="1", DIFFPIR_COLLECTIVE=collective, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "diffpir_amd.main_ddpir", "--opt", str(p), "--synthetic", "3", "--max-sweeps", "1"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "Average PSNR(RGB)" in r.stdout + r.stderr


def test_bench_launch_line_on_the_default_collective_world_1():
    """The driver's multi-GPU launch line (`python -m torch.distributed.run ... bench.py --gpus N`) on the one GPU of this box:
    DIFFPIR_FORCE_DIST=1 makes rank 0 create a 1-rank RCCL communicator through the C ABI, so the rendezvous, the barrier, the
    uint8 all-gather and the MAX all-reduce of the timing run exactly as they do at N > 1 (DIFFPIR_COLLECTIVE unset)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DIFFPIR_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DIFFPIR_COLLECTIVE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--nfe", "6", "--batch", "4", "--no-cpu-baseline", "--no-c3", "--no-alt"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["global_batch"] == 4
    assert "C ABI" in line["config"]["collective"]


def test_bench_launch_line_two_ranks_on_one_gpu_gloo():
    """The N = 2 launch line with both ranks on the one GPU of this box (gloo: RCCL refuses two ranks per GPU): every rank > 0 code
    path of bench.py -- sharded image offsets, the padded all-gather, MAX over ranks, rank 0 printing the one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DIFFPIR_BENCH_BACKEND="gloo", DIFFPIR_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--nfe", "6", "--batch", "2"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["scaling"] == "weak" and line["value"] > 0
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None


# ---- N > 1 ranks on N devices through the DEFAULT collective (RCCL over xGMI, bound by the C ABI).  These run the moment a box shows
# ---- two GPUs and skip -- naming the device count -- on the one-GPU boxes this project has had so far (main_ddpir.py:135).

def _need_two_devices():
    from diffpir_amd.engine import device_count
    n = device_count()
    if n < 2:
        pytest.skip(f"RCCL with two ranks needs two GPUs: dpir_device_count() = hipGetDeviceCount() = {n} on this box "
                    f"(ncclCommInitRank(world=2) / the TCP unique-id exchange of dist.init_rccl stay unexecuted here)")
    return n


def _run_rccl(rank, world, port, n, ret):
    """One rank of the product launch: device = LOCAL_RANK, dist.init('rccl') -> attach (unique id over TCP, ncclCommInitRank) ->
    restore_sharded (ncclAllGather of engine-owned uint8 buffers on the engine stream)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("DIFFPIR_COLLECTIVE", None)
    import diffpir_amd
    from diffpir_amd import dist as ddist, restore
    from oracle import unet_oracle as uo
    from tests.gpu_common import make_model
    r, lr, w = ddist.init("rccl")
    eng = diffpir_amd.Engine(lr)
    ddist.attach(eng)
    if w > 1:
        assert "C ABI" in ddist.collective_name(), ddist.collective_name()       # no silent fallback to torch.distributed
    make_model(eng, uo.tiny_hp())
    cfg = restore.LoopConfig(task="deblur", iter_num=5, lambda_=7.0, zeta=0.3)
    case = _case(n)
    u8, _ = ddist.restore_sharded(eng, cfg, case["y"], k=case["k"], rank=r, world=w, image_offset=100, seed=9, use_graph=True,
                                  noise_source="device")
    ddist.barrier()
    assert ddist.max_over_ranks(float(r)) == float(w - 1)
    ret[rank] = u8.numpy().tobytes()
    ddist.shutdown()
    eng.close()


@pytest.mark.parametrize("n", [4, 3])
def test_rccl_two_ranks_two_devices(n):
    """Gathered uint8 batch of two ranks on devices 0 / 1 == the one-rank result, bit for bit (even and ragged shards)."""
    _need_two_devices()
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    one, two = mgr.dict(), mgr.dict()
    mp.spawn(_run_rccl, args=(1, _free_port(), n, one), nprocs=1, join=True)
    mp.spawn(_run_rccl, args=(2, _free_port(), n, two), nprocs=2, join=True)
    assert len(one[0]) == n * 32 * 32 * 3
    assert two[0] == one[0] and two[1] == one[0]


def test_bench_launch_line_two_ranks_two_devices_c4():
    """The N = 2 launch line on the default collective with `--config c4`: BASELINE configs[3]'s per-GPU work (FFHQ topology, motion PSF,
    32 images per GPU), shortened to 6 NFE.  (Without --config the bench keeps configs[1]'s 16 images per GPU at every N.)"""
    _need_two_devices()
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("DIFFPIR_COLLECTIVE", "DIFFPIR_BENCH_BACKEND", "DIFFPIR_BENCH_DEVICE"):
        env.pop(v, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--config", "c4", "--steps", "1", "--warmup", "1",
           "--nfe", "6"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64 and line["value"] > 0
    assert "C ABI" in line["config"]["collective"]
