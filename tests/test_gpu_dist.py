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
    make_model(eng, uo.tiny_hp())
    cfg = restore.LoopConfig(task="deblur", iter_num=5, lambda_=7.0, zeta=0.3, eta=0.5 if noise == "host" else 0.0)
    case = _case(n)
    drawn = None
    if noise == "host":                  # every rank draws the GLOBAL noise in the reference's order and keeps its image slice
        _, steps, _ = restore._steps(cfg)
        drawn = restore.draw_host_noise(seeded_noise_fn_np(77), steps, (n, 3, 32, 32), True)
    u8, _ = ddist.restore_sharded(eng, cfg, case["y"], k=case["k"], rank=r, world=w, image_offset=100, seed=9, use_graph=True,
                                  noise_source=noise, host_noise=drawn)
    ret[rank] = u8.cpu().numpy().tobytes()
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


def test_rccl_allgather_through_the_c_abi_world_1():
    """dpir_comm_* / dpir_allgather_results bind ncclAllGather from librccl.so (dlopen).  One GPU admits one rank per
    communicator, so this checks the binding, the stream ordering and the padding / slicing logic at world = 1."""
    import torch
    import diffpir_amd
    from diffpir_amd import dist as ddist
    eng = diffpir_amd.Engine(0)
    try:
        ddist.init_rccl(eng, 0, 1)
        local = torch.randint(0, 256, (3, 8, 8, 3), dtype=torch.uint8, device="cuda:0")
        out = ddist.all_gather_results(local, 3, 0, 1, engine=eng)
        assert out.shape == local.shape and torch.equal(out, local) and out.data_ptr() != local.data_ptr()
    finally:
        eng.close()


def test_bench_launch_line_with_the_nccl_backend_world_1():
    """The driver's multi-GPU launch line (`python -m torch.distributed.run ... bench.py --gpus N`) on the one GPU of this box:
    DIFFPIR_FORCE_DIST=1 makes rank 0 join a 1-rank RCCL group, so process-group init, the barrier, the uint8 all-gather and the
    MAX all-reduce of the timing run through backend "nccl" exactly as they do at N > 1."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DIFFPIR_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--nfe", "6", "--batch", "4", "--no-cpu-baseline", "--no-c3", "--no-alt"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["global_batch"] == 4


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
