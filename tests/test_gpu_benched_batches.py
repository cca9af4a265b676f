"""-m gpu: parity at the batches bench.py actually runs for BASELINE configs[2] (ImageNet-256 topology, B = 32) and configs[3]'s per-GPU
share (FFHQ topology, motion PSF, B = 32).  The conv7 / conv6 dispatch (`blocks < 384`, csrc/conv6.hip), the split-K factors and
the fused low-resolution prologue's workgroup count all depend on B, so the kernels the bench times are only exercised at these
batches (round-3 review: only configs[1]'s B = 16 had such tests).  Oracle on a subset of the images + invariance against the same
engine at a small batch; everything through the C ABI."""
import numpy as np
import pytest
import torch

import diffpir_amd
from diffpir_amd import restore, synth
from oracle import unet_oracle as uo, diffpir_oracle as do
from tests import gpu_common
from tests.gpu_common import make_model, seeded_noise_fn_np, rel_err, fft_prox_parity

pytestmark = pytest.mark.gpu
TOL_LAYER = 2e-5


def _forward_at_batch(e, sd, hp, B, seed, probe, pair, key, precision):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, 3, 256, 256), generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    out = e.unet_forward(e.to_device(x.numpy()), t.numpy()).numpy()
    worst = 0.0
    for i in probe:
        k = f"{key}_{i}"
        if k not in gpu_common._ORACLE_CACHE:
            gpu_common._ORACLE_CACHE[k] = uo.unet_forward(sd, hp, x[i:i + 1], t[i:i + 1]).numpy()
        worst = max(worst, rel_err(out[i:i + 1], gpu_common._ORACLE_CACHE[k]))
    lo, hi = pair
    out2 = e.unet_forward(e.to_device(x[lo:hi].numpy()), t[lo:hi].numpy()).numpy()
    inv = rel_err(out[lo:hi], out2)
    print(f"{key} 256^2 forward B={B} [{precision}]: worst rel err vs oracle (images {probe}) {worst:.3e}; B={B} vs B={hi - lo} on images "
          f"{lo}-{hi - 1} {inv:.3e}")
    assert worst < TOL_LAYER and inv < TOL_LAYER


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_c3_imagenet256_forward_at_the_benched_batch_b32(precision):
    """configs[2] as benched (`config_c3`): ImageNet-256 topology at 256^2, B = 32, a different timestep per image."""
    e = diffpir_amd.Engine(0)
    try:
        e.set_precision(precision)
        hp = uo.imagenet256_hp()
        model, sd = make_model(e, hp)
        _forward_at_batch(e, sd, hp, 32, 91, (5, 31), (20, 22), "c3_b32", precision)
    finally:
        e.close()


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def ffhq(request):
    e = diffpir_amd.Engine(0)
    e.set_precision(request.param)
    model, sd = make_model(e, uo.ffhq_hp())
    yield e, sd, request.param
    e.close()


def test_c4_ffhq_forward_at_the_per_gpu_batch_b32(ffhq):
    """configs[3]'s per-GPU share (32 of 256 images): FFHQ topology forward at B = 32."""
    e, sd, precision = ffhq
    _forward_at_batch(e, sd, uo.ffhq_hp(), 32, 92, (0, 17, 31), (30, 32), "c4_b32", precision)


def test_c4_motion_loop_at_the_per_gpu_batch_b32_4nfe(ffhq):
    """configs[3] per GPU: 32 images, one non-symmetric 61x61 motion PSF per image, 4 NFE through the replayed graph, host noise
    drawn for the whole batch in the reference's order; the oracle restores images 7 and 30 with the same per-image noise slices."""
    e, sd, precision = ffhq
    B, sub = 32, [7, 30]
    case = synth.make_case("deblur", B, 256, 256, seed=17, ksize=61, blur="motion")
    cfg = restore.LoopConfig(task="deblur", iter_num=4, lambda_=7.0, zeta=0.3)
    out = restore.restore_batch(e, cfg, case["y"], k=case["k"], noise_source="host", noise_fn=seeded_noise_fn_np(67),
                                use_graph=True).numpy()

    def sliced(seed):
        g = torch.Generator().manual_seed(seed)
        return lambda like: torch.randn((B,) + tuple(like.shape[1:]), generator=g, dtype=torch.float32)[sub]
    key = "c4_b32_4nfe"
    if key not in gpu_common._ORACLE_CACHE:
        ocfg = do.LoopConfig("deblur", 4, 12.75 / 255, 7.0, 0.3)
        ty, tk = torch.from_numpy(case["y"][sub]), torch.from_numpy(case["k"][sub])
        ref = do.restore(sd, uo.ffhq_hp(), ocfg, ty, k=tk, noise_fn=sliced(67)).numpy()
        exact = do.restore(sd, uo.ffhq_hp(), ocfg, ty, k=tk, noise_fn=sliced(67), exact_prox=True).numpy()
        gpu_common._ORACLE_CACHE[key] = (ref, exact)
    ref, exact = gpu_common._ORACLE_CACHE[key]
    fft_prox_parity(out[sub], ref, case["gt"][sub], f"C4 motion B=32 4-NFE, images 7 and 30 [{precision}] vs oracle", exact=exact)


_FUSE_SNIPPET = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
import diffpir_amd
from oracle import unet_oracle as uo
from tests.gpu_common import make_model
e = diffpir_amd.Engine(0); e.set_precision({precision!r})
make_model(e, uo.ffhq_hp())
g = torch.Generator().manual_seed(77)
x = torch.randn((16, 3, 256, 256), generator=g); t = torch.randint(0, 1000, (16,), generator=g)
xd = e.to_device(x.numpy())
a = e.unet_forward(xd, t.numpy()).numpy()
b = e.unet_forward(xd, t.numpy()).numpy()
assert np.array_equal(a, b), "two forwards of the same input differ"
np.save({out!r}, a)
"""


@pytest.mark.parametrize("precision", ["f16x3", "f16x1"])
def test_fused_h1_hop_is_reproducible_and_equals_the_unfused_path(tmp_path, precision):
    """conv7's fused hop conv1 -> GroupNorm + FiLM + SiLU -> conv2 (Conv6Emit: per-image integer accumulators, arrival counter, planes
    written by conv1's epilogue; on for every ResBlock at >= 64^2 when B = 16) against the same forward with DPIR_FUSE_H1=0 (fp32 h1,
    gn_prm, act_split): two forwards in one process are bit-identical (integer accumulation is order-independent), and the two paths
    agree to the level of the GroupNorm statistics' summation order."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for fuse in ("1", "0"):
        out = str(tmp_path / f"fwd_{fuse}.npy")
        env = dict(os.environ, DPIR_FUSE_H1=fuse)
        r = subprocess.run([sys.executable, "-c", _FUSE_SNIPPET.format(root=root, precision=precision, out=out)], cwd=root, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        outs[fuse] = np.load(out)
    err = rel_err(outs["1"], outs["0"])
    print(f"fused h1 hop vs unfused, FFHQ 256^2 B=16 [{precision}]: rel err {err:.3e}")
    assert err < (5e-6 if precision == "f16x3" else 2e-3)


def test_fused_hop_is_offered_only_where_groups_stay_inside_a_wave():
    """conv7's emission folds GroupNorm group sums inside one wave's 64 channels: channels per group (Cout / 32) must divide 64.  Widths from
    channel_mult x3 / x5 / x6 / x7 (384, 640, 768, 896) pass the 128-channel-block test but have groups that straddle waves: the dispatch
    must fall back to the unfused path for them (round-4 advisor finding: silently wrong statistics).  The waiting-set limit comes from the
    device (CUs x occupancy), half of it at most."""
    import ctypes as C
    from diffpir_amd import _lib
    e = diffpir_amd.Engine(0)
    dbg = _lib.load_debug()
    cap = C.c_int(0)
    assert dbg.dpir_debug_conv7_emit_supported(e.h, 16, 256, 128, 128, C.byref(cap)) == 1
    assert cap.value >= 2 and cap.value % 2 == 0
    for cout in (128, 256, 512, 1024):          # 128 x 128: 64 tiles per image, >= 384 workgroups at B = 16 for every width
        assert dbg.dpir_debug_conv7_emit_supported(e.h, 16, cout, 128, 128, None) == 1, cout
    for cout in (384, 640, 768, 896):
        assert dbg.dpir_debug_conv7_emit_supported(e.h, 16, cout, 128, 128, None) == 0, cout
    # a waiting set larger than half the resident workgroups is refused: 512 x 512 has 1024 tiles per image
    assert dbg.dpir_debug_conv7_emit_supported(e.h, 16, 128, 512, 512, None) == (1 if 1024 <= cap.value // 2 else 0)
    e.close()


_CONV8_SNIPPET = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
import diffpir_amd
from oracle import unet_oracle as uo
from tests.gpu_common import make_model
e = diffpir_amd.Engine(0); e.set_precision({precision!r})
make_model(e, uo.ffhq_hp())
g = torch.Generator().manual_seed(78)
x = torch.randn((3, 3, 256, 256), generator=g); t = torch.randint(0, 1000, (3,), generator=g)
np.save({out!r}, e.unet_forward(e.to_device(x.numpy()), t.numpy()).numpy())
"""


@pytest.mark.parametrize("precision", ["f16x3", "f16x1"])
def test_fused_output_layer_equals_the_planes_route(tmp_path, precision):
    """conv8 (round 5): the output layer GroupNorm -> SiLU -> 3x3 conv 128 -> 6 with the normalisation, activation and f16 split done in
    the convolution's own LDS fill (v_mfma_f32_16x16x32_f16) against the same forward with DPIR_CONV8=0 (act_split planes + conv7's
    narrow variant, v_mfma_f32_32x32x16_f16).  Same operands, same three products per accumulator; only the K blocking inside an MFMA
    differs, so the two agree to fp32 summation-order level.  B = 3: tiles of three images, image borders included."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for on in ("1", "0"):
        out = str(tmp_path / f"fwd_{on}.npy")
        r = subprocess.run([sys.executable, "-c", _CONV8_SNIPPET.format(root=root, precision=precision, out=out)], cwd=root,
                           env=dict(os.environ, DPIR_CONV8=on), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        outs[on] = np.load(out)
    err = rel_err(outs["1"], outs["0"])
    print(f"fused output layer vs planes route, FFHQ 256^2 B=3 [{precision}]: rel err {err:.3e}")
    assert err < 2e-6


_TIMEOUT_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import diffpir_amd
from diffpir_amd import restore, synth
from oracle import unet_oracle as uo
from tests.gpu_common import make_model
e = diffpir_amd.Engine(0); e.set_precision("f16x3")
make_model(e, uo.ffhq_hp())
x = e.to_device(np.random.default_rng(0).standard_normal((16, 3, 256, 256)).astype(np.float32))
a = e.unet_forward(x, np.full(16, 500)).numpy()          # the D2H copy synchronises: time-out seen, hop latched off, forward re-issued
b = e.unet_forward(x, np.full(16, 500)).numpy()          # unfused from the start
print("FORWARD EQUAL:", bool(np.array_equal(a, b)))
case = synth.make_case("deblur", 16, 256, 256, seed=5, ksize=25)
cfg = restore.LoopConfig(task="deblur", iter_num=3, lambda_=7.0, zeta=0.3)
o1 = restore.restore_batch(e, cfg, case["y"], k=case["k"], noise_source="device", seed=3).numpy()
print("LOOP FINITE:", bool(np.isfinite(o1).all()))
np.save({out!r}, a[:4])
np.save({out!r} + ".loop.npy", o1)
"""

_TIMEOUT_LOOP_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import diffpir_amd
from diffpir_amd import restore, synth
from oracle import unet_oracle as uo
from tests.gpu_common import make_model
e = diffpir_amd.Engine(0); e.set_precision("f16x3")
make_model(e, uo.ffhq_hp())
case = synth.make_case("deblur", 16, 256, 256, seed=5, ksize=25)
cfg = restore.LoopConfig(task="deblur", iter_num=3, lambda_=7.0, zeta=0.3)
o1 = restore.restore_batch(e, cfg, case["y"], k=case["k"], noise_source="device", seed=3).numpy()     # first thing this engine does
np.save({out!r}, o1)
"""


def test_fused_h1_hop_time_out_degrades_to_the_unfused_path(tmp_path):
    """The fused hop's workgroups wait for one another.  Should that wait ever not end (a GPU shared by three or more engines: DESIGN 3.1),
    a workgroup gives up -- and the engine must neither hang, nor return a wrong image, nor die: it latches the hop off for its lifetime
    (one line on stderr) and re-runs the invalidated work on the unfused path.  Forced here with the test hooks (an arrival count that cannot
    be reached, a short spin limit) for an eager forward and for the restoration loop as the engine's first action; both must equal a
    DPIR_FUSE_H1=0 run bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env_extra in (("timeout", dict(DPIR_FUSE_EXPECT_EXTRA="1", DPIR_FUSE_SPIN_LIMIT="300")), ("unfused", dict(DPIR_FUSE_H1="0"))):
        for kind, snippet in (("fwd", _TIMEOUT_SNIPPET), ("loop", _TIMEOUT_LOOP_SNIPPET)):
            out = str(tmp_path / f"{tag}_{kind}.npy")
            r = subprocess.run([sys.executable, "-c", snippet.format(root=root, out=out)], cwd=root, env=dict(os.environ, **env_extra),
                               capture_output=True, text=True, timeout=600)
            txt = r.stdout + r.stderr
            assert r.returncode == 0, txt[-2000:]
            if tag == "timeout":
                assert txt.count("fused GroupNorm hop timed out") == 1, txt[-2000:]          # said once, then the hop is off
            if kind == "fwd":
                assert "FORWARD EQUAL: True" in txt and "LOOP FINITE: True" in txt, txt[-2000:]
                res[tag, "fwd_loop"] = np.load(out + ".loop.npy")
            res[tag, kind] = np.load(out)
    assert np.array_equal(res["timeout", "fwd"], res["unfused", "fwd"])
    assert np.array_equal(res["timeout", "fwd_loop"], res["unfused", "fwd_loop"])
    assert np.array_equal(res["timeout", "loop"], res["unfused", "loop"])          # the loop re-ran itself after the time-out
