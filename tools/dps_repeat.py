"""Root-cause probe for the one-off DPS_y0 excursion (round-5 review item 2): the engine side of
tests/test_gpu_dps.py::test_dps_y0_loop_full_size_ffhq_vs_oracle run N times in ONE process, every output compared BITWISE with the
first; between runs optional churn (a second engine created / destroyed, a plain DiffPIR loop on the same engine, a burst of eager
forwards) to imitate what the full suite does before the test.  Also two back-to-back VJPs per iteration compared bitwise tap by tap.
usage: python tools/dps_repeat.py [N] [churn: 0|1]      (GPU box only)"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import restore, synth, script_util, weights

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
churn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hp = weights.model_hp("ffhq")


def engine(grad=True):
    e = diffpir_amd.Engine(0); e.set_precision(os.environ.get("DIFFPIR_PRECISION", "f16x3"))
    if grad:
        e.enable_grad()
    m = script_util.create_model(**weights.create_model_kwargs(hp), engine=e); m.load_state_dict(weights.synth_state_dict(hp, 0))
    return e


def noise_fn(seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    return lambda shape: torch.randn(tuple(shape), generator=g, dtype=torch.float32).numpy()


poison = os.environ.get("DPS_POISON")
if poison:
    # recycled device memory: fill a few GB with a pattern, free it, THEN create the engine -- its lazily allocated workspace buffers now
    # start from that pattern instead of the zero pages of a fresh process (what the full suite's earlier tests leave behind)
    e0 = diffpir_amd.Engine(0)
    val = {"nan": np.nan, "big": 3.0e4, "one": 1.0}[poison]
    blk = np.full((256, 1024, 1024), val, np.float32)          # 1 GiB
    arrs = [e0.to_device(blk) for _ in range(int(os.environ.get("DPS_POISON_GB", "24")))]
    e0.sync(); del arrs; e0.close(); del blk
e = engine()
case = synth.make_case("sr", 2, 256, 256, seed=31, sf=4)
cfg = restore.LoopConfig(task="sr", iter_num=5, lambda_=6.0, zeta=0.25, sf=4, sr_mode="cubic", generate_mode="DPS_y0")
cfg_plain = restore.LoopConfig(task="sr", iter_num=3, lambda_=6.0, zeta=0.25, sf=4, sr_mode="cubic")
rng = np.random.default_rng(5)
xv = rng.standard_normal((2, 3, 256, 256)).astype(np.float32)
gv = rng.standard_normal((2, 6, 256, 256)).astype(np.float32)
tv = np.array([400, 400], dtype=np.int64)
first = None; first_dx = None
bad = 0
t0 = time.time()
for it in range(N):
    out = restore.restore_batch(e, cfg, case["y"], noise_source="host", noise_fn=noise_fn(81)).numpy()
    if first is None:
        first = out.copy()
    d = np.abs(out - first)
    if d.max() != 0.0:
        bad += 1
        idx = np.argwhere(d > 0)
        print(f"iter {it}: LOOP DEVIATES max {d.max():.3e}, {len(idx)} elements differ, first at {idx[0].tolist()}, "
              f"bbox rows {idx[:,2].min()}..{idx[:,2].max()} cols {idx[:,3].min()}..{idx[:,3].max()} images {sorted(set(idx[:,0].tolist()))}", flush=True)
    # two back-to-back VJPs, bitwise
    o1, dx1 = e.unet_vjp(e.to_device(xv), tv, e.to_device(gv)); a = dx1.numpy().copy()
    o2, dx2 = e.unet_vjp(e.to_device(xv), tv, e.to_device(gv)); b = dx2.numpy()
    if first_dx is None:
        first_dx = a.copy()
    for nm, v in (("vjp#1-vs-#2", np.abs(a - b).max()), ("vjp#1-vs-first", np.abs(a - first_dx).max())):
        if v != 0.0:
            bad += 1
            print(f"iter {it}: {nm} deviates max {v:.3e} (|dx| max {np.abs(a).max():.3e})", flush=True)
    if churn:
        if it % 3 == 0:
            e2 = engine(grad=False)
            restore.restore_batch(e2, cfg_plain, case["y"], noise_source="device", seed=it)
            e2.close()
        if it % 3 == 1:
            restore.restore_batch(e, cfg_plain, case["y"], noise_source="device", seed=it)
        if it % 3 == 2:
            x = e.to_device(xv)
            for _ in range(6):
                e.unet_forward(x, tv)
            e.sync()
print(f"{os.environ.get('RUN_LABEL', 'dps_repeat')}: {N} iterations, churn={churn}, deviations={bad}, sha={hashlib.sha1(first.tobytes()).hexdigest()[:12]}, "
      f"{time.time() - t0:.1f} s")
e.close()
