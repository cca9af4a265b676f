"""ms per NFE of generate_mode DPS_y0 (forward + UNet input-gradient pass), FFHQ topology, x4 SISR to 256^2.  DPIR_DGRAD_F32=1 forces
the fp32-MFMA dgrad kernels for an A/B.  usage: python tools/dps_time.py [B] [nfe]   (GPU box only)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import restore, synth, script_util, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nfe = int(sys.argv[2]) if len(sys.argv) > 2 else 6
e = diffpir_amd.Engine(0); e.set_precision(os.environ.get("DIFFPIR_PRECISION", "f16x3")); e.enable_grad()
hp = weights.model_hp("ffhq")
m = script_util.create_model(**weights.create_model_kwargs(hp), engine=e); m.load_state_dict(weights.synth_state_dict(hp, 0))
case = synth.make_case("sr", B, 256, 256, seed=400, sf=4)
cfg = restore.LoopConfig(task="sr", iter_num=nfe, lambda_=6.0, zeta=0.25, sf=4, sr_mode="cubic", generate_mode="DPS_y0")
y = e.to_device(case["y"])
restore.restore_batch(e, cfg, y, noise_source="device", seed=1)
t0 = time.perf_counter(); o = restore.restore_batch(e, cfg, y, noise_source="device", seed=1); e.sync(); dt = time.perf_counter() - t0
e.prof_enable(True); e.prof_reset(); restore.restore_batch(e, cfg, y, noise_source="device", seed=1); e.sync(); prof = e.prof_read(); e.prof_enable(False)
print(f"{os.environ.get('RUN_LABEL','dps')}: B={B} {dt/(nfe-1)*1e3:.2f} ms per NFE (forward + backward), finite={bool(np.isfinite(o.numpy()).all())} | "
      f"{ {k: round(v[0]/(nfe-1), 2) for k, v in prof.items() if v[1]} }")
