"""Small fixed workload for rocprofv3 (kernel trace / PMC passes): 3 UNet forwards at 256x256 (PROF_MODEL ffhq | imagenet256, PROF_B,
DIFFPIR_PRECISION f16x3 | f32 | f16x1) plus 3 FFT-prox applications (PROF_SF 1 | 4).  PROF_UNET=0 PROF_SIZE=512: the data step alone at 512x512
(configs[4]'s shard).  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from diffpir_amd import script_util, weights, synth, utils_sisr as sr

B = int(os.environ.get("PROF_B", "16"))
eng = diffpir_amd.Engine(0)
eng.set_precision(os.environ.get("DIFFPIR_PRECISION", "f16x3"))
hp = weights.model_hp(os.environ.get("PROF_MODEL", "ffhq"))
SF = int(os.environ.get("PROF_SF", "1"))
S = int(os.environ.get("PROF_SIZE", "256"))
if os.environ.get("PROF_UNET", "1") == "1":
    model = script_util.create_model(**weights.create_model_kwargs(hp), engine=eng)
    model.load_state_dict(weights.synth_state_dict(hp, 0))
    x = eng.to_device(np.random.default_rng(0).standard_normal((B, 3, S, S)).astype(np.float32))
    t = np.full(B, 500)
    out = eng.unet_forward(x, t)
    eng.sync()
    for _ in range(2):
        eng.unet_forward(x, t, out=out)
    eng.sync()
case = synth.make_case("deblur", B, S, S, seed=1, ksize=61) if SF == 1 else synth.make_case("sr", B, S, S, seed=1, sf=SF)
y, k = eng.to_device(case["y"]), eng.to_device(case["k"])
pre = sr.pre_calculate(y, k, SF)
x0 = eng.to_device(case["gt"] * 2 - 1)
for _ in range(3):
    eng._check(eng.lib.dpir_prox_fft_apply(eng.h, pre[0].spectra.handle, x0.ptr, 0.01, 1.0))
eng.sync()
print("done")
