"""Golden vectors for the non-default generate modes of the inpainting loop (main_ddpir.py:349-358, 384-385, 448) from the
LIVE reference functions: 'repaint' and 'vanilla', 6 NFE, tiny UNet, same inputs as tests/golden/loops.npz ('inpaint_*').
TEST INFRASTRUCTURE ONLY; build container only (needs /root/reference).
python -m oracle.gen_golden_modes   ->  tests/golden/loops_modes.npz"""
import os
import numpy as np
import torch

from . import ref_exec, unet_oracle as uo, diffpir_oracle as do

OUT = os.environ.get("DIFFPIR_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def seeded_noise_fn(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda like: torch.randn(like.shape, generator=g, dtype=torch.float32)


def main():
    hp = uo.tiny_hp()
    sd = uo.synth_state_dict(hp, 0)
    model, diffusion = ref_exec.build_unet(hp, sd)
    g = np.load(os.path.join(OUT, "loops.npz"))
    y, mask = torch.from_numpy(g["inpaint_y"]), torch.from_numpy(g["inpaint_mask"])
    out = {}
    for mode, seed in (("repaint", 45), ("vanilla", 46)):
        cfg = do.LoopConfig(task="inpaint", iter_num=6, noise_level_img=0.0, lambda_=1.0, zeta=1.0, generate_mode=mode)
        with torch.no_grad():
            ref = ref_exec.restore_ref(model, diffusion, cfg, y, mask=mask, noise_fn=seeded_noise_fn(seed)).numpy()
            ora = do.restore(sd, hp, cfg, y, mask=mask, noise_fn=seeded_noise_fn(seed)).numpy()
        out[f"inpaint_{mode}_out"] = ref
        out[f"inpaint_{mode}_seed"] = np.array(seed)
        print(mode, "live-reference vs oracle max abs diff:", float(np.abs(ref - ora).max()), "range", float(ref.min()), float(ref.max()))
    # skip_type: uniform (main_ddpir.py:328-331: seq = [i*skip] + [T-1]) on the DiffPIR inpainting loop, 5 + 1 steps
    cfg = do.LoopConfig(task="inpaint", iter_num=5, noise_level_img=0.0, lambda_=1.0, zeta=1.0, skip_type="uniform")
    with torch.no_grad():
        ref = ref_exec.restore_ref(model, diffusion, cfg, y, mask=mask, noise_fn=seeded_noise_fn(47)).numpy()
        ora = do.restore(sd, hp, cfg, y, mask=mask, noise_fn=seeded_noise_fn(47)).numpy()
    out["inpaint_uniform_out"] = ref
    out["inpaint_uniform_seed"] = np.array(47)
    print("uniform live-reference vs oracle max abs diff:", float(np.abs(ref - ora).max()))
    np.savez_compressed(os.path.join(OUT, "loops_modes.npz"), **out)


if __name__ == "__main__":
    main()
