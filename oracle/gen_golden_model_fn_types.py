"""Golden vectors for the model_fn output types beyond 'pred_xstart' and for the DPS loop with config.ddim_sample (round 4) from the
LIVE reference (utils/utils_model.py:207-258, gaussian_diffusion.py:395-439, 537-585; main_ddpir.py:370-373, 434-438):

  * `<type>_<level>_<psample|ddim>`: model_fn(x, noise_level, model_out_type = 'pred_x_prev_and_start' | 'epsilon' | 'score') on the tiny
    UNet with a FIXED randn_like tensor (`noise`);
  * `dpsddim_*`: a whole generate_mode 'DPS_y0' restoration with ddim_sample=True (task sr x4, tiny topology, B = 2, 5 NFE).

TEST INFRASTRUCTURE ONLY; build container only (needs /root/reference).
    python -m oracle.gen_golden_model_fn_types   ->  tests/golden/model_fn_types.npz
"""
import os
import numpy as np
import torch

from . import ref_exec, ref_import, unet_oracle as uo, diffpir_oracle as do

OUT = os.environ.get("DIFFPIR_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    from diffpir_amd import synth
    ns = ref_import.load()
    hp = uo.tiny_hp()
    sd = uo.synth_state_dict(hp, 0)
    model, diffusion = ref_exec.build_unet(hp, sd)
    betas = torch.from_numpy(np.linspace(0.0001, 0.02, 1000, dtype=np.float32))        # main_ddpir.py:184-190
    alphas_cumprod = np.cumprod((1.0 - betas).cpu(), axis=0)
    g = torch.Generator().manual_seed(12)
    x = torch.randn((2, 3, 32, 32), generator=g)
    noise = torch.randn((2, 3, 32, 32), generator=g)
    out = {"x": x.numpy(), "noise": noise.numpy(), "noise_levels": np.array([0.9, 0.05], np.float32)}
    for j, sig in enumerate(out["noise_levels"]):
        for ddim in (False, True):
            tag = f"{j}_{'ddim' if ddim else 'psample'}"
            for typ in ("pred_x_prev_and_start", "epsilon", "score"):
                with ref_exec.patched_randn_like(lambda t: noise.clone()), torch.no_grad():
                    r = ns.utils_model.model_fn(x, noise_level=float(sig) * 255, model_out_type=typ, model_diffusion=model,
                                                diffusion=diffusion, ddim_sample=ddim, alphas_cumprod=alphas_cumprod)
                if typ == "pred_x_prev_and_start":
                    out[f"xt_{tag}"], out[f"x0_{tag}"] = r[0].numpy(), r[1].numpy()
                else:
                    out[f"{typ}_{tag}"] = r.numpy()
    # DPS_y0 with ddim_sample=True through the reference's own model_fn / Resizer / grad_and_value
    model, diffusion = ref_exec.build_unet(hp, sd, frozen=False)
    case = synth.make_case("sr", 2, 64, 64, seed=3, sf=4)
    cfg = do.LoopConfig("sr", 5, 12.75 / 255, 6.0, 0.25, sf=4, sr_mode="cubic", generate_mode="DPS_y0", ddim_sample=True)
    y, k = torch.from_numpy(case["y"]), torch.from_numpy(case["k"])

    def seeded(seed):
        gg = torch.Generator().manual_seed(seed)
        return lambda like: torch.randn(like.shape, generator=gg, dtype=torch.float32)
    ref = ref_exec.restore_ref(model, diffusion, cfg, y, k=k, noise_fn=seeded(84)).numpy()
    ora = do.restore_dps_y0(sd, hp, cfg, y, noise_fn=seeded(84)).numpy()
    out.update(dpsddim_y=case["y"], dpsddim_gt=case["gt"], dpsddim_out=ref, dpsddim_seed=np.array(84), dpsddim_nfe=np.array(5))
    print("DPS_y0 + ddim_sample 5-NFE loop: live reference vs oracle max abs diff", float(np.abs(ref - ora).max()), "range", float(np.abs(ref).max()))
    np.savez_compressed(os.path.join(OUT, "model_fn_types.npz"), **out)
    for kk, v in out.items():
        print(kk, v.shape, float(np.abs(v).max()))


if __name__ == "__main__":
    main()
