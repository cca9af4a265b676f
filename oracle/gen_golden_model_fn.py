"""Golden vectors for utils_model.model_fn (utils/utils_model.py:207-258) from the LIVE reference: pred_xstart with
ddim_sample=False and ddim_sample=True (eta=0) on the tiny UNet.  TEST INFRASTRUCTURE ONLY; build container only
(needs /root/reference).   python -m oracle.gen_golden_model_fn   ->  tests/golden/model_fn.npz"""
import os
import numpy as np
import torch

from . import ref_exec, ref_import, unet_oracle as uo

OUT = os.environ.get("DIFFPIR_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    ns = ref_import.load()
    hp = uo.tiny_hp()
    sd = uo.synth_state_dict(hp, 0)
    model, diffusion = ref_exec.build_unet(hp, sd)
    betas = torch.from_numpy(np.linspace(0.0001, 0.02, 1000, dtype=np.float32))        # main_ddpir.py:184-190
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 3, 32, 32), generator=g)
    out = {"x": x.numpy(), "noise_levels": np.array([0.9, 0.2, 0.02], np.float32)}
    for j, sig in enumerate(out["noise_levels"]):
        for ddim in (False, True):
            draws = []
            def noise_fn(t):
                draws.append(1)
                return torch.zeros_like(t)
            with ref_exec.patched_randn_like(noise_fn), torch.no_grad():
                x0 = ns.utils_model.model_fn(x, noise_level=float(sig) * 255, model_out_type="pred_xstart", model_diffusion=model,
                                             diffusion=diffusion, ddim_sample=ddim, alphas_cumprod=alphas_cumprod)
            out[f"x0_{j}_{'ddim' if ddim else 'psample'}"] = x0.numpy()
            out[f"draws_{j}_{'ddim' if ddim else 'psample'}"] = np.array(len(draws))
    np.savez_compressed(os.path.join(OUT, "model_fn.npz"), **out)
    for k, v in out.items():
        print(k, v.shape if hasattr(v, "shape") else v)
    print("max |ddim - psample| per level:",
          [float(np.abs(out[f"x0_{j}_ddim"] - out[f"x0_{j}_psample"]).max()) for j in range(3)],
          "draws:", [(int(out[f"draws_{j}_psample"]), int(out[f"draws_{j}_ddim"])) for j in range(3)])


if __name__ == "__main__":
    main()
