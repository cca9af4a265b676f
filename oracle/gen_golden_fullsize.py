"""Golden vectors at the BENCHED sizes and for the schedule corner cases, from the LIVE reference functions:

  * `ffhq256_out`   : guided_diffusion UNetModel.forward, FFHQ topology (diffusion_ffhq_10m hyper-parameters), ONE 256x256
                      image -- pins the full-size network (the 256^2 layers carry 50 % of the FLOPs) to the reference itself;
  * `c2_*`          : BASELINE config 2 in miniature: FFHQ topology, 256x256, 61x61 Gaussian PSF, sigma 12.75/255,
                      lambda 7, zeta 0.3, B=1, 4 NFE through model_fn -> data_solution -> re-noise (main_ddpir.py:341-470);
  * `tstart_*`      : noise_init_img != 'max' (t_start below T-1: steps above it are skipped, main_ddpir.py:197-200, 346);
  * `duplast_*`     : quad skipping with iter_num > T/2, where TWO steps satisfy seq[i] == seq[-1] (both dead denoiser calls).

TEST INFRASTRUCTURE ONLY; build container only (needs /root/reference).
    python -m oracle.gen_golden_fullsize   ->  tests/golden/fullsize.npz
"""
import os
import numpy as np
import torch

from . import ref_exec, unet_oracle as uo, diffpir_oracle as do

OUT = os.environ.get("DIFFPIR_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def seeded_noise_fn(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda like: torch.randn(like.shape, generator=g, dtype=torch.float32)


def main():
    from diffpir_amd import synth              # deterministic numpy generators of synthetic inputs (inputs are stored too)
    torch.set_num_threads(8)
    out = {}
    hp = uo.ffhq_hp()
    sd = uo.synth_state_dict(hp, 0)
    model, diffusion = ref_exec.build_unet(hp, sd)
    g = torch.Generator().manual_seed(11)
    x = torch.randn((1, 3, 256, 256), generator=g)
    t = torch.tensor([417])
    with torch.no_grad():
        ref = model(x, t).numpy()
        ora = uo.unet_forward(sd, hp, x, t).numpy()
    out["ffhq256_x_seed"], out["ffhq256_t"], out["ffhq256_out"] = np.array(11), t.numpy(), ref
    print("ffhq256 forward: live reference vs oracle max abs diff", float(np.abs(ref - ora).max()), "range", float(np.abs(ref).max()))

    case = synth.make_case("deblur", 1, 256, 256, seed=5, ksize=61)
    cfg = do.LoopConfig("deblur", 4, 12.75 / 255, 7.0, 0.3)
    y, k = torch.from_numpy(case["y"]), torch.from_numpy(case["k"])
    with torch.no_grad():
        ref = ref_exec.restore_ref(model, diffusion, cfg, y, k=k, noise_fn=seeded_noise_fn(51)).numpy()
        ora = do.restore(sd, hp, cfg, y, k=k, noise_fn=seeded_noise_fn(51)).numpy()
        # the reference's OWN fp32 rounding noise on this case: distance to the same loop with the (ill-conditioned) closed-form
        # prox evaluated in float64 -- the yardstick of the conditioning-aware parity bound (tests/gpu_common.py::fft_prox_parity)
        exact = do.restore(sd, hp, cfg, y, k=k, noise_fn=seeded_noise_fn(51), exact_prox=True).numpy()
    d = ref - exact
    out.update(c2_y=case["y"], c2_k=case["k"], c2_gt=case["gt"], c2_out=ref, c2_seed=np.array(51),
               c2_floor_max=np.array(np.abs(d).max()), c2_floor_rms=np.array(np.sqrt(np.mean(d * d))))
    print("c2 4-NFE loop: live reference vs oracle max abs diff", float(np.abs(ref - ora).max()),
          "| reference vs exact-prox loop: max", float(np.abs(d).max()), "rms", float(np.sqrt(np.mean(d * d))))

    # schedule corner cases on the tiny UNet, inputs of tests/golden/loops.npz
    hp = uo.tiny_hp()
    sd = uo.synth_state_dict(hp, 0)
    model, diffusion = ref_exec.build_unet(hp, sd)
    lg = np.load(os.path.join(OUT, "loops.npz"))
    y, mask = torch.from_numpy(lg["inpaint_y"]), torch.from_numpy(lg["inpaint_mask"])
    cfg = do.LoopConfig(task="inpaint", iter_num=8, noise_level_img=0.0, lambda_=1.0, zeta=1.0, noise_init_img=60.0)
    with torch.no_grad():
        ref = ref_exec.restore_ref(model, diffusion, cfg, y, mask=mask, noise_fn=seeded_noise_fn(52)).numpy()
        ora = do.restore(sd, hp, cfg, y, mask=mask, noise_fn=seeded_noise_fn(52)).numpy()
    out.update(tstart_out=ref, tstart_seed=np.array(52), tstart_noise_init_img=np.array(60.0))
    print("t_start loop: live reference vs oracle max abs diff", float(np.abs(ref - ora).max()))

    cfg = do.LoopConfig(task="inpaint", iter_num=520, noise_level_img=0.0, lambda_=1.0, zeta=1.0)
    seq = do.make_seq(1000, 520, "quad")
    n_last = sum(1 for s_ in seq if s_ == seq[-1])
    assert n_last >= 2, n_last
    with torch.no_grad():
        ref = ref_exec.restore_ref(model, diffusion, cfg, y, mask=mask, noise_fn=seeded_noise_fn(53)).numpy()
        ora = do.restore(sd, hp, cfg, y, mask=mask, noise_fn=seeded_noise_fn(53)).numpy()
    out.update(duplast_out=ref, duplast_seed=np.array(53), duplast_n_last=np.array(n_last))
    print("duplicate-last loop (", n_last, "final steps): live reference vs oracle max abs diff", float(np.abs(ref - ora).max()))
    np.savez_compressed(os.path.join(OUT, "fullsize.npz"), **out)


if __name__ == "__main__":
    main()
