"""Drive the LIVE reference functions for one batch of the DiffPIR loop (build container only).

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py and tests that are skipped when
/root/reference is absent.  The loop body below calls the reference's own
utils_model.model_fn / utils_sisr.{pre_calculate,data_solution} / Resizer and re-states
only the glue lines of main_ddpir.py:274-470 that cannot be imported (they live inside
main()), in the same order, so that torch.randn_like is hit in the reference's order.
"""
from __future__ import annotations

import contextlib
from functools import partial

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_import


@contextlib.contextmanager
def patched_randn_like(noise_fn):
    """Route every torch.randn_like (p_sample's and the driver's) through noise_fn, in call order."""
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: noise_fn(t)
    try:
        yield
    finally:
        torch.randn_like = orig


def build_unet(hp, sd, frozen=True):
    """Instantiate the reference UNetModel + diffusion for oracle.unet_oracle.UNetHP `hp`.  frozen=False mirrors
    main_ddpir.py:236-239 for generate_mode DPS_y0: the parameters keep requires_grad (AttentionBlock's CheckpointFunction
    differentiates w.r.t. them and fails otherwise)."""
    ns = ref_import.load()
    model = ns.script_util.create_model(
        image_size=hp.image_size, num_channels=hp.model_channels, num_res_blocks=hp.num_res_blocks,
        channel_mult=",".join(str(int(c)) for c in hp.channel_mult) if hp.channel_mult else "",
        learn_sigma=hp.learn_sigma, class_cond=hp.class_cond, use_checkpoint=False,
        attention_resolutions=hp.attention_resolutions, num_heads=4,
        num_head_channels=hp.num_head_channels, num_heads_upsample=-1, use_scale_shift_norm=True,
        dropout=0.1, resblock_updown=True, use_fp16=False, use_new_attention_order=False)
    if hp.class_cond and hp.num_classes != 1000:
        model.label_emb = torch.nn.Embedding(hp.num_classes, 4 * hp.model_channels)
        model.num_classes = hp.num_classes
    model.load_state_dict(sd)
    model.eval()
    if frozen:
        for _, v in model.named_parameters():
            v.requires_grad = False
    diffusion = ns.script_util.create_gaussian_diffusion(steps=1000, learn_sigma=hp.learn_sigma)
    return model, diffusion


def restore_live(model, diffusion, cfg, y, k=None, mask=None, noise_fn=None, y_label=None, trace=None):
    """main_ddpir.py:259-470 for generate_mode DiffPIR / repaint / vanilla / DPS_y0 (the latter: task 'sr'; call WITHOUT
    torch.no_grad), model_output_type='pred_xstart'."""
    ns = ref_import.load()
    utils_model, sr, Resizer = ns.utils_model, ns.utils_sisr, ns.utils_resizer.Resizer
    T = cfg.T
    # main_ddpir.py:184-190
    betas = torch.from_numpy(np.linspace(cfg.beta_start, cfg.beta_end, T, dtype=np.float32))
    alphas = 1.0 - betas
    alphas_cumprod = np.cumprod(alphas.cpu(), axis=0)
    sqrt_alphas_cumprod = torch.sqrt(alphas_cumprod)
    sqrt_1m_alphas_cumprod = torch.sqrt(1. - alphas_cumprod)
    reduced_alpha_cumprod = torch.div(sqrt_1m_alphas_cumprod, sqrt_alphas_cumprod)
    # :197-200
    nii = getattr(cfg, "noise_init_img", "max")
    t_start = T - 1 if nii == 'max' else utils_model.find_nearest(reduced_alpha_cumprod, 2 * nii / 255)
    # :274-286
    sigmas, sigma_ks, rhos = [], [], []
    for i in range(T):
        sigmas.append(reduced_alpha_cumprod[T - 1 - i])
        sigma_ks.append((sqrt_1m_alphas_cumprod[i] / sqrt_alphas_cumprod[i]))
        rhos.append(cfg.lambda_ * (cfg.sigma ** 2) / (sigma_ks[i] ** 2))
    rhos, sigmas = torch.tensor(rhos), torch.tensor(sigmas)
    model_kwargs = {} if y_label is None else {"y": y_label}
    with patched_randn_like(noise_fn):
        y = y.float()
        B, C = y.shape[0], 3
        H, W = y.shape[2] * cfg.sf, y.shape[3] * cfg.sf
        if cfg.task == "sr":
            degrade_op = Resizer((B, C, H, W), 1 / cfg.sf)
            x = F.interpolate(y, size=(H, W), mode='bicubic', align_corners=False)
            up_sample = partial(F.interpolate, scale_factor=cfg.sf)
        elif cfg.task == "deblur":
            x = y
        else:
            mask = mask.float()
            x = y * mask
        x = sqrt_alphas_cumprod[t_start] * (2 * x - 1) + sqrt_1m_alphas_cumprod[t_start] * torch.randn_like(x)
        if cfg.task in ("sr", "deblur"):
            FB, FBC, F2B, FBFy = sr.pre_calculate(y, k.float(), cfg.sf)
        # :327-335
        if cfg.skip_type == 'uniform':
            skip = T // cfg.iter_num
            seq = [i * skip for i in range(cfg.iter_num)]
            if skip > 1:
                seq.append(T - 1)
        else:
            seq = np.sqrt(np.linspace(0, T ** 2, cfg.iter_num))
            seq = [int(s) for s in list(seq)]
            seq[-1] = seq[-1] - 1
        for i in range(len(seq)):
            curr_sigma = sigmas[seq[i]].cpu().numpy()
            t_i = utils_model.find_nearest(reduced_alpha_cumprod, curr_sigma)
            if t_i > t_start:                                              # main_ddpir.py:346-347
                continue
            gen_mode = getattr(cfg, "generate_mode", "DiffPIR")
            if cfg.task == "inpaint" and gen_mode == 'repaint':            # main_ddpir.py:355-358
                x = (sqrt_alphas_cumprod[t_i] * (2 * y - 1) + sqrt_1m_alphas_cumprod[t_i] * torch.randn_like(x)) * mask \
                    + (1 - mask) * x
            if 'DPS' in gen_mode:                                          # main_ddpir.py:370-373
                x = x.requires_grad_()
                xt, x0 = utils_model.model_fn(x, noise_level=curr_sigma * 255, model_out_type='pred_x_prev_and_start',
                                              model_diffusion=model, diffusion=diffusion, ddim_sample=getattr(cfg, "ddim_sample", False),
                                              alphas_cumprod=alphas_cumprod, **model_kwargs)
            else:
                x0 = utils_model.model_fn(x, noise_level=curr_sigma * 255, model_out_type='pred_xstart',
                                          model_diffusion=model, diffusion=diffusion, ddim_sample=False,
                                          alphas_cumprod=alphas_cumprod, **model_kwargs)
            if trace is not None:
                trace.append(("x0", int(t_i), x0.detach().clone()))
            if seq[i] != seq[-1] and gen_mode == 'DPS_y0':                 # main_ddpir.py:433-438
                measurement = y if cfg.task == "deblur" else 2 * y - 1
                norm_grad, norm = utils_model.grad_and_value(operator=degrade_op, x=x, x_hat=x0, measurement=measurement)
                if trace is not None:
                    trace.append(("norm_grad", int(t_i), norm_grad.clone()))
                x = xt - norm_grad * 1.
                x = x.detach_()
            elif seq[i] != seq[-1] and gen_mode == 'DPS_yt':               # main_ddpir.py:439-445
                y_t = sqrt_alphas_cumprod[t_i] * (2 * y - 1) + sqrt_1m_alphas_cumprod[t_i] * torch.randn_like(y)
                measurement = y_t / 2 + 0.5 if cfg.task == "deblur" else y_t
                norm_grad, norm = utils_model.grad_and_value(operator=degrade_op, x=xt, x_hat=xt, measurement=measurement)
                x = xt - norm_grad * cfg.lambda_ * norm / (rhos[t_i]) * 0.35
                x = x.detach_()
            elif seq[i] != seq[-1]:
                tau = rhos[t_i].float().repeat(1, 1, 1, 1)
                if gen_mode != 'DiffPIR':
                    pass                                                   # main_ddpir.py:385: step 2 is DiffPIR-only
                elif not getattr(cfg, "sub_1_analytic", True):             # main_ddpir.py:420-430: first-order solver
                    x0 = x0.requires_grad_()
                    measurement = y if cfg.task == "deblur" else 2 * y - 1
                    norm_grad, norm = utils_model.grad_and_value(operator=degrade_op, x=x0, x_hat=x0, measurement=measurement)
                    x0 = x0 - norm_grad * norm / (rhos[t_i])
                    x0 = x0.detach_()
                elif cfg.task == "inpaint":
                    x0_p = (mask * (2 * y - 1) + tau * x0).div(mask + tau)
                    x0 = x0 + cfg.guidance_scale * (x0_p - x0)
                elif cfg.task == "deblur" or cfg.sr_mode == 'blur':
                    x0_p = x0 / 2 + 0.5
                    x0_p = sr.data_solution(x0_p.float(), FB, FBC, F2B, FBFy, tau, cfg.sf)
                    x0_p = x0_p * 2 - 1
                    x0 = x0 + cfg.guidance_scale * (x0_p - x0)
                else:
                    for _ in range(cfg.inIter):
                        x0 = x0 / 2 + 0.5
                        x0 = x0 + cfg.gamma * up_sample((y - degrade_op(x0))) / (1 + rhos[t_i])
                        x0 = x0 * 2 - 1
                if not (cfg.task == "inpaint" or gen_mode == 'DiffPIR'):   # main_ddpir.py:448: no re-noising otherwise
                    continue
                t_im1 = utils_model.find_nearest(reduced_alpha_cumprod, sigmas[seq[i + 1]].cpu().numpy())
                eps = (x - sqrt_alphas_cumprod[t_i] * x0) / sqrt_1m_alphas_cumprod[t_i]
                eta_sigma = cfg.eta * sqrt_1m_alphas_cumprod[t_im1] / sqrt_1m_alphas_cumprod[t_i] * torch.sqrt(betas[t_i])
                x = sqrt_alphas_cumprod[t_im1] * x0 + np.sqrt(1 - cfg.zeta) * (torch.sqrt(sqrt_1m_alphas_cumprod[t_im1] ** 2 - eta_sigma ** 2) * eps
                        + eta_sigma * torch.randn_like(x)) + np.sqrt(cfg.zeta) * sqrt_1m_alphas_cumprod[t_im1] * torch.randn_like(x)
                if trace is not None:
                    trace.append(("x", int(t_im1), x.clone()))
        x_0 = (x.detach() / 2 + 0.5)
    return x_0
