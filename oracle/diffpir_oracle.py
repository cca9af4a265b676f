"""Oracle: restatement of the DiffPIR restoration loop and its operators (torch-CPU fp32 / numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows (reference paths relative to /root/reference):
  * driver schedule tables      main_ddpir.py:184-190 (float32), 274-286 (rhos/sigmas)
  * diffusion tables            guided_diffusion/gaussian_diffusion.py:27-35, 133-151 (float64)
  * timestep sequence           main_ddpir.py:327-335, 342-344, 451
  * model_fn -> pred_xstart     utils/utils_model.py:202-258, gaussian_diffusion.py:232-333, 395-439
  * FFT prox                    utils/utils_sisr.py:9-95
  * masked prox                 main_ddpir.py:392-394
  * cubic IBP prox + Resizer    main_ddpir.py:401-406, utils/utils_resizer.py:9-178
  * re-noise                    main_ddpir.py:448-456
  * init / output               main_ddpir.py:291-315, 470, 482; utils/utils_image.py:238-242
  * PSNR                        utils/utils_image.py:601-610
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import unet_oracle as uo


# ------------------------------------------------------------------ schedules
class DriverTables:
    """main_ddpir.py:184-190 -- float32 tables, cumprod done by numpy on the float32 tensor."""

    def __init__(self, beta_start=0.0001, beta_end=0.02, T=1000):
        betas = torch.from_numpy(np.linspace(beta_start, beta_end, T, dtype=np.float32))
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)                      # torch tensor in, torch tensor out
        ac = torch.as_tensor(ac)
        self.T = T
        self.betas = betas
        self.alphas = alphas
        self.alphas_cumprod = ac
        self.sqrt_ac = torch.sqrt(ac)
        self.sqrt_1m_ac = torch.sqrt(1.0 - ac)
        self.reduced = torch.div(self.sqrt_1m_ac, self.sqrt_ac)   # sigma-bar


class DiffusionTables:
    """gaussian_diffusion.py:27-35,133-151 -- float64 linear schedule used inside p_mean_variance."""

    def __init__(self, T=1000):
        scale = 1000 / T
        betas = np.linspace(scale * 0.0001, scale * 0.02, T, dtype=np.float64)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.sqrt_recip_ac = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_ac = np.sqrt(1.0 / ac - 1)
        # posterior q(x_{t-1} | x_t, x_0) and the learned-range variance bounds (gaussian_diffusion.py:142-167, 268-276)
        ac_prev = np.append(1.0, ac[:-1])
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.posterior_log_variance_clipped = np.log(np.append(post_var[1], post_var[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - ac_prev) * np.sqrt(1.0 - betas) / (1.0 - ac)
        self.log_betas = np.log(betas)
        self.alphas_cumprod_prev = ac_prev


def loader_strides(t):
    """The same values with the strides the reference's DataLoader path gives a batch: CustomDataset yields NHWC numpy arrays and
    util.single2tensor4_batch (utils_image.py:255-256) / main_ddpir.py:295 only `.permute(0, 3, 1, 2)` them, so `y`, `mask` and hence
    `x` enter the loop as NCHW VIEWS of NHWC memory (channels-last).  ATen's CPU convolution / interpolation kernels are chosen by
    memory format and round differently in the last bits (measured: up to 8e-5 after a 5-step loop on the tiny network, 0.0 once the
    strides match -- profiles/r05/oracle_glue_equivalence.log), so a bit-level comparison with a reference-executed fixture has to
    start from the same layout."""
    return None if t is None else t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def find_nearest(array, value) -> int:
    """utils_model.py:202-205."""
    array = np.asarray(array)
    return int(np.abs(array - value).argmin())


def make_seq(T: int, iter_num: int, skip_type: str = "quad") -> List[int]:
    """main_ddpir.py:327-335."""
    skip = T // iter_num
    if skip_type == "uniform":
        seq = [i * skip for i in range(iter_num)]
        if skip > 1:
            seq.append(T - 1)
    else:
        s = np.sqrt(np.linspace(0, T ** 2, iter_num))
        seq = [int(v) for v in list(s)]
        seq[-1] = seq[-1] - 1
    return seq


@dataclass
class LoopConfig:
    """The YAML keys that reach the loop (configs/*.yaml; main_ddpir.py:138-158)."""
    task: str = "deblur"                 # deblur | sr | inpaint
    iter_num: int = 100
    noise_level_img: float = 12.75 / 255.0   # already divided by 255 (main_ddpir.py:138)
    lambda_: float = 7.0
    zeta: float = 0.3
    eta: float = 0.0
    guidance_scale: float = 1.0
    sf: int = 1
    sr_mode: str = "blur"
    inIter: int = 1
    gamma: float = 0.01
    skip_type: str = "quad"
    T: int = 1000
    beta_start: float = 0.0001
    beta_end: float = 0.02
    generate_mode: str = "DiffPIR"       # DiffPIR | repaint | vanilla (the latter two: inpainting only) | DPS_y0 | DPS_yt (task sr)
    sub_1_analytic: bool = True          # False: first-order data step (main_ddpir.py:420-430; runnable for task sr)
    noise_init_img: object = "max"       # 'max' or a noise level in /255 units (main_ddpir.py:197-200)
    ddim_sample: bool = False            # DPS modes: xt from ddim_sample(eta=0) instead of p_sample (utils_model.py:219-240)

    @property
    def sigma(self):                     # main_ddpir.py:141
        return max(0.001, self.noise_level_img)

    def t_start(self, dt) -> int:        # main_ddpir.py:197-200
        if self.noise_init_img == "max":
            return self.T - 1
        return find_nearest(dt.reduced, 2 * float(self.noise_init_img) / 255)


def step_tables(cfg: LoopConfig):
    """Per-step scalars (SURVEY 8 a-S): list of dicts with t_i, t_im1, tau, is_last."""
    dt = DriverTables(cfg.beta_start, cfg.beta_end, cfg.T)
    T = cfg.T
    # main_ddpir.py:274-286
    sigmas = [dt.reduced[T - 1 - i] for i in range(T)]
    if cfg.generate_mode == "DiffPIR":                      # main_ddpir.py:279-283 (model_out_type is 'pred_xstart' throughout the oracle)
        sigma_ks = [dt.sqrt_1m_ac[i] / dt.sqrt_ac[i] for i in range(T)]
    else:
        sigma_ks = [torch.sqrt(dt.betas[i] / dt.alphas[i]) for i in range(T)]
    rhos = [cfg.lambda_ * (cfg.sigma ** 2) / (sigma_ks[i] ** 2) for i in range(T)]
    rhos = torch.tensor(rhos)
    sigmas = torch.tensor(sigmas)
    seq = make_seq(T, cfg.iter_num, cfg.skip_type)
    steps = []
    for i in range(len(seq)):
        curr_sigma = sigmas[seq[i]].cpu().numpy()
        t_i = find_nearest(dt.reduced, curr_sigma)
        last = seq[i] == seq[-1]
        t_im1 = None if last else find_nearest(dt.reduced, sigmas[seq[i + 1]].cpu().numpy())
        steps.append(dict(i=i, t_i=t_i, t_im1=t_im1, curr_sigma=curr_sigma, tau=rhos[t_i].float(), last=last))
    return dt, steps


# ------------------------------------------------------------------ denoiser plug
def pred_xstart_from_eps(x, eps, t: int, dtab: DiffusionTables):
    """gaussian_diffusion.py:328-333 + clamp :297: float64 table entries cast .float()."""
    c1 = torch.tensor(dtab.sqrt_recip_ac[t]).float()
    c2 = torch.tensor(dtab.sqrt_recipm1_ac[t]).float()
    return (c1 * x - c2 * eps).clamp(-1, 1)


def model_fn_xstart(sd, hp, x, noise_level, dt: DriverTables, dtab: DiffusionTables,
                    noise_fn: Optional[Callable] = None, y_label=None):
    """utils_model.py:207-258 with model_out_type='pred_xstart'; ddim_sample False or True (eta=0) give the same x0 and the
    same single randn_like draw (gaussian_diffusion.py:395-439 vs 537-585; pinned by tests/golden/model_fn.npz).
    noise_fn(x) mirrors the (dead but RNG-consuming) randn_like in p_sample (gaussian_diffusion.py:430)."""
    t_step = find_nearest(dt.reduced, noise_level / 255.0)
    vec_t = torch.tensor([t_step] * x.shape[0])
    out = uo.unet_forward(sd, hp, x, vec_t, y_label)
    eps = out[:, :3]
    x0 = pred_xstart_from_eps(x, eps, t_step, dtab)
    if noise_fn is not None:
        noise_fn(x)
    return x0


# ------------------------------------------------------------------ FFT prox
def splits(a, sf):
    """utils_sisr.py:9-19: [N,C,H,W] -> [N,C,H/sf,W/sf,sf*sf] alias blocks."""
    b = torch.stack(torch.chunk(a, sf, dim=2), dim=4)
    return torch.cat(torch.chunk(b, sf, dim=3), dim=4)


def p2o(psf, shape):
    """utils_sisr.py:22-41: zero-pad PSF to `shape`, circularly centre it, fft2."""
    otf = torch.zeros(psf.shape[:-2] + tuple(shape)).type_as(psf)
    otf[..., :psf.shape[2], :psf.shape[3]].copy_(psf)
    for axis, n in enumerate(psf.shape[2:]):
        otf = torch.roll(otf, -int(n / 2), dims=axis + 2)
    return torch.fft.fftn(otf, dim=(-2, -1))


def pre_calculate(y, k, sf):
    """utils_sisr.py:78-95."""
    h, w = y.shape[-2:]
    FB = p2o(k, (h * sf, w * sf))
    FBC = torch.conj(FB)
    F2B = torch.pow(torch.abs(FB), 2)
    STy = torch.zeros((y.shape[0], y.shape[1], h * sf, w * sf)).type_as(y)
    STy[..., 0::sf, 0::sf].copy_(y)
    FBFy = FBC * torch.fft.fftn(STy, dim=(-2, -1))
    return FB, FBC, F2B, FBFy


def data_solution(x, FB, FBC, F2B, FBFy, alpha, sf):
    """utils_sisr.py:65-75: closed-form argmin ||y - S(k*x)||^2 + alpha ||x - z||^2."""
    FR = FBFy + torch.fft.fftn(alpha * x, dim=(-2, -1))
    x1 = FB.mul(FR)
    FBR = torch.mean(splits(x1, sf), dim=-1, keepdim=False)
    invW = torch.mean(splits(F2B, sf), dim=-1, keepdim=False)
    invWBR = FBR.div(invW + alpha)
    FCBinvWBR = FBC * invWBR.repeat(1, 1, sf, sf)
    FX = (FR - FCBinvWBR) / alpha
    return torch.real(torch.fft.ifftn(FX, dim=(-2, -1)))


def prox_fft(x0, pre, tau, sf, guidance=1.0, exact=False):
    """main_ddpir.py:395-400.  exact=True evaluates the same closed form in float64 (spectra `pre` computed from float64
    inputs by the caller) and rounds once at the end: the arithmetic-free yardstick against which the fp32 rounding noise of
    the reference's own evaluation -- and the engine's -- is measured (the expression is ill-conditioned at small tau)."""
    FB, FBC, F2B, FBFy = pre
    x0_p = x0 / 2 + 0.5
    if exact:
        x0_p = data_solution(x0_p.double(), FB, FBC, F2B, FBFy, tau.double(), sf).float()
    else:
        x0_p = data_solution(x0_p.float(), FB, FBC, F2B, FBFy, tau, sf)
    x0_p = x0_p * 2 - 1
    return x0 + guidance * (x0_p - x0)


def prox_mask(x0, y, mask, tau, guidance=1.0):
    """main_ddpir.py:392-394."""
    x0_p = (mask * (2 * y - 1) + tau * x0).div(mask + tau)
    return x0 + guidance * (x0_p - x0)


# ------------------------------------------------------------------ Resizer (cubic, antialiased)
def _cubic(x):
    ax = np.abs(x)
    ax2, ax3 = ax ** 2, ax ** 3
    return ((1.5 * ax3 - 2.5 * ax2 + 1) * (ax <= 1) +
            (-0.5 * ax3 + 2.5 * ax2 - 4 * ax + 2) * ((1 < ax) & (ax <= 2)))


def resizer_contributions(in_len: int, out_len: int, scale: float):
    """utils_resizer.py:104-167 for kernel='cubic', antialiasing when scale<1.
    Returns (weights [out,taps] float64, indices [out,taps] int)."""
    aa = scale < 1
    kern = (lambda a: scale * _cubic(scale * a)) if aa else _cubic
    kw = 4.0 / scale if aa else 4.0
    outc = np.arange(1, out_len + 1)
    shifted = outc - (out_len - in_len * scale) / 2
    match = shifted / scale + 0.5 * (1 - 1 / scale)
    left = np.floor(match - kw / 2)
    ekw = np.ceil(kw) + 2
    fov = np.squeeze(np.int16(np.expand_dims(left, 1) + np.arange(ekw) - 1))
    w = kern(1.0 * np.expand_dims(match, 1) - fov - 1)
    sw = np.sum(w, axis=1)
    sw[sw == 0] = 1.0
    w = 1.0 * w / np.expand_dims(sw, 1)
    mirror = np.uint(np.concatenate((np.arange(in_len), np.arange(in_len - 1, -1, step=-1))))
    fov = mirror[np.mod(fov, mirror.shape[0])]
    nz = np.nonzero(np.any(w, axis=0))
    w = np.squeeze(w[:, nz])
    fov = np.squeeze(fov[:, nz])
    return w, fov.astype(np.int64)


def resizer_apply(x, sf_inv: float):
    """utils_resizer.py:55-74 for a 4-D tensor and scalar scale: both spatial dims have equal
    scale so the stable argsort processes dim 2 (H) first, then dim 3 (W)."""
    out = x
    for dim in (2, 3):
        n = out.shape[dim]
        m = int(np.ceil(n * sf_inv))
        w, fov = resizer_contributions(n, m, sf_inv)
        w_t = torch.tensor(w.T, dtype=torch.float32)                   # [taps,out]
        fov_t = torch.tensor(fov.T.astype(np.int32), dtype=torch.long)   # [taps,out]
        xt = torch.transpose(out, dim, 0)
        wv = w_t.reshape(list(w_t.shape) + [1] * 3)
        xt = torch.sum(xt[fov_t] * wv, dim=0)
        out = torch.transpose(xt, dim, 0)
    return out


def prox_ibp(x0, y, rho, sf, gamma, in_iter):
    """main_ddpir.py:401-406: iterative back-projection; up-sampler is F.interpolate default (nearest)."""
    for _ in range(in_iter):
        x0 = x0 / 2 + 0.5
        x0 = x0 + gamma * F.interpolate(y - resizer_apply(x0, 1.0 / sf), scale_factor=sf) / (1 + rho)
        x0 = x0 * 2 - 1
    return x0


# ------------------------------------------------------------------ loop
def renoise(x, x0, dt: DriverTables, t_i, t_im1, eta, zeta, n1, n2):
    """main_ddpir.py:451-456 verbatim arithmetic order (float32 tensors x numpy float64 sqrt)."""
    eps = (x - dt.sqrt_ac[t_i] * x0) / dt.sqrt_1m_ac[t_i]
    eta_sigma = eta * dt.sqrt_1m_ac[t_im1] / dt.sqrt_1m_ac[t_i] * torch.sqrt(dt.betas[t_i])
    return dt.sqrt_ac[t_im1] * x0 + np.sqrt(1 - zeta) * (
        torch.sqrt(dt.sqrt_1m_ac[t_im1] ** 2 - eta_sigma ** 2) * eps + eta_sigma * n1) \
        + np.sqrt(zeta) * dt.sqrt_1m_ac[t_im1] * n2


def init_x(cfg: LoopConfig, y, mask, dt: DriverTables, noise0, t_start=None):
    """main_ddpir.py:293-315."""
    if t_start is None:
        t_start = cfg.T - 1
    if cfg.task == "sr":
        x = F.interpolate(y, size=(y.shape[2] * cfg.sf, y.shape[3] * cfg.sf), mode="bicubic", align_corners=False)
    elif cfg.task == "deblur":
        x = y
    else:
        x = y * mask
    return dt.sqrt_ac[t_start] * (2 * x - 1) + dt.sqrt_1m_ac[t_start] * noise0


def tensor2uint_batch(x01):
    """utils_image.py:238-242: clamp, NCHW->NHWC, *255 round -> u8."""
    img = x01.float().clamp(0, 1).cpu().numpy()
    img = np.transpose(img, (0, 2, 3, 1))
    return np.uint8((img * 255.0).round())


def p_sample_prev_and_start(sd, hp, x, t_step: int, dtab: DiffusionTables, noise, y_label=None, ddim=False):
    """utils_model.model_fn(..., model_out_type='pred_x_prev_and_start') = GaussianDiffusion.p_sample with the LEARNED_RANGE
    variance (gaussian_diffusion.py:232-326, 395-439): returns (sample, pred_xstart), differentiable w.r.t. x."""
    vec_t = torch.tensor([t_step] * x.shape[0])
    out = uo.unet_forward(sd, hp, x, vec_t, y_label)
    eps, v = out[:, :3], out[:, 3:]
    min_log = torch.tensor(dtab.posterior_log_variance_clipped[t_step]).float()
    max_log = torch.tensor(dtab.log_betas[t_step]).float()
    frac = (v + 1) / 2
    log_var = frac * max_log + (1 - frac) * min_log
    x0 = pred_xstart_from_eps(x, eps, t_step, dtab)
    if ddim:     # ddim_sample(eta=0), gaussian_diffusion.py:537-585: eps re-derived from the clamped x0, sigma = 0 (the draw is consumed)
        c1 = torch.tensor(dtab.sqrt_recip_ac[t_step]).float()
        c2 = torch.tensor(dtab.sqrt_recipm1_ac[t_step]).float()
        eps2 = (c1 * x - x0) / c2
        abp = torch.tensor(dtab.alphas_cumprod_prev[t_step]).float()
        return x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - 0.0) * eps2, x0
    mean = torch.tensor(dtab.posterior_mean_coef1[t_step]).float() * x0 + torch.tensor(dtab.posterior_mean_coef2[t_step]).float() * x
    nonzero = 0.0 if t_step == 0 else 1.0
    return mean + nonzero * torch.exp(0.5 * log_var) * noise, x0


def restore_dps_y0(sd, hp, cfg: LoopConfig, y, noise_fn: Callable, y_label=None, trace: Optional[list] = None):
    """generate_mode 'DPS_y0' / 'DPS_yt' for task 'sr' (main_ddpir.py:370-373, 433-445 with utils_model.grad_and_value :390-394; the
    deblurring variant cannot run in the reference as shipped, SURVEY Q1, and the inpainting branch never defines xt):
        DPS_y0:  xt, x0 = p_sample(x);   norm = || (2y - 1) - Resizer(x0) ||_2 over the WHOLE batch;   x <- xt - d norm / d x
        DPS_yt:  y_t = sa_t (2y - 1) + s1m_t n;   norm = || y_t - Resizer(xt) ||_2;   x <- xt - d norm / d xt * lambda * norm / rho_t * 0.35
                 (no backward through the network)
    No re-noising (main_ddpir.py:448 is DiffPIR / inpainting only); the final step's denoiser call is dead.  RNG order: init,
    then per step the p_sample draw [and, DPS_yt, the y_t draw]."""
    if cfg.task != "sr":
        raise ValueError("DPS_y0 is runnable in the reference for task 'sr' only")
    dt, steps = step_tables(cfg)
    dtab = DiffusionTables(cfg.T)
    y = loader_strides(y.float())                   # the memory layout the reference's loader gives (see loader_strides)
    t_start = cfg.t_start(dt)
    H, W = y.shape[2] * cfg.sf, y.shape[3] * cfg.sf
    x = init_x(cfg, y, None, dt, noise_fn(torch.empty(y.shape[0], 3, H, W)), t_start)
    for st in steps:
        t_i = st["t_i"]
        if t_i > t_start:
            continue
        yt_mode = cfg.generate_mode == "DPS_yt"
        x = x.detach()
        if not yt_mode:
            x = x.requires_grad_()
        t_step = find_nearest(dt.reduced, st["curr_sigma"] * 255 / 255.0)
        xt, x0 = p_sample_prev_and_start(sd, hp, x, t_step, dtab, noise_fn(x), y_label, ddim=cfg.ddim_sample)
        if trace is not None:
            trace.append(("x0", t_i, x0.detach().clone()))
        if not st["last"] and yt_mode:
            y_t = dt.sqrt_ac[t_i] * (2 * y - 1) + dt.sqrt_1m_ac[t_i] * noise_fn(y)
            xt = xt.detach().requires_grad_()
            norm = torch.linalg.norm(y_t - resizer_apply(xt, 1.0 / cfg.sf))
            norm_grad = torch.autograd.grad(outputs=norm, inputs=xt)[0]
            x = (xt - norm_grad * cfg.lambda_ * norm / st["tau"] * 0.35).detach()
        elif not st["last"]:
            difference = (2 * y - 1) - resizer_apply(x0, 1.0 / cfg.sf)
            norm = torch.linalg.norm(difference)
            norm_grad = torch.autograd.grad(outputs=norm, inputs=x)[0]
            if trace is not None:
                trace.append(("norm_grad", t_i, norm_grad.clone()))
            x = (xt - norm_grad * 1.).detach()
    return x.detach() / 2 + 0.5


def psnr_batch(a, b, max_pixel=2.0, eps=1e-10):
    """utils_image.py:601-610."""
    mse = torch.mean((a - b) ** 2, dim=(1, 2, 3))
    v = torch.where(mse == 0, torch.full_like(mse, float("inf")), 20 * torch.log10(max_pixel / torch.sqrt(mse + eps)))
    v = torch.where(torch.isnan(v), torch.zeros_like(v), v)
    return float(torch.mean(v))


def restore(sd, hp, cfg: LoopConfig, y, k=None, mask=None, noise_fn: Callable = None, y_label=None,
            trace: Optional[list] = None, denoiser: Optional[Callable] = None, exact_prox: bool = False):
    """One batch of main_ddpir.py:259-470 (generate_mode DiffPIR / repaint / vanilla, pred_xstart, iter_num_U=1).

    y [B,3,h,w] in [0,1]; k [B,1,kh,kw] (deblur/sr-blur); mask [B,3,H,W] float {0,1} (inpaint).
    noise_fn(like) -> N(0,1) tensor; called in the reference's draw order (SURVEY 8 a-R):
    init, then per step: [repaint mix], p_sample, n1 (eta term), n2 (zeta term).
    `denoiser(x, t_i) -> x0` overrides the UNet (used to test the loop without a network).
    exact_prox: the FFT data-fidelity step (and its pre-calculated spectra) in float64 -- NOT the reference's arithmetic; the
    yardstick for the conditioning-aware parity bounds (tests/gpu_common.py::fft_prox_parity).
    Returns x_0 in [0,1] (un-clamped, main_ddpir.py:470)."""
    dt, steps = step_tables(cfg)
    dtab = DiffusionTables(cfg.T)
    y = loader_strides(y.float())                   # the memory layout the reference's loader gives (see loader_strides)
    if cfg.task == "inpaint":
        mask = loader_strides(mask.float())
    t_start = cfg.t_start(dt)
    x = init_x(cfg, y, mask, dt, noise_fn(torch.empty(y.shape[0], 3, y.shape[2] * cfg.sf, y.shape[3] * cfg.sf)), t_start)
    pre = None
    if cfg.task in ("sr", "deblur"):
        pre = pre_calculate(y.double(), k.double(), cfg.sf) if exact_prox else pre_calculate(y, k.float(), cfg.sf)
    if cfg.generate_mode != "DiffPIR" and cfg.task != "inpaint":
        raise ValueError("repaint / vanilla: inpainting only (main_ddpir.py:448 re-noises only for inpainting or DiffPIR)")
    for st in steps:
        t_i = st["t_i"]
        if t_i > t_start:                           # main_ddpir.py:346-347: nothing runs (and nothing is drawn) above t_start
            continue
        if cfg.generate_mode == "repaint":          # main_ddpir.py:355-358
            x = (dt.sqrt_ac[t_i] * (2 * y - 1) + dt.sqrt_1m_ac[t_i] * noise_fn(x)) * mask + (1 - mask) * x
        if denoiser is not None:
            x0 = denoiser(x, t_i)
            noise_fn(x)
        else:
            x0 = model_fn_xstart(sd, hp, x, st["curr_sigma"] * 255, dt, dtab, noise_fn, y_label)
        if trace is not None:
            trace.append(("x0", t_i, x0.clone()))
        if not st["last"]:
            tau = st["tau"].repeat(1, 1, 1, 1)
            if cfg.generate_mode != "DiffPIR":
                pass                                # main_ddpir.py:385: the data-fidelity step is DiffPIR-only
            elif not cfg.sub_1_analytic:
                # first-order solver (main_ddpir.py:420-430): x0 <- x0 - d||m - A(x0)|| / dx0 * ||.|| / rho, m = 2y - 1 (task sr)
                if cfg.task != "sr":
                    raise ValueError("the first-order data step is runnable in the reference for task 'sr' only")
                x0 = x0.detach().requires_grad_()
                norm = torch.linalg.norm((2 * y - 1) - resizer_apply(x0, 1.0 / cfg.sf))
                norm_grad = torch.autograd.grad(outputs=norm, inputs=x0)[0]
                x0 = (x0 - norm_grad * norm / st["tau"]).detach()
            elif cfg.task == "inpaint":
                x0 = prox_mask(x0, y, mask, tau, cfg.guidance_scale)
            elif cfg.task == "deblur" or cfg.sr_mode == "blur":
                x0 = prox_fft(x0, pre, tau, cfg.sf, cfg.guidance_scale, exact=exact_prox)
            else:
                x0 = prox_ibp(x0, y, st["tau"], cfg.sf, cfg.gamma, cfg.inIter)
            n1 = noise_fn(x)
            n2 = noise_fn(x)
            x = renoise(x, x0, dt, t_i, st["t_im1"], cfg.eta, cfg.zeta, n1, n2)
            if trace is not None:
                trace.append(("x", st["t_im1"], x.clone()))
    return x / 2 + 0.5
