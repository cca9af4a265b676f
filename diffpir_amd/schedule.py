"""Host-side schedule tables and per-step scalars of the DiffPIR loop (numpy only).

Mirrors main_ddpir.py:184-190 (float32 driver tables), :274-286 (sigmas / rhos), :327-344 + :451
(timestep sequence), :454-456 (re-noise coefficients) and the float64 tables of
guided_diffusion/gaussian_diffusion.py:27-35,133-151 used by eps -> x0 (:328-333).
Everything here is a few thousand scalar operations per batch; it is evaluated once, before the
loop, instead of 12 host<->device crossings per step in the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

from . import _lib


@dataclass
class DriverTables:
    """main_ddpir.py:184-190.  np.cumprod on a float32 array is a sequential float32 product,
    which is what the reference gets from np.cumprod(torch_tensor_float32)."""
    T: int
    betas: np.ndarray
    alphas_cumprod: np.ndarray
    sqrt_ac: np.ndarray
    sqrt_1m_ac: np.ndarray
    reduced: np.ndarray

    @staticmethod
    def make(beta_start=0.0001, beta_end=0.02, T=1000) -> "DriverTables":
        # The reference evaluates these 1000-entry tables with torch CPU ops (main_ddpir.py:184-190);
        # torch.sqrt(float32) is not correctly rounded on every entry, so the same ops are used here
        # (host-side scalar prep) to obtain bit-identical tables.
        import torch
        betas = torch.from_numpy(np.linspace(beta_start, beta_end, T, dtype=np.float32))
        alphas = 1.0 - betas
        ac = torch.as_tensor(np.cumprod(alphas.numpy(), axis=0))
        s_ac = torch.sqrt(ac)
        s_1m = torch.sqrt(1. - ac)
        red = torch.div(s_1m, s_ac)
        return DriverTables(T, betas.numpy(), ac.numpy(), s_ac.numpy(), s_1m.numpy(), red.numpy())


@dataclass
class DiffusionTables:
    """GaussianDiffusion.__init__ float64 tables (gaussian_diffusion.py:133-151), linear schedule :27-35."""
    sqrt_recip_ac: np.ndarray
    sqrt_recipm1_ac: np.ndarray
    # posterior q(x_{t-1} | x_t, x_0) and the learned-range variance bounds (gaussian_diffusion.py:153-167, 268-276): p_sample of
    # the DPS modes (model_out_type 'pred_x_prev_and_start')
    posterior_mean_coef1: np.ndarray = None
    posterior_mean_coef2: np.ndarray = None
    posterior_log_variance_clipped: np.ndarray = None
    log_betas: np.ndarray = None
    alphas_cumprod_prev: np.ndarray = None

    @staticmethod
    def make(T=1000) -> "DiffusionTables":
        scale = 1000 / T
        betas = np.linspace(scale * 0.0001, scale * 0.02, T, dtype=np.float64)
        ac = np.cumprod(1.0 - betas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        return DiffusionTables(np.sqrt(1.0 / ac), np.sqrt(1.0 / ac - 1),
                               betas * np.sqrt(ac_prev) / (1.0 - ac), (1.0 - ac_prev) * np.sqrt(1.0 - betas) / (1.0 - ac),
                               np.log(np.append(post_var[1], post_var[1:])), np.log(betas), ac_prev)

    def ddim_coef(self, t: int):
        """ddim_sample(eta = 0) (gaussian_diffusion.py:568-580): (sqrt(alpha_bar_prev), sqrt(1 - alpha_bar_prev - sigma^2)) with sigma = 0,
        evaluated in float32 as the reference does on its _extract_into_tensor(...).float() tensors."""
        abp = np.float32(self.alphas_cumprod_prev[t])
        return np.sqrt(abp, dtype=np.float32), np.sqrt(np.float32(np.float32(1.0) - abp) - np.float32(0.0), dtype=np.float32)

    def dps_coef(self, t: int):
        """(posterior_mean_coef1, posterior_mean_coef2, min_log, max_log) at t as float32 (_extract_into_tensor(...).float())."""
        return (np.float32(self.posterior_mean_coef1[t]), np.float32(self.posterior_mean_coef2[t]),
                np.float32(self.posterior_log_variance_clipped[t]), np.float32(self.log_betas[t]))

    def c1c2(self, t: int):
        return np.float32(self.sqrt_recip_ac[t]), np.float32(self.sqrt_recipm1_ac[t])


def find_nearest(array, value) -> int:
    """utils/utils_model.py:202-205."""
    array = np.asarray(array)
    return int(np.abs(array - value).argmin())


def make_seq(T: int, iter_num: int, skip_type: str = "quad") -> List[int]:
    """main_ddpir.py:327-335."""
    if skip_type == "uniform":
        skip = T // iter_num
        seq = [i * skip for i in range(iter_num)]
        if skip > 1:
            seq.append(T - 1)
        return seq
    if skip_type != "quad":
        raise ValueError(f"unknown skip_type {skip_type}")
    s = np.sqrt(np.linspace(0, T ** 2, iter_num))
    seq = [int(v) for v in list(s)]
    seq[-1] = seq[-1] - 1
    return seq


def build_steps(*, iter_num: int, sigma: float, lambda_: float, zeta: float, eta: float = 0.0,
                skip_type: str = "quad", T: int = 1000, beta_start=0.0001, beta_end=0.02,
                t_start: int = None, generate_mode: str = "DiffPIR", model_output_type: str = "pred_xstart"):
    """Returns (DriverTables, list of python dicts, ctypes Step array) for one (lambda, zeta) setting.

    sigma = max(0.001, noise_level_img/255) (main_ddpir.py:141).  generate_mode / model_output_type select the sigma_k table the
    rhos are built from (main_ddpir.py:279-283): sigma_bar_t only for (pred_xstart, DiffPIR), sqrt(beta_t / alpha_t) otherwise --
    which is what generate_mode DPS_yt divides its step by (:443)."""
    dt = DriverTables.make(beta_start, beta_end, T)
    dtab = DiffusionTables.make(T)
    if t_start is None:
        t_start = T - 1
    seq = make_seq(T, iter_num, skip_type)
    # sigmas[i] = reduced[T-1-i]; rhos[i] = lambda*sigma^2/sigma_k[i]^2, sigma_k = s1m/sa, evaluated with
    # torch 0-dim float32 tensors like main_ddpir.py:277-286 (python-float / tensor = reciprocal * scalar)
    import torch
    t_s1m, t_sa = torch.from_numpy(dt.sqrt_1m_ac), torch.from_numpy(dt.sqrt_ac)
    if model_output_type == "pred_xstart" and generate_mode == "DiffPIR":
        sigma_ks = t_s1m / t_sa                           # main_ddpir.py:279-280
    else:
        t_betas = torch.from_numpy(dt.betas)
        sigma_ks = torch.sqrt(t_betas / (1.0 - t_betas))  # main_ddpir.py:282-283 (alphas = 1.0 - betas, :186)
    rhos = (lambda_ * (sigma ** 2) / (sigma_ks ** 2)).float().numpy()
    t_list = [find_nearest(dt.reduced, dt.reduced[T - 1 - s]) for s in seq]
    steps = []
    for i, t_i in enumerate(t_list):
        if t_i > t_start:
            continue                                      # main_ddpir.py:346-347
        last = seq[i] == seq[-1]
        c1, c2 = dtab.c1c2(t_i)
        st = dict(t=t_i, last=int(last), c1=float(c1), c2=float(c2), tau=float(rhos[t_i]),
                  sa_t=float(dt.sqrt_ac[t_i]), s1m_t=float(dt.sqrt_1m_ac[t_i]),
                  sa_p=0.0, k1=0.0, q=0.0, es=0.0, k2=0.0, t_im1=None)
        if not last:
            t_p = t_list[i + 1]
            # main_ddpir.py:454-456 with 0-dim float32 tensors
            s1m_p, s1m_t = t_s1m[t_p], t_s1m[t_i]
            es = eta * s1m_p / s1m_t * torch.sqrt(torch.from_numpy(dt.betas)[t_i])
            q = torch.sqrt(s1m_p ** 2 - es ** 2)
            k2 = np.sqrt(zeta) * s1m_p
            st.update(sa_p=float(dt.sqrt_ac[t_p]), k1=float(np.float32(np.sqrt(1 - zeta))), q=float(q), es=float(es),
                      k2=float(k2), t_im1=t_p)
        steps.append(st)
    arr = (_lib.Step * len(steps))()
    for i, st in enumerate(steps):
        for f, _ in _lib.Step._fields_:
            setattr(arr[i], f, st[f])
    return dt, steps, arr
