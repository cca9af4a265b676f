"""tests/golden/shim_trace.npz: the CALL TRACE of the reference's own loop body (build container only):

    python -m oracle.gen_golden_shim_trace

TEST INFRASTRUCTURE ONLY.  INTEGRATION.md section A claims that /root/reference/main_ddpir.py:341-470 runs unchanged when its five plug names --
`utils_model.model_fn`, `utils_model.grad_and_value`, `sr.pre_calculate`, `sr.data_solution`, `Resizer` -- are bound to diffpir_amd's mirrors.  The GPU
box has no reference tree, so that cannot be run there in one process.  What CAN travel is data: this script executes the reference's unmodified
`test_rho` (oracle/ref_exec.py: taken out of the source file with `ast`) with recording wrappers around the REAL five plugs and stores, per call and in
call order, the arguments the loop body handed over and what the plug returned (plus the randn_like draws the reference consumed in between).  The
-m gpu test tests/test_gpu_shim_trace.py replays every recorded call through the engine's mirror with the recorded arguments and compares the returns,
then checks the last recorded x_0 against dpir_run_loop on the same inputs.  No line of the reference is stored: arrays, scalars and plug names only.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from . import ref_exec, ref_import
from . import unet_oracle as uo
from . import diffpir_oracle as do
from .gen_golden import OUT, smooth_images, gaussian_kernel


class Recorder:
    def __init__(self):
        self.calls, self.arrays = [], {}

    def put(self, key, t):
        a = t.detach().cpu().numpy().copy() if torch.is_tensor(t) else np.array(t)      # a COPY: the loop body goes on to modify some of these tensors in place
        self.arrays[key] = a
        return key

    def add(self, fn, scalars, **tensors):
        i = len(self.calls)
        rec = dict(fn=fn, **scalars)
        for name, t in tensors.items():
            if t is not None:
                rec[name] = self.put(f"{i:03d}_{fn}_{name}", t)
        self.calls.append(rec)
        return rec


def record_case(case, cfg, hp, y, k=None, mask=None, seed=0, frozen=True):
    md = ref_exec.main_module()
    sd = uo.synth_state_dict(hp, 0)
    model, diffusion = ref_exec.build_unet(hp, sd, frozen=frozen)
    R = Recorder()
    g = torch.Generator().manual_seed(seed)

    def noise_fn(like):                                       # every torch.randn_like of the run (p_sample's and the loop's), in order
        n = torch.randn(like.shape, generator=g, dtype=torch.float32)
        R.add("randn_like", dict(shape=[int(v) for v in like.shape]))      # the draw itself is torch.Generator(seed)'s stream: reproducible, not stored
        return n

    def model_fn(x, noise_level, model_out_type="pred_xstart", ddim_sample=False, **kw):
        out = md.utils_model.model_fn(x, noise_level=noise_level, model_out_type=model_out_type, ddim_sample=ddim_sample, **kw)
        sc = dict(noise_level=float(noise_level), model_out_type=model_out_type, ddim_sample=bool(ddim_sample))
        if isinstance(out, tuple):
            R.add("model_fn", sc, x=x, out0=out[0], out1=out[1])
        else:
            R.add("model_fn", sc, x=x, out0=out)
        return out

    def grad_and_value(operator, x, x_hat, measurement):
        gnorm, norm = md.utils_model.grad_and_value(operator=operator, x=x, x_hat=x_hat, measurement=measurement)
        R.add("grad_and_value", dict(x_is_x_hat=bool(x is x_hat), sf=int(round(1.0 / operator.scale_factor[-1])) if hasattr(operator, "scale_factor") else 0),
              x=x, x_hat=x_hat, measurement=measurement, norm_grad=gnorm, norm=norm)
        return gnorm, norm

    def pre_calculate(yy, kk, sf):
        out = md.sr.pre_calculate(yy, kk, sf)
        R.add("pre_calculate", dict(sf=int(sf)), y=yy, k=kk, FB=out[0], F2B=out[2], FBFy=out[3])
        return out

    def data_solution(x, FB, FBC, F2B, FBFy, alpha, sf):
        out = md.sr.data_solution(x, FB, FBC, F2B, FBFy, alpha, sf)
        R.add("data_solution", dict(sf=int(sf), alpha=float(alpha.reshape(-1)[0])), x=x, out=out)
        return out

    class RecResizer(md.Resizer):
        def __init__(self, in_shape, scale_factor=None, *a, **kw):
            super().__init__(in_shape, scale_factor, *a, **kw)
            self._sf = int(round(1.0 / scale_factor))
            R.add("Resizer", dict(in_shape=[int(v) for v in in_shape], sf=self._sf))

        def forward(self, x):
            out = super().forward(x)
            if not x.requires_grad:                                  # calls inside grad_and_value are part of that record
                R.add("Resizer_forward", dict(sf=self._sf), x=x, out=out)
            return out

    over = dict(utils_model=ref_exec._Proxy(md.utils_model, model_fn=model_fn, grad_and_value=grad_and_value),
                sr=ref_exec._Proxy(md.sr, pre_calculate=pre_calculate, data_solution=data_solution), Resizer=RecResizer)
    x0 = ref_exec.restore_ref(model, diffusion, cfg, torch.from_numpy(y), k=None if k is None else torch.from_numpy(k),
                              mask=None if mask is None else torch.from_numpy(mask), noise_fn=noise_fn, ns_over=over)
    R.put("x0", x0)
    meta = dict(calls=R.calls, cfg=dict(task=cfg.task, iter_num=int(cfg.iter_num), noise_level_img=float(cfg.noise_level_img), lambda_=float(cfg.lambda_),
                                        zeta=float(cfg.zeta), sf=int(cfg.sf), sr_mode=cfg.sr_mode, generate_mode=cfg.generate_mode, seed=int(seed)))
    out = {f"{case}/{kk}": v for kk, v in R.arrays.items()}
    out[f"{case}/y"] = y
    if k is not None:
        out[f"{case}/k"] = k
    if mask is not None:
        out[f"{case}/mask"] = mask
    out[f"{case}/meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    print(case, ":", len(R.calls), "calls:", {f: sum(1 for c in R.calls if c["fn"] == f) for f in sorted({c["fn"] for c in R.calls})})
    return out


def main():
    ns = ref_import.load()
    torch.set_num_threads(8)
    hp = uo.tiny_hp()
    import scipy.io
    from scipy import ndimage
    kb = scipy.io.loadmat(os.path.join(ref_import.REF_ROOT, "kernels", "kernels_bicubicx234.mat"))["kernels"]
    k_bic4 = kb[0, 2].astype(np.float32)
    out = {}
    # deblur (FFT prox, sf = 1)
    gt = smooth_images(2, 32, 32, 9)
    kg = gaussian_kernel(9, 1.2)
    yb = np.stack([ndimage.convolve(gt[b].transpose(1, 2, 0), kg[:, :, None], mode="wrap").transpose(2, 0, 1) for b in range(2)])
    yb = (yb + np.random.default_rng(2).normal(0, 0.05, yb.shape)).astype(np.float32)
    kt = np.stack([kg, kg])[:, None].astype(np.float32)
    out.update(record_case("deblur", do.LoopConfig("deblur", 5, 12.75 / 255, 7.0, 0.3), hp, yb, k=kt, seed=52))
    # inpaint (inline masked prox: model_fn is the only plug)
    m = np.ones((2, 3, 32, 32), np.float32)
    m[0, :, 8:24, 6:22] = 0
    m[1, :, 4:20, 12:28] = 0
    out.update(record_case("inpaint", do.LoopConfig("inpaint", 5, 0.0, 1.0, 1.0), hp, (gt * m).astype(np.float32), mask=m, seed=54))
    # sr x4: blur mode (FFT prox with the bicubic PSF) and DPS_y0 (Resizer + grad_and_value through the denoiser)
    gt64 = smooth_images(2, 64, 64, 10)
    ylr = ns.utils_resizer.Resizer((2, 3, 64, 64), 0.25)(torch.from_numpy(gt64))
    ylr = (ylr + torch.from_numpy(np.random.default_rng(4).normal(0, 0.05, ylr.shape).astype(np.float32))).float().numpy()
    k4 = np.stack([k_bic4, k_bic4])[:, None].astype(np.float32)
    out.update(record_case("sr_blur", do.LoopConfig("sr", 4, 12.75 / 255, 6.0, 0.25, sf=4), hp, ylr, k=k4, seed=55))
    out.update(record_case("dps_y0", do.LoopConfig("sr", 4, 12.75 / 255, 6.0, 0.25, sf=4, sr_mode="cubic", generate_mode="DPS_y0"), hp, ylr, k=k4, seed=56,
                           frozen=False))
    path = os.path.join(OUT, "shim_trace.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
