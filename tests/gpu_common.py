"""Helpers shared by the -m gpu tests: engine fixture plumbing and oracle-side model loading."""
import numpy as np
import torch

from diffpir_amd import _lib, script_util
from oracle import unet_oracle as uo


def make_model(engine, hp: uo.UNetHP, seed=0, precision=None):
    """Build the engine model through the reference-shaped factory and load the oracle's synthetic weights."""
    if precision is not None:
        engine.set_precision(precision)
    model = script_util.create_model(
        image_size=hp.image_size, num_channels=hp.model_channels, num_res_blocks=hp.num_res_blocks,
        channel_mult=",".join(str(c) for c in hp.channel_mult) if hp.channel_mult else "",
        learn_sigma=True, class_cond=hp.class_cond, attention_resolutions=hp.attention_resolutions,
        num_heads=4, num_head_channels=hp.num_head_channels, num_heads_upsample=-1, use_scale_shift_norm=True,
        dropout=0.1, resblock_updown=True, use_fp16=False, use_new_attention_order=False, engine=engine,
        num_classes=hp.num_classes)
    sd = uo.synth_state_dict(hp, seed)
    model.load_state_dict({k: v.numpy() for k, v in sd.items()})
    return model, sd


def seeded_noise_fn_torch(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda like: torch.randn(like.shape, generator=g, dtype=torch.float32)


def seeded_noise_fn_np(seed):
    """Same stream as seeded_noise_fn_torch, returning numpy (for the engine's host-noise path)."""
    g = torch.Generator().manual_seed(seed)
    return lambda shape: torch.randn(tuple(shape), generator=g, dtype=torch.float32).numpy()


def rel_err(a, b):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


_ORACLE_CACHE = {}


def oracle_pair(key, sd, hp, ocfg, y, k, seed, y_label=None):
    """(reference-arithmetic result, exact-prox result) of the oracle loop, cached per test key (both arithmetic modes of
    the engine are compared with the same pair)."""
    from oracle import diffpir_oracle as do
    if key not in _ORACLE_CACHE:
        ty, tk = torch.from_numpy(y), torch.from_numpy(k)
        ref = do.restore(sd, hp, ocfg, ty, k=tk, noise_fn=seeded_noise_fn_torch(seed), y_label=y_label).numpy()
        exact = do.restore(sd, hp, ocfg, ty, k=tk, noise_fn=seeded_noise_fn_torch(seed), exact_prox=True, y_label=y_label).numpy()
        _ORACLE_CACHE[key] = (ref, exact)
    return _ORACLE_CACHE[key]


def fft_prox_parity(out, ref, gt, label, exact=None, floor=None, nfe=None, floor_dpsnr=None):
    """Parity gate for loops through the FFT data-fidelity step.

    The reference's closed form (utils_sisr.py:65-75) divides a near-cancelling difference by alpha = tau (7e-7 at the first
    steps of config 2): its OWN fp32 evaluation differs from exact arithmetic by ~4e-2 per call there, and after a few NFE the
    final image still carries that rounding noise (it is contracted away by ~100 NFE).  Two fp32 implementations therefore agree
    only to the reference's own rounding-noise level, which is measured, not guessed: `exact` is the oracle loop with the prox in
    float64 (same UNet, same noise), floor = (max, rms) of |ref - exact|.  Asserted:
      * |dPSNR| <= 1e-3 dB (north-star tolerance), or the reference's own PSNR shift against exact arithmetic where that is
        larger (2-4 NFE cases, where the final image still carries the first steps' rounding noise)
      * engine vs reference   : rms <= floor rms, max <= 1.5 x floor max  (closer to the reference than the reference is to exact)
      * engine vs exact       : rms <= 2 x floor rms   (the triangle-inequality consequence of the line above; the rounding noise of
        any fp32 evaluation lies along the same few ill-conditioned spectral modes, so the two deviations can add coherently --
        measured on ImageNet-256 sr x4: reference-vs-exact 2.0e-3, engine-vs-reference 1.2e-3, engine-vs-exact 2.9e-3)
    Runs of nfe >= 8 (the first steps' rounding noise has been contracted) get NO allowance on the north-star quantity:
    |dPSNR| <= 1e-3 dB flat.  The 1.5 x on the MAXIMUM stays for every length: it compares the largest of 2e5..8e5 samples of two
    different rounding-noise fields, and measured on config 3 at 20 NFE the engine's rms is 0.60 x the reference's own while its
    maximum is 1.35 x (5.9e-3 vs 4.4e-3) -- in BOTH arithmetic modes, i.e. a property of the sample maximum, not of the kernels."""
    from diffpir_amd import restore
    if floor is None:
        d = ref - exact
        floor = (float(np.abs(d).max()), float(np.sqrt(np.mean(d * d))))
    e = out - ref
    emax, erms = float(np.abs(e).max()), float(np.sqrt(np.mean(e * e)))
    gap = abs(restore.psnr_batch(out * 2 - 1, gt * 2 - 1) - restore.psnr_batch(ref * 2 - 1, gt * 2 - 1))
    msg = (f"{label}: engine-vs-reference max {emax:.3e} rms {erms:.3e} | reference-vs-exact (its own fp32 noise) max {floor[0]:.3e} "
           f"rms {floor[1]:.3e} | |dPSNR| {gap:.2e} dB")
    gap_floor = 0.0 if floor_dpsnr is None else float(floor_dpsnr)
    if exact is not None:
        x = out - exact
        xrms = float(np.sqrt(np.mean(x * x)))
        gap_floor = abs(restore.psnr_batch(ref * 2 - 1, gt * 2 - 1) - restore.psnr_batch(exact * 2 - 1, gt * 2 - 1))
        msg += f" | engine-vs-exact rms {xrms:.3e} | reference's own |dPSNR| vs exact {gap_floor:.2e} dB"
        assert xrms <= 2.0 * floor[1] + 1e-6, msg
    print(msg)
    flat_bar = nfe is not None and nfe >= 8
    assert gap <= (1e-3 if flat_bar else max(1e-3, gap_floor)), msg
    assert erms <= floor[1] + 1e-6 and emax <= 1.5 * floor[0] + 1e-5, msg
    return emax, erms, gap
