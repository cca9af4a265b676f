"""The half-spectrum formulation of the sf > 1 data-fidelity step (oracle/half_spectrum_sr.py: the index algebra for extending the
register-FFT path of csrc/fft2.hip to super-resolution) against the oracle's full-spectrum restatement of utils_sisr.py:65-75,
in float64 where the two must agree to rounding."""
import numpy as np
import pytest
import torch

from oracle import diffpir_oracle as do, half_spectrum_sr as hs


@pytest.mark.parametrize("H,W,sf", [(64, 64, 2), (64, 64, 4), (48, 96, 3), (128, 256, 4), (32, 32, 1)])
def test_half_spectrum_fold_equals_full_spectrum_solution(H, W, sf):
    rng = np.random.default_rng(H + W + sf)
    B = 2
    k = rng.random((B, 1, 7, 9))
    k /= k.sum(axis=(2, 3), keepdims=True)                       # asymmetric, odd x odd PSF
    y = rng.random((B, 3, H // sf, W // sf))
    x = rng.random((B, 3, H, W))
    pre = do.pre_calculate(torch.from_numpy(y), torch.from_numpy(k), sf)
    FB, _, F2B, FBFy = [t.resolve_conj().numpy() for t in pre]
    WP = W // 2 + 1
    assert np.abs(hs.full_from_half(FBFy[..., :WP], W) - FBFy).max() < 1e-9 * np.abs(FBFy).max()      # the premise: Hermitian spectra
    for a in (1e-3, 0.3, 20.0):
        ref = do.data_solution(torch.from_numpy(x), *pre, torch.tensor(a, dtype=torch.float64), sf).numpy()
        out = hs.data_solution_half(x, FB[..., :WP], F2B[..., :WP], FBFy[..., :WP], a, sf)
        assert np.abs(out - ref).max() < 1e-10 / min(a, 1.0), (a, np.abs(out - ref).max())
