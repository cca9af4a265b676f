"""TEST INFRASTRUCTURE / design prototype (numpy, CPU): the sf > 1 data-fidelity step of utils_sisr.py:65-75 evaluated on
HALF spectra only (rfft2 storage, W/2+1 columns) -- the index algebra a column-strip HIP kernel needs to extend csrc/fft2.hip's
half-spectrum path from sf = 1 to sf = 2, 3, 4.  Not used by the product path.

Every quantity the closed form touches is the spectrum of a REAL image, hence Hermitian: S[u, v] = conj(S[(H-u)%H, (W-v)%W]).
  FR   = FBFy + F(alpha x)                                   stored for v <= W/2
  x1   = FB * FR                                             stored for v <= W/2
  FBR[p, q]  = mean_{a,b < sf} x1[p + a Hs, q + b Ws]        (Hs = H/sf, Ws = W/sf)   <- utils_sisr.splits + mean
  invW[p, q] = mean_{a,b} F2B[p + a Hs, q + b Ws]
  FX   = (FR - conj(FB) * tile(FBR / (invW + alpha))) / alpha
An alias column q + b Ws beyond W/2 is read through the symmetry from column W - (q + b Ws) and row (H - u) % H, conjugated.
FBR is Hermitian on the (Hs, Ws) grid, so the fold is only evaluated for q <= Ws/2; a stored column v maps to q = v % Ws and, when
q > Ws/2, to the conjugate of entry ((Hs - p) % Hs, Ws - q).  For sf = 4, W = 256 the stored columns that meet in one fold are
{q, 64 - q, 64 + q, 128 - q}: four 16-byte-aligned strips per workgroup, and the row aliases u + a Hs of a thread that holds rows
t + 16 k2 of a 256-point column FFT stay in that thread (Hs = 64 = 16 * 4); only the row mirror (H - u) crosses threads."""
import numpy as np


def full_from_half(S, W):
    """[..., H, W/2+1] Hermitian half -> full [..., H, W] (reference for the tests only)."""
    H = S.shape[-2]
    out = np.empty(S.shape[:-1] + (W,), S.dtype)
    out[..., : W // 2 + 1] = S
    v = np.arange(W // 2 + 1, W)
    u = (-np.arange(H)) % H
    out[..., v] = np.conj(S[..., u, :][..., W - v])
    return out


def herm(S, u, v, H, W):
    """Value of the full Hermitian spectrum at (u, v) from its stored half."""
    u = np.asarray(u) % H
    v = np.asarray(v) % W
    m = v > W // 2
    uu = np.where(m, (H - u) % H, u)
    vv = np.where(m, W - v, v)
    val = S[..., uu, vv]
    return np.where(m, np.conj(val), val)


def data_solution_half(x, FBh, F2Bh, FBFyh, alpha, sf):
    """x [B,3,H,W] real; FBh [B,1,H,W/2+1] complex, F2Bh real, FBFyh [B,3,H,W/2+1] complex -- all HALF spectra."""
    H, W = x.shape[-2:]
    Hs, Ws = H // sf, W // sf
    WP = W // 2 + 1
    FR = FBFyh + np.fft.rfft2(alpha * x, axes=(-2, -1))
    x1 = FBh * FR
    # fold: only q <= Ws/2 is evaluated
    p = np.arange(Hs)[:, None]
    q = np.arange(Ws // 2 + 1)[None, :]
    FBR = np.zeros(x1.shape[:-2] + (Hs, Ws // 2 + 1), x1.dtype)
    invW = np.zeros(F2Bh.shape[:-2] + (Hs, Ws // 2 + 1), F2Bh.dtype)
    for a in range(sf):
        for b in range(sf):
            FBR += _herm(x1, p + a * Hs, q + b * Ws, H, W)
            invW += _herm(F2Bh, p + a * Hs, q + b * Ws, H, W).real
    FBR /= sf * sf
    invW /= sf * sf
    R = FBR / (invW + alpha)                                   # [B,3,Hs,Ws/2+1], Hermitian on the (Hs, Ws) grid
    # un-fold onto the stored columns v <= W/2
    u = np.arange(H)[:, None]
    v = np.arange(WP)[None, :]
    Rt = _herm(R, u % Hs, v % Ws, Hs, Ws)
    FX = (FR - np.conj(FBh) * Rt) / alpha
    return np.fft.irfft2(FX, s=(H, W), axes=(-2, -1))


def _herm(S, u, v, H, W):
    u = np.asarray(u) % H
    v = np.asarray(v) % W
    u, v = np.broadcast_arrays(u, v)
    m = v > W // 2
    uu = np.where(m, (H - u) % H, u)
    vv = np.where(m, W - v, v)
    val = S[..., uu, vv]
    return np.where(m, np.conj(val), val)
