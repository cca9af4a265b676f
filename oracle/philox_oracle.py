"""Philox4x32-10 + Box-Muller in numpy: an INDEPENDENT statement of the engine's device noise source (diffpir_amd/csrc/philox.h), so that the
perf-mode replacement of torch.randn_like is pinned by construction and not only by its statistics.

TEST INFRASTRUCTURE ONLY.  Philox4x32-10 is Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3" (SC'11), the generator behind
torch.cuda / cuRAND; its multipliers, Weyl constants and round count are public.  `philox4x32_10` is checked against the Random123 known-answer
vectors in tests/test_oracle_golden.py.  The reference itself draws from torch's CPU Mersenne Twister; no stream of the device generator can equal
that one -- which is why parity runs feed host noise -- but this file fixes WHICH numbers a given (seed, image, draw, element) produces on the device.

Keying (philox.h): counter = (j, image, stream, mix of the high halves), key = seed; one call gives the 4 normals of elements [4 j, 4 j + 4) of image
`image` in draw `stream`; inside dpir_run_loop stream = draw kind + 4 * step (kinds: 1 eta, 2 zeta, 3 repaint mix), 0 for the initial x_T draw."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: uint32 [..., 2] (broadcastable) -> uint32 [..., 4]"""
    c = [np.asarray(ctr[..., i], np.uint32).copy() for i in range(4)]
    k0, k1 = np.asarray(key[..., 0], np.uint32).copy(), np.asarray(key[..., 1], np.uint32).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c[0].astype(np.uint64) * M0
            p1 = c[2].astype(np.uint64) * M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def randn(seed, stream_id, image_offset, B, per_image):
    """What dpir_randn(e, out, seed, stream_id, image_offset, B, C, H, W) writes: float32 [B, per_image]."""
    q = (per_image + 3) // 4
    j = np.arange(q, dtype=np.uint64)[None, :]
    img = (np.uint64(image_offset) + np.arange(B, dtype=np.uint64))[:, None]
    s = np.uint64(stream_id)
    ctr = np.empty((B, q, 4), np.uint32)
    ctr[..., 0] = (j & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[..., 1] = (img & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[..., 2] = np.uint32(int(s) & 0xFFFFFFFF)
    # philox.h: (img >> 32) ^ (stream_id >> 32) << 16 ^ (j >> 32)   with C precedence: << binds tighter than ^
    ctr[..., 3] = (((img >> np.uint64(32)) ^ ((s >> np.uint64(32)) << np.uint64(16)) ^ (j >> np.uint64(32))) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    key = np.array([int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF], np.uint32)
    r = philox4x32_10(ctr, key[None, None, :])
    u = ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    u64 = u.astype(np.float64)
    r0 = np.sqrt(-2.0 * np.log(u64[..., 0])); r1 = np.sqrt(-2.0 * np.log(u64[..., 2]))
    z = np.stack([r0 * np.cos(2 * np.pi * u64[..., 1]), r0 * np.sin(2 * np.pi * u64[..., 1]),
                  r1 * np.cos(2 * np.pi * u64[..., 3]), r1 * np.sin(2 * np.pi * u64[..., 3])], axis=-1)
    return z.reshape(B, q * 4)[:, :per_image].astype(np.float32)
