// Philox4x32-10 + Box-Muller: the device noise source of the loop (torch.randn_like's stand-in in performance mode).
// One call = 4 standard normals for elements [4j, 4j+4) of image `img`, draw `stream_id`; keyed by (seed, global image index,
// draw index), so results do not depend on how a batch is sharded over GPUs.  Shared by randn_kernel (elem.hip) and the fused
// inverse-row-FFT + re-noise kernel (fft2.hip), which must produce the same numbers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpir {

__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ float philox_u01(uint32_t v) { return ((float)(v >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t stream_id, uint64_t img, size_t j, float (&z)[4]) {
    uint32_t c0 = (uint32_t)j, c1 = (uint32_t)img, c2 = (uint32_t)stream_id, c3 = (uint32_t)((img >> 32) ^ (stream_id >> 32) << 16 ^ (uint64_t)(j >> 32));
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    float u0 = philox_u01(c0), u1 = philox_u01(c1), u2 = philox_u01(c2), u3 = philox_u01(c3);
    float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    z[0] = r0 * cospif(2.0f * u1); z[1] = r0 * sinpif(2.0f * u1);
    z[2] = r1 * cospif(2.0f * u3); z[3] = r1 * sinpif(2.0f * u3);
}

}  // namespace dpir
