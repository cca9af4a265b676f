// act_split: the activation pre-pass of the operand-split f16 convolution path (conv6.hip).
//
// One HBM-bound elementwise kernel applies everything the reference runs between two convolutions --
// GroupNorm affine (nn.py:17-19), FiLM scale/shift (unet.py:250-251), SiLU (unet.py:184,208), 2x2 average pooling or
// nearest x2 up-sampling (unet.py:107,136) and the channel concat (unet.py:660) -- ONCE per element, splits the
// result into f16 hi/lo halves (x = hi + lo) and stores it in the blocked layout [n][C/8][H][W][8] that conv6's
// LDS-DMA copies straight into MFMA B-operand order.  Bytes per element: 4 read + 4 written.
#include "common.h"

namespace dpir {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float silu_a(float v) {
    float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// f16 operand range guard.  The split clamps to +-65000 so that hi stays finite; a value that needed the clamp (or a NaN)
// makes the result wrong, so it is COUNTED: one atomic per wave that saw one (none in the normal case), read back by
// dpir_sync / dpir_d2h, which then fail with DPIR_ERR_RANGE instead of returning a silently saturated image.
__device__ __forceinline__ void range_report(bool bad, unsigned long long* ctr) {
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && ctr && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(ctr, (unsigned long long)__builtin_popcountll(m));
}

// grid: (ceil(H*W/256), B*C8); MODE 0 plain, 1 nearest-up (source is H/2 x W/2), 2 avg-pool (source is 2H x 2W)
template <int MODE>
__global__ __launch_bounds__(256) void act_split_kernel(CatSrc src, const float4* prm, int C, int C8, int H, int W,
                                                        _Float16* hi, _Float16* lo, unsigned long long* range_ctr) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y / C8, c8 = blockIdx.y - n * C8;
    const int HW = H * W;
    if (pix >= HW) return;
    bool bad = false;
    const int y = pix / W, x = pix - y * W;
    const int Hs = MODE == 1 ? H >> 1 : (MODE == 2 ? H * 2 : H);
    const int Ws = MODE == 1 ? W >> 1 : (MODE == 2 ? W * 2 : W);
    const int so = MODE == 0 ? pix : (MODE == 1 ? (y >> 1) * Ws + (x >> 1) : (2 * y) * Ws + 2 * x);
    const size_t HsWs = (size_t)Hs * Ws;
    half8 h8, l8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;          // uniform per workgroup
        float v = 0.f;
        if (c < C) {
            const float* plane = c < src.ca ? src.a + ((size_t)n * src.ca + c) * HsWs : src.b + ((size_t)n * src.cb + (c - src.ca)) * HsWs;
            float4 m = prm ? prm[(size_t)n * C + c] : make_float4(0.f, 1.f, 0.f, 0.f);
            if (MODE == 2) {
                float v0 = plane[so], v1 = plane[so + 1], v2 = plane[so + Ws], v3 = plane[so + Ws + 1];
                if (prm) {
                    v0 = (v0 - m.x) * m.y + m.z; v1 = (v1 - m.x) * m.y + m.z; v2 = (v2 - m.x) * m.y + m.z; v3 = (v3 - m.x) * m.y + m.z;
                    if (m.w != 0.f) { v0 = silu_a(v0); v1 = silu_a(v1); v2 = silu_a(v2); v3 = silu_a(v3); }
                }
                v = ((v0 + v1) + (v2 + v3)) * 0.25f;
            } else {
                v = plane[so];
                if (prm) {
                    v = (v - m.x) * m.y + m.z;
                    if (m.w != 0.f) v = silu_a(v);
                }
            }
        }
        bad |= !(fabsf(v) <= 65000.f);
        v = fminf(fmaxf(v, -65000.f), 65000.f);
        _Float16 hh = (_Float16)v;
        h8[j] = hh;
        l8[j] = (_Float16)(v - (float)hh);
    }
    const size_t o = (((size_t)n * C8 + c8) * HW + pix) * 8;
    *reinterpret_cast<half8*>(hi + o) = h8;
    if (lo) *reinterpret_cast<half8*>(lo + o) = l8;
    range_report(bad, range_ctr);
}

// plain-resolution fast path: 4 consecutive pixels per thread (float4 loads per channel, 64-byte stores per plane)
__global__ __launch_bounds__(256) void act_split4_kernel(CatSrc src, const float4* prm, int C, int C8, int HW, _Float16* hi, _Float16* lo,
                                                         unsigned long long* range_ctr) {
    const int p4 = blockIdx.x * 256 + threadIdx.x;          // group of 4 pixels
    const int n = blockIdx.y / C8, c8 = blockIdx.y - n * C8;
    const bool live = p4 * 4 < HW;
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C && live) {
            const float* plane = c < src.ca ? src.a + ((size_t)n * src.ca + c) * HW : src.b + ((size_t)n * src.cb + (c - src.ca)) * HW;
            v[j] = *reinterpret_cast<const float4*>(plane + (size_t)p4 * 4);
        }
    }
    if (prm) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c8 * 8 + j;
            if (c < C) {
                float4 m = prm[(size_t)n * C + c];
                v[j].x = (v[j].x - m.x) * m.y + m.z; v[j].y = (v[j].y - m.x) * m.y + m.z;
                v[j].z = (v[j].z - m.x) * m.y + m.z; v[j].w = (v[j].w - m.x) * m.y + m.z;
                if (m.w != 0.f) { v[j].x = silu_a(v[j].x); v[j].y = silu_a(v[j].y); v[j].z = silu_a(v[j].z); v[j].w = silu_a(v[j].w); }
            }
        }
    }
    // The thread's 4 pixels are 4 consecutive 16-byte entries per plane; written directly that is a 64-byte lane stride
    // (quarter-dense store instructions).  Transpose through LDS ([q][thread] -> entry order) so that every store
    // instruction writes 1 KiB of consecutive entries.
    constexpr int PS = 256 + 4;                                 // plane stride in entries: conflict-free reads
    __shared__ half8 sh_hi[4 * PS], sh_lo[4 * PS];
    bool bad = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        half8 h8, l8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = q == 0 ? v[j].x : (q == 1 ? v[j].y : (q == 2 ? v[j].z : v[j].w));
            bad |= !(fabsf(x) <= 65000.f);
            x = fminf(fmaxf(x, -65000.f), 65000.f);
            _Float16 hh = (_Float16)x;
            h8[j] = hh;
            l8[j] = (_Float16)(x - (float)hh);
        }
        sh_hi[q * PS + threadIdx.x] = h8;
        sh_lo[q * PS + threadIdx.x] = l8;
    }
    range_report(bad, range_ctr);
    __syncthreads();
    const size_t o = ((size_t)n * C8 + c8) * HW + (size_t)blockIdx.x * 1024;      // first entry of this workgroup
    const int rem = HW - blockIdx.x * 1024;                                        // valid entries (multiple of 4)
    half8* gh = reinterpret_cast<half8*>(hi) + o;
    half8* gl = reinterpret_cast<half8*>(lo) + o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = threadIdx.x + 256 * k;                     // entry = 4 * thread' + q'
        if (e < rem) {
            const int sl = (e & 3) * PS + (e >> 2);
            gh[e] = sh_hi[sl];
            if (lo) gl[e] = sh_lo[sl];
        }
    }
}

Status launch_act_split(hipStream_t s, CatSrc src, const float4* prm, int mode, int B, int H, int W, void* hi, void* lo,
                        unsigned long long* range_ctr) {
    const int C = src.ca + src.cb, C8 = 2 * ((C + 15) / 16);   // whole 16-channel K chunks (zero padded)
    dim3 grid((unsigned)((H * W + 255) / 256), (unsigned)(B * C8));
    _Float16* h = reinterpret_cast<_Float16*>(hi);
    _Float16* l = reinterpret_cast<_Float16*>(lo);
    if (mode == 0 && (H * W) % 4 == 0) {
        dim3 g4((unsigned)((H * W / 4 + 255) / 256), (unsigned)(B * C8));
        hipLaunchKernelGGL(act_split4_kernel, g4, dim3(256), 0, s, src, prm, C, C8, H * W, h, l, range_ctr);
    } else if (mode == 0) hipLaunchKernelGGL(act_split_kernel<0>, grid, dim3(256), 0, s, src, prm, C, C8, H, W, h, l, range_ctr);
    else if (mode == 1) hipLaunchKernelGGL(act_split_kernel<1>, grid, dim3(256), 0, s, src, prm, C, C8, H, W, h, l, range_ctr);
    else hipLaunchKernelGGL(act_split_kernel<2>, grid, dim3(256), 0, s, src, prm, C, C8, H, W, h, l, range_ctr);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
