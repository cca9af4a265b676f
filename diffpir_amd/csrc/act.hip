// act_split: the activation pre-pass of the operand-split f16 convolution path (conv6.hip).
//
// One HBM-bound elementwise kernel applies everything the reference runs between two convolutions --
// GroupNorm affine (nn.py:17-19), FiLM scale/shift (unet.py:250-251), SiLU (unet.py:184,208), 2x2 average pooling or
// nearest x2 up-sampling (unet.py:107,136) and the channel concat (unet.py:660) -- ONCE per element, splits the
// result into f16 hi/lo halves (x = hi + lo) and stores it in the blocked layout [n][C/8][H][W][8] that conv6's
// LDS-DMA copies straight into MFMA B-operand order.  Bytes per element: 4 read + 4 written.
#include "common.h"
#include "elem.h"

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

// fast path: 4 consecutive OUTPUT pixels per thread (float4 loads per channel, 64-byte stores per plane).
// UP = false: plain resolution.  UP = true: nearest x2 up-sampling (unet.py:107) -- the 4 output pixels of a row are 2 source pixels
// (one float2 load per channel); W is the OUTPUT width (W % 4 == 0), HW the output plane.
template <bool UP>
__global__ __launch_bounds__(256) void act_split4_kernel(CatSrc src, const float4* prm, int C, int C8, int HW, int W, _Float16* hi, _Float16* lo,
                                                         unsigned long long* range_ctr) {
    const int p4 = blockIdx.x * 256 + threadIdx.x;          // group of 4 pixels
    const int n = blockIdx.y / C8, c8 = blockIdx.y - n * C8;
    const bool live = p4 * 4 < HW;
    const int HWs = UP ? HW >> 2 : HW;                       // source plane
    size_t soff = (size_t)p4 * 4;
    if (UP) {
        const int y = (p4 * 4) / W, x0 = p4 * 4 - y * W;
        soff = (size_t)(y >> 1) * (W >> 1) + (x0 >> 1);
    }
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C && live) {
            const float* plane = c < src.ca ? src.a + ((size_t)n * src.ca + c) * HWs : src.b + ((size_t)n * src.cb + (c - src.ca)) * HWs;
            if (UP) {
                const float2 s2 = *reinterpret_cast<const float2*>(plane + soff);
                v[j] = make_float4(s2.x, s2.x, s2.y, s2.y);
            } else {
                v[j] = *reinterpret_cast<const float4*>(plane + soff);
            }
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

// ------------------------------------------------------------------------------------------------------------------
// gn_act_small: the WHOLE elementwise chain between two low-resolution convolutions in one launch.  At <= 32 x 32 one
// (image, group) of GroupNorm32 is a few thousand values, so a workgroup can own it completely:
//   pass 1  read the group's channels -- for a tensor that is still split-K slabs (PendingConv) combine them in slab order and
//           add bias / residual exactly as conv6_reduce_kernel does, and store the finished fp32 tensor (skip connections and
//           residuals read it later) -- and accumulate {sum, sum of squares} in fp64;
//           fold to mean / rstd, fold gamma, beta and FiLM into per-channel (a, b)       (gn_prm_kernel's arithmetic)
//   pass 2  normalise + FiLM + SiLU + resample + f16 hi/lo split into conv6's blocked operand planes  (act_split's arithmetic).
// Replaces conv6_reduce + gn_prm + act_split (three latency-bound launches and two HBM round trips of the tensor) for the
// ~33 low-resolution convolutions of a forward.  Deterministic: fixed thread -> element mapping, tree reductions.
// Items of pass 2 are CQ consecutive channels of one output pixel (CQ = 4, 2 or 1, the largest that divides the group size).
struct GnActK {
    const float* a; const float* b; int ca, cb;
    const float* partial; int ksplit; const float* bias; const float* res; int res_mode; float* pend_out;
    const float* gamma; const float* beta;
    const float* film; int film_stride; int film_off; const StepDev* fstep; int frows;
    float act;
    int mode, Hs, Ws, Ho, Wo, C8;
    _Float16* hi; _Float16* lo;
    unsigned long long* range_ctr;
    float4* prm_out; float2* stats_out;        // optional (gradient mode): gn_prm_kernel's tables, for the backward pass
};

template <int CQ>
__device__ __forceinline__ void store_halves(_Float16* dst, const _Float16* v) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    typedef _Float16 half4v __attribute__((ext_vector_type(4)));
    if constexpr (CQ == 4) { half4v t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3]; *reinterpret_cast<half4v*>(dst) = t; }
    else if constexpr (CQ == 2) { half2v t; t[0] = v[0]; t[1] = v[1]; *reinterpret_cast<half2v*>(dst) = t; }
    else dst[0] = v[0];
}

// KMAX: compile-time bound of the split-K factor (0: no pending convolution).  All slab loads of an element are issued before the
// first add -- a run-time loop of load/add pairs is a chain of KMAX dependent HBM latencies.
template <int CQ, int KMAX>
__global__ __launch_bounds__(1024) void gn_act_small_kernel(GnActK p) {
    const int tid = threadIdx.x, NT = blockDim.x;      // 256 ... 1024 threads: enough to give every thread <= ~4 elements
    const int n = blockIdx.x >> 5, g = blockIdx.x & 31;
    const int C = p.ca + p.cb, cg = C >> 5, c0 = g * cg;
    const int HWs = p.Hs * p.Ws, n4 = HWs >> 2;
    // ---- pass 1
    double S = 0.0, SS = 0.0;
    const size_t slab = (size_t)gridDim.x / 32 * p.ca * HWs;          // one split-K slab = the whole [B, ca, Hs, Ws] tensor
    for (int e = tid; e < cg * n4; e += NT) {
        const int k = e / n4, i4 = e - k * n4;
        const int c = c0 + k;
        float4 v;
        if (c < p.ca) {
            const size_t plane = (size_t)n * p.ca + c;
            const size_t o = plane * HWs + (size_t)i4 * 4;
            if (KMAX > 0) {
                float4 t[KMAX > 0 ? KMAX : 1];
#pragma unroll
                for (int q = 0; q < KMAX; ++q)          // slabs beyond ksplit: the last one again (cache hit), not added
                    t[q] = *reinterpret_cast<const float4*>(p.partial + (size_t)(q < p.ksplit ? q : p.ksplit - 1) * slab + o);
                v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < KMAX; ++q)
                    if (q < p.ksplit) { v.x += t[q].x; v.y += t[q].y; v.z += t[q].z; v.w += t[q].w; }
                const float bv = p.bias[c];
                v.x += bv; v.y += bv; v.z += bv; v.w += bv;
                if (p.res) {
                    const int r = i4 * 4;
                    const int y = r / p.Ws, x = r - y * p.Ws;
                    if (p.res_mode == 0) {
                        const float4 t = *reinterpret_cast<const float4*>(p.res + o);
                        v.x = t.x + v.x; v.y = t.y + v.y; v.z = t.z + v.z; v.w = t.w + v.w;
                    } else if (p.res_mode == 1) {
                        const int Hr = p.Hs >> 1, Wr = p.Ws >> 1;
                        const float2 t = *reinterpret_cast<const float2*>(p.res + plane * (size_t)(Hr * Wr) + (y >> 1) * Wr + (x >> 1));
                        v.x = t.x + v.x; v.y = t.x + v.y; v.z = t.y + v.z; v.w = t.y + v.w;
                    } else {
                        const int Wr = p.Ws * 2;
                        const float* rp = p.res + plane * (4 * (size_t)HWs) + (size_t)(2 * y) * Wr + 2 * x;
                        const float4 a0 = *reinterpret_cast<const float4*>(rp), a1 = *reinterpret_cast<const float4*>(rp + 4);
                        const float4 b0 = *reinterpret_cast<const float4*>(rp + Wr), b1 = *reinterpret_cast<const float4*>(rp + Wr + 4);
                        v.x = ((a0.x + a0.y) + (b0.x + b0.y)) * 0.25f + v.x;
                        v.y = ((a0.z + a0.w) + (b0.z + b0.w)) * 0.25f + v.y;
                        v.z = ((a1.x + a1.y) + (b1.x + b1.y)) * 0.25f + v.z;
                        v.w = ((a1.z + a1.w) + (b1.z + b1.w)) * 0.25f + v.w;
                    }
                }
                *reinterpret_cast<float4*>(p.pend_out + o) = v;
            } else {
                v = *reinterpret_cast<const float4*>(p.a + o);
            }
        } else {
            v = *reinterpret_cast<const float4*>(p.b + ((size_t)n * p.cb + (c - p.ca)) * HWs + (size_t)i4 * 4);
        }
        S += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        SS += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    __shared__ double red[2][16];
    __shared__ float sh_mean;
    __shared__ float sh_a[64], sh_b[64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { S += __shfl_xor(S, o, 64); SS += __shfl_xor(SS, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = S; red[1][tid >> 6] = SS; }
    __threadfence_block();
    __syncthreads();
    if (tid < cg) {
        double s = 0.0, ss = 0.0;
        for (int w = 0; w < (NT >> 6); ++w) { s += red[0][w]; ss += red[1][w]; }       // fixed order
        const double cnt = (double)cg * HWs;
        const double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        const int c = c0 + tid;
        float a = rstd * p.gamma[c];
        float b = p.beta[c];
        if (p.film) {   // h = GN(h) * (1 + scale) + shift   (unet.py:250-251)
            const float* f = p.film + (p.fstep ? (size_t)p.fstep->i * p.frows : 0) + (size_t)n * p.film_stride + p.film_off;
            const float sc = 1.0f + f[c];
            const float sh = f[C + c];
            a = a * sc;
            b = b * sc + sh;
        }
        sh_a[tid] = a; sh_b[tid] = b;
        if (tid == 0) sh_mean = (float)mean;
        if (p.prm_out) p.prm_out[(size_t)n * C + c] = make_float4((float)mean, a, b, p.act);
        if (p.stats_out && tid == 0) p.stats_out[blockIdx.x] = make_float2((float)mean, rstd);
    }
    __syncthreads();
    // ---- pass 2
    const float mean = sh_mean;
    const int HWo = p.Ho * p.Wo;
    const int nq = cg / CQ;
    const float* ta = KMAX > 0 ? p.pend_out : p.a;
    bool bad = false;
    for (int q = tid; q < nq * HWo; q += NT) {
        const int cq = q / HWo, pix = q - cq * HWo;
        const int y = pix / p.Wo, x = pix - y * p.Wo;
        const int so = p.mode == 0 ? pix : (p.mode == 1 ? (y >> 1) * p.Ws + (x >> 1) : (2 * y) * p.Ws + 2 * x);
        _Float16 hh[CQ], ll[CQ];
#pragma unroll
        for (int j = 0; j < CQ; ++j) {
            const int k = cq * CQ + j, c = c0 + k;
            const float* plane = c < p.ca ? ta + ((size_t)n * p.ca + c) * HWs : p.b + ((size_t)n * p.cb + (c - p.ca)) * HWs;
            const float a = sh_a[k], b = sh_b[k];
            float v;
            if (p.mode == 2) {
                float v0 = plane[so], v1 = plane[so + 1], v2 = plane[so + p.Ws], v3 = plane[so + p.Ws + 1];
                v0 = (v0 - mean) * a + b; v1 = (v1 - mean) * a + b; v2 = (v2 - mean) * a + b; v3 = (v3 - mean) * a + b;
                if (p.act != 0.f) { v0 = silu_a(v0); v1 = silu_a(v1); v2 = silu_a(v2); v3 = silu_a(v3); }
                v = ((v0 + v1) + (v2 + v3)) * 0.25f;
            } else {
                v = plane[so];
                v = (v - mean) * a + b;
                if (p.act != 0.f) v = silu_a(v);
            }
            bad |= !(fabsf(v) <= 65000.f);
            v = fminf(fmaxf(v, -65000.f), 65000.f);
            hh[j] = (_Float16)v;
            ll[j] = (_Float16)(v - (float)hh[j]);
        }
        const int cb0 = c0 + cq * CQ;
        const size_t o = (((size_t)n * p.C8 + (cb0 >> 3)) * HWo + pix) * 8 + (cb0 & 7);
        store_halves<CQ>(p.hi + o, hh);                        // CQ consecutive halves, (2 CQ)-byte aligned: one store
        if (p.lo) store_halves<CQ>(p.lo + o, ll);
    }
    range_report(bad, p.range_ctr);
}

bool gn_act_small_supported(int C, int Hs, int Ws, int mode) {
    if (C <= 0 || C % 32 || C / 32 > 64) return false;
    const int HWs = Hs * Ws;
    if (HWs % 4 || HWs > 1024 || (size_t)(C / 32) * HWs > 49152) return false;
    if (mode == 2 && ((Hs | Ws) & 1)) return false;
    return true;
}

Status launch_gn_act_small(hipStream_t s, const GnActArgs& a) {
    const int C = a.src.ca + a.src.cb;
    if (!gn_act_small_supported(C, a.Hs, a.Ws, a.mode)) return invalid("gn_act_small: shape not supported");
    if (C % 16) return invalid("gn_act_small: channels must fill whole 16-channel operand chunks");
    GnActK k;
    k.a = a.src.a; k.b = a.src.b; k.ca = a.src.ca; k.cb = a.src.cb;
    k.partial = a.pend.partial; k.ksplit = a.pend.ksplit; k.bias = a.pend.bias; k.res = a.pend.res; k.res_mode = a.pend.res_mode;
    k.pend_out = a.pend.out;
    if (k.partial && (a.pend.out != a.src.a || a.pend.Cout != a.src.ca || a.pend.H != a.Hs || a.pend.W != a.Ws))
        return invalid("gn_act_small: the pending convolution is not the first source tensor");
    k.gamma = a.gamma; k.beta = a.beta; k.film = a.film; k.film_stride = a.film_stride; k.film_off = a.film_off;
    k.fstep = a.fstep; k.frows = a.frows; k.act = a.silu ? 1.0f : 0.0f;
    k.mode = a.mode; k.Hs = a.Hs; k.Ws = a.Ws;
    k.Ho = a.mode == 1 ? a.Hs * 2 : (a.mode == 2 ? a.Hs / 2 : a.Hs);
    k.Wo = a.mode == 1 ? a.Ws * 2 : (a.mode == 2 ? a.Ws / 2 : a.Ws);
    k.C8 = 2 * ((C + 15) / 16);
    k.hi = reinterpret_cast<_Float16*>(a.hi); k.lo = reinterpret_cast<_Float16*>(a.lo);
    k.range_ctr = a.range_ctr;
    k.prm_out = a.prm_out; k.stats_out = a.stats_out;
    const int cg = C / 32;
    const dim3 grid((unsigned)(a.B * 32));
    if (k.partial && (k.ksplit < 2 || k.ksplit > 16)) return invalid("gn_act_small: split-K factor out of range");
    const int kmax = !k.partial ? 0 : (k.ksplit <= 4 ? 4 : (k.ksplit <= 8 ? 8 : 16));
    const int work = (C / 32) * ((a.Hs * a.Ws) >> 2);                      // float4 elements of pass 1 per workgroup
    const unsigned nt = work >= 2048 ? 1024u : (work >= 1024 ? 512u : 256u);
#define DPIR_GNACT(CQ)                                                                                                  \
    do {                                                                                                                \
        if (kmax == 0) hipLaunchKernelGGL((gn_act_small_kernel<CQ, 0>), grid, dim3(nt), 0, s, k);                       \
        else if (kmax == 4) hipLaunchKernelGGL((gn_act_small_kernel<CQ, 4>), grid, dim3(nt), 0, s, k);                  \
        else if (kmax == 8) hipLaunchKernelGGL((gn_act_small_kernel<CQ, 8>), grid, dim3(nt), 0, s, k);                  \
        else hipLaunchKernelGGL((gn_act_small_kernel<CQ, 16>), grid, dim3(nt), 0, s, k);                                \
    } while (0)
    if (cg % 4 == 0) DPIR_GNACT(4);
    else if (cg % 2 == 0) DPIR_GNACT(2);
    else DPIR_GNACT(1);
#undef DPIR_GNACT
    DPIR_HIP(hipGetLastError());
    return Status{};
}

Status launch_act_split(hipStream_t s, CatSrc src, const float4* prm, int mode, int B, int H, int W, void* hi, void* lo,
                        unsigned long long* range_ctr) {
    const int C = src.ca + src.cb, C8 = 2 * ((C + 15) / 16);   // whole 16-channel K chunks (zero padded)
    dim3 grid((unsigned)((H * W + 255) / 256), (unsigned)(B * C8));
    _Float16* h = reinterpret_cast<_Float16*>(hi);
    _Float16* l = reinterpret_cast<_Float16*>(lo);
    if (mode == 0 && (H * W) % 4 == 0) {
        dim3 g4((unsigned)((H * W / 4 + 255) / 256), (unsigned)(B * C8));
        hipLaunchKernelGGL(act_split4_kernel<false>, g4, dim3(256), 0, s, src, prm, C, C8, H * W, W, h, l, range_ctr);
    } else if (mode == 1 && W % 4 == 0 && H % 2 == 0) {
        dim3 g4((unsigned)((H * W / 4 + 255) / 256), (unsigned)(B * C8));
        hipLaunchKernelGGL(act_split4_kernel<true>, g4, dim3(256), 0, s, src, prm, C, C8, H * W, W, h, l, range_ctr);
    } else if (mode == 0) hipLaunchKernelGGL(act_split_kernel<0>, grid, dim3(256), 0, s, src, prm, C, C8, H, W, h, l, range_ctr);
    else if (mode == 1) hipLaunchKernelGGL(act_split_kernel<1>, grid, dim3(256), 0, s, src, prm, C, C8, H, W, h, l, range_ctr);
    else hipLaunchKernelGGL(act_split_kernel<2>, grid, dim3(256), 0, s, src, prm, C, C8, H, W, h, l, range_ctr);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
