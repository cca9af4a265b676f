// Input-gradient kernels of the UNet and the DPS data-consistency step.
//
// Replaces what `torch.autograd.grad(outputs=norm, inputs=x)` walks in the reference's DPS branch
// (utils/utils_model.py:390-394 `grad_and_value`, main_ddpir.py:370-373, 434-438): the vector-Jacobian product of
//     x -> UNet eps(x) -> x0 = clamp(c1 x - c2 eps) -> || measurement - Resizer(x0) ||_2
// Only INPUT gradients exist (weights are frozen, main_ddpir.py:238-239), so a layer's backward is: dgrad of the convolutions
// (the forward kernels on transposed / flipped weights, unet_bwd.hip), GroupNorm + FiLM + SiLU backward (here), the attention
// core backward (here, as small batched GEMMs over the materialised probabilities), and the adjoints of the resampling ops.
// Everything is fp32 with fp64 reductions and a fixed summation order (no atomics): gradients are bitwise reproducible.
#include "common.h"
#include "elem.h"
#include "grad.h"

namespace dpir {

__device__ __forceinline__ float sigmoid_g(float v) { return 1.0f / (1.0f + expf(-v)); }

// Adjoint read of the resampling that sits between SiLU and the convolution (act.hip modes): the gradient arrives at the conv
// input resolution (Ho x Wo) and is needed at the source resolution (Hs x Ws).
//   mode 0: identity;  mode 1 (forward nearest x2 up, unet.py:107): sum of the 2x2 children;
//   mode 2 (forward 2x2 average pooling, unet.py:136): a quarter of the parent.
__device__ __forceinline__ float adj_read(const float* plane, int mode, int y, int x, int Ws) {
    if (mode == 0) return plane[(size_t)y * Ws + x];
    if (mode == 1) {
        const int Wo = Ws * 2;
        const float* p = plane + (size_t)(2 * y) * Wo + 2 * x;
        return (p[0] + p[1]) + (p[Wo] + p[Wo + 1]);
    }
    const int Wo = Ws >> 1;
    return plane[(size_t)(y >> 1) * Wo + (x >> 1)] * 0.25f;
}

// G = d(loss)/d(x_hat * a-part) expressed per element: with u = (x - mean) * a + b, act = silu(u) (or u), and dAs the adjoint-
// resampled incoming gradient,  G = dAs * silu'(u) * a.   GroupNorm backward:  dx = G - mean_grp(G) - x_hat * mean_grp(G * x_hat).
__device__ __forceinline__ float gn_G(float xv, float dAs, float4 m, float rstd, float* xhat) {
    const float xc = xv - m.x;
    *xhat = xc * rstd;
    const float u = xc * m.y + m.z;
    float du = dAs;
    if (m.w != 0.f) {
        const float sg = sigmoid_g(u);
        du = dAs * (sg * (1.0f + u * (1.0f - sg)));
    }
    return du * m.y;
}

// pass 1: grid (B * 32, NS): workgroup (n, g, part) sums a contiguous share of the group's elements -> fp64 {sum G, sum G x_hat}
// partials [B * 32][NS]; pass 2 folds the NS partials of its group in index order.  (One workgroup per group left 256 workgroups with
// 262 144 elements each at 256^2: hundreds of microseconds per GroupNorm site.)
constexpr int GN_BWD_Q = 4;      // float4 groups per thread of gn_bwd_apply_kernel

__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(GnBwdArgs p, int NS) {
    const int n = blockIdx.x >> 5, g = blockIdx.x & 31;
    const int part = blockIdx.y;
    const int C = p.x.ca + p.x.cb, cg = C >> 5;
    const int HWs = p.Hs * p.Ws;
    const int Ho = p.mode == 1 ? p.Hs * 2 : (p.mode == 2 ? p.Hs >> 1 : p.Hs), Wo = p.mode == 1 ? p.Ws * 2 : (p.mode == 2 ? p.Ws >> 1 : p.Ws);
    const float rstd = p.stats[blockIdx.x].y;
    const long long total = (long long)cg * HWs;
    const long long per = ((total + NS - 1) / NS + 3) & ~3ll;           // multiple of 4: float4 runs never straddle two workgroups
    const long long lo = (long long)part * per, hi = lo + per < total ? lo + per : total;
    double S1 = 0.0, S2 = 0.0;
    const bool vec = p.mode == 0 && (HWs & 3) == 0;
    if (vec) {
        // four float4 groups of a thread are requested together (clamped index: no branch around the loads) and then accumulated in the order a
        // one-at-a-time loop would visit them -- same sums, a quarter of the serialised memory round trips (the loop was latency-bound: 2.1 TB/s at 256^2)
        const long long last4 = (hi - 1) / 4;
        for (long long b4 = lo / 4 + threadIdx.x; b4 * 4 < hi; b4 += 1024) {
            float4 xv[4], dv[4], mm[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long e = (b4 + 256 * q <= last4 ? b4 + 256 * q : last4) * 4;
                const int k = (int)(e / HWs), i = (int)(e - (long long)k * HWs);
                const int c = g * cg + k;
                const float* xp = c < p.x.ca ? p.x.a + ((size_t)n * p.x.ca + c) * HWs : p.x.b + ((size_t)n * p.x.cb + (c - p.x.ca)) * HWs;
                xv[q] = *reinterpret_cast<const float4*>(xp + i);
                dv[q] = *reinterpret_cast<const float4*>(p.dA + ((size_t)n * C + c) * HWs + i);
                mm[q] = p.prm[(size_t)n * C + c];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if ((b4 + 256 * q) * 4 >= hi) continue;
                const float xs[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w}, ds[4] = {dv[q].x, dv[q].y, dv[q].z, dv[q].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float xh;
                    const float G = gn_G(xs[u], ds[u], mm[q], rstd, &xh);
                    S1 += (double)G;
                    S2 += (double)G * (double)xh;
                }
            }
        }
    } else {
        for (long long e = lo + threadIdx.x; e < hi; e += 256) {
            const int k = (int)(e / HWs), i = (int)(e - (long long)k * HWs);
            const int c = g * cg + k;
            const float* xp = c < p.x.ca ? p.x.a + ((size_t)n * p.x.ca + c) * HWs : p.x.b + ((size_t)n * p.x.cb + (c - p.x.ca)) * HWs;
            const float* dp = p.dA + ((size_t)n * C + c) * ((size_t)Ho * Wo);
            const int y = i / p.Ws, x = i - y * p.Ws;
            float xh;
            const float G = gn_G(xp[i], adj_read(dp, p.mode, y, x, p.Ws), p.prm[(size_t)n * C + c], rstd, &xh);
            S1 += (double)G;
            S2 += (double)G * (double)xh;
        }
    }
    __shared__ double red[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { S1 += __shfl_xor(S1, o, 64); S2 += __shfl_xor(S2, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = S1; red[1][threadIdx.x >> 6] = S2; }
    __syncthreads();
    if (threadIdx.x == 0)
        p.sums[(size_t)blockIdx.x * NS + part] = make_double2((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
}

// pass 2: grid (B * C, ceil(HWs / 1024)): one channel plane per blockIdx.x, 4 consecutive pixels per thread (float4 when the plane
// allows it); dx written (acc == 0) or accumulated (acc != 0) into the gradient of the source tensor the channel belongs to (the two
// halves of a virtual concat have their own gradient buffers).  The NS partial sums of the group are folded ONCE per workgroup, in
// index order (first version: every thread re-read all of them -- 24 ms of a 60 ms backward pass).
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnBwdArgs p, int NS) {
    const int C = p.x.ca + p.x.cb, cg = C >> 5;
    const int n = blockIdx.x / C, c = blockIdx.x - n * C;
    const int HWs = p.Hs * p.Ws;
    const int Ho = p.mode == 1 ? p.Hs * 2 : (p.mode == 2 ? p.Hs >> 1 : p.Hs), Wo = p.mode == 1 ? p.Ws * 2 : (p.mode == 2 ? p.Ws >> 1 : p.Ws);
    const int g = c / cg;
    const bool in_a = c < p.x.ca;
    const size_t base = in_a ? ((size_t)n * p.x.ca + c) * HWs : ((size_t)n * p.x.cb + (c - p.x.ca)) * HWs;
    const float* xp = (in_a ? p.x.a : p.x.b) + base;
    float* dst = (in_a ? p.ga : p.gb) + base;
    const int acc = in_a ? p.acc_a : p.acc_b;
    const float* dp = p.dA + ((size_t)n * C + c) * ((size_t)Ho * Wo);
    const bool vec = p.mode == 0 && (HWs & 3) == 0;
    // GN_BWD_Q float4 groups per thread (4096 pixels per workgroup).  Everything that does not depend on the group's folded sums -- the pixels, their
    // incoming gradients, the accumulation target -- is requested BEFORE the fold (it used to be a chain of four dependent round trips per 12 KB: 1.9 TB/s)
    const int j0 = blockIdx.y * (256 * GN_BWD_Q) + threadIdx.x;         // float4 index of this thread's first group; the others are 256 apart
    const int n4 = HWs >> 2;
    const float* ex = (p.extra && in_a) ? p.extra + base : nullptr;
    float4 xv[GN_BWD_Q], dv[GN_BWD_Q], ov[GN_BWD_Q];
    if (vec) {
#pragma unroll
        for (int q = 0; q < GN_BWD_Q; ++q) {
            const int j = min(j0 + 256 * q, n4 - 1) * 4;
            xv[q] = *reinterpret_cast<const float4*>(xp + j);
            dv[q] = *reinterpret_cast<const float4*>(dp + j);
            ov[q] = acc ? *reinterpret_cast<const float4*>(dst + j) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (ex) { const float4 t = *reinterpret_cast<const float4*>(ex + j); ov[q].x += t.x; ov[q].y += t.y; ov[q].z += t.z; ov[q].w += t.w; }
        }
    }
    const float rstd = p.stats[n * 32 + g].y;
    const float4 m = p.prm[(size_t)n * C + c];
    __shared__ double2 part_sh[64];
    __shared__ float m_sh[2];
    if ((int)threadIdx.x < NS) part_sh[threadIdx.x] = p.sums[(size_t)(n * 32 + g) * NS + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        double s1 = 0.0, s2 = 0.0;
        for (int q = 0; q < NS; ++q) { s1 += part_sh[q].x; s2 += part_sh[q].y; }
        const double cnt = (double)cg * HWs;
        m_sh[0] = (float)(s1 / cnt); m_sh[1] = (float)(s2 / cnt);
    }
    __syncthreads();
    const float m1 = m_sh[0], m2 = m_sh[1];
    if (vec) {
#pragma unroll
        for (int q = 0; q < GN_BWD_Q; ++q) {
            const int j = j0 + 256 * q;
            if (j >= n4) break;
            float4 o = ov[q];
            float xh, G;
            G = gn_G(xv[q].x, dv[q].x, m, rstd, &xh); o.x += G - m1 - xh * m2;
            G = gn_G(xv[q].y, dv[q].y, m, rstd, &xh); o.y += G - m1 - xh * m2;
            G = gn_G(xv[q].z, dv[q].z, m, rstd, &xh); o.z += G - m1 - xh * m2;
            G = gn_G(xv[q].w, dv[q].w, m, rstd, &xh); o.w += G - m1 - xh * m2;
            *reinterpret_cast<float4*>(dst + j * 4) = o;
        }
        return;
    }
    for (int q = 0; q < GN_BWD_Q; ++q) {
        const int i0 = (j0 + 256 * q) * 4;
        for (int i = i0; i < i0 + 4 && i < HWs; ++i) {
            const int y = i / p.Ws, x = i - y * p.Ws;
            float xh;
            const float G = gn_G(xp[i], adj_read(dp, p.mode, y, x, p.Ws), m, rstd, &xh);
            const float dx = G - m1 - xh * m2;
            float o = acc ? dst[i] : 0.f;
            if (ex) o += ex[i];
            dst[i] = o + dx;
        }
    }
}

// partial sums per (image, group): ~8192 elements per workgroup, at most 64
int gn_bwd_parts(int C, int Hs, int Ws) {
    const long long total = (long long)(C / 32) * Hs * Ws;
    long long ns = (total + 8191) / 8192;
    return (int)(ns < 1 ? 1 : (ns > 64 ? 64 : ns));
}

Status launch_gn_bwd(hipStream_t s, const GnBwdArgs& a, int B) {
    const int C = a.x.ca + a.x.cb;
    if (C % 32 || !a.prm || !a.stats || !a.sums || !a.dA || !a.ga || (a.x.cb && !a.gb)) return invalid("gn_bwd: bad arguments");
    if (a.mode == 2 && ((a.Hs | a.Ws) & 1)) return invalid("gn_bwd: pooled source must be even");
    const int NS = gn_bwd_parts(C, a.Hs, a.Ws);
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(B * 32, NS), dim3(256), 0, s, a, NS);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(B * C, (a.Hs * a.Ws + 1024 * GN_BWD_Q - 1) / (1024 * GN_BWD_Q)), dim3(256), 0, s, a, NS);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// dst[n, c, :] (=|+=) adj(src[n, c0 + c, :]) for c < Cd: the identity / resampled skip connection and the channel split of a
// concat gradient.  src has Cs channels at the OUTPUT resolution of `mode`, dst Cd channels at Hs x Ws.
__global__ __launch_bounds__(256) void accum_adj_kernel(const float* src, int Cs, int c0, float* dst, int Cd, int mode, int Hs, int Ws, int acc) {
    const int n = blockIdx.x / Cd, c = blockIdx.x - n * Cd;
    const int HWs = Hs * Ws;
    if (mode == 0 && (HWs & 3) == 0) {                 // plain copy / add of a channel plane: 4 pixels per thread, one pass per 1024
        const float* sp4 = src + ((size_t)n * Cs + c0 + c) * HWs;
        float* dp4 = dst + ((size_t)n * Cd + c) * HWs;
        for (int i4 = (blockIdx.y * 256 + threadIdx.x) * 4; i4 < HWs; i4 += gridDim.y * 1024) {
            float4 v = *reinterpret_cast<const float4*>(sp4 + i4);
            if (acc) { const float4 o = *reinterpret_cast<const float4*>(dp4 + i4); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4*>(dp4 + i4) = v;
        }
        return;
    }
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= HWs) return;
    const int Ho = mode == 1 ? Hs * 2 : (mode == 2 ? Hs >> 1 : Hs), Wo = mode == 1 ? Ws * 2 : (mode == 2 ? Ws >> 1 : Ws);
    const float* sp = src + ((size_t)n * Cs + c0 + c) * ((size_t)Ho * Wo);
    const int y = i / Ws, x = i - y * Ws;
    const float v = adj_read(sp, mode, y, x, Ws);
    const size_t o = ((size_t)n * Cd + c) * HWs + i;
    dst[o] = acc ? dst[o] + v : v;
}
Status launch_accum_adj(hipStream_t s, const float* src, int Cs, int c0, float* dst, int Cd, int mode, int B, int Hs, int Ws, bool acc) {
    const int per_block = (mode == 0 && ((Hs * Ws) & 3) == 0) ? 1024 : 256;
    hipLaunchKernelGGL(accum_adj_kernel, dim3(B * Cd, (Hs * Ws + per_block - 1) / per_block), dim3(256), 0, s, src, Cs, c0, dst, Cd, mode, Hs, Ws, acc ? 1 : 0);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// f16 dgrad (unet_bwd.hip): gradients span many orders of magnitude, the f16 operand split does not -- dY is brought to
// max|dY| * s in [512, 1024) with a power of two s before the split (exactly what the loader does to the weights) and the convolution
// epilogue multiplies by 1 / s.  Stage 1: per-workgroup max; stage 2: s, 1 / s and the uniform {0, s, 0, 0} table act_split / conv5 read.
__global__ __launch_bounds__(256) void absmax_kernel(const float* x, size_t total, float* part) {
    float m = 0.f;
    const size_t n4 = total >> 2;                       // tensors here are [B, C, H, W] with H * W % 4 == 0 or tiny
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {     // four requests in flight per thread (max is order-free)
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const size_t j = i + q * stride; v[q] = x4[j < n4 ? j : n4 - 1]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[q].x), fabsf(v[q].y))), fmaxf(fabsf(v[q].z), fabsf(v[q].w)));
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void grad_scale_kernel(const float* part, int nparts, float* scal, float4* prm, int n_prm) {
    __shared__ float sh;
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float sc = 1.0f;
        if (mx > 0.f && mx < 3.0e38f) {
            int ex = (int)floorf(log2f(1024.0f / mx));
            ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
            sc = exp2f((float)ex);
            while (mx * sc >= 1024.0f) sc *= 0.5f;
        }
        scal[0] = sc; scal[1] = 1.0f / sc;
        sh = sc;
    }
    __syncthreads();
    const float sc = sh;
    for (int i = threadIdx.x; i < n_prm; i += 256) prm[i] = make_float4(0.f, sc, 0.f, 0.f);
}
Status launch_grad_scale(hipStream_t s, const float* x, size_t total, float* part, float* scal, float4* prm, int n_prm) {
    const int nparts = 512;
    hipLaunchKernelGGL(absmax_kernel, dim3(nparts), dim3(256), 0, s, x, total, part);
    hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(256), 0, s, part, nparts, scal, prm, n_prm);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------------------------------------------------------- attention
// Batched fp32 GEMM for the attention backward (sizes 64 x T x T and T x T x 64, T <= 1024: < 0.5 % of a backward pass):
//   C[b] (M x N, ldc) = alpha * opA(A[b]) * opB(B[b]);  TA: A is stored K x M (lda), else M x K;  TB: B is stored N x K (ldb), else K x N.
// 64 x 64 tile per workgroup, 16-deep LDS stages, 4 x 4 outputs per thread, k summed in ascending order.
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void bgemm_kernel(const float* A, const float* Bm, float* Cm, int M, int N, int K, int lda, int ldb, int ldc,
                                                    size_t sA, size_t sB, size_t sC, float alpha) {
    __shared__ float As[16][64 + 1], Bs[16][64 + 1];
    const float* a = A + (size_t)blockIdx.z * sA;
    const float* b = Bm + (size_t)blockIdx.z * sB;
    float* c = Cm + (size_t)blockIdx.z * sC;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            int kk, mm;
            if (TA) { kk = e >> 6; mm = e & 63; } else { mm = e >> 4; kk = e & 15; }       // walk the contiguous axis of the stored operand
            const int gm = m0 + mm, gk = k0 + kk;
            As[kk][mm] = (gm < M && gk < K) ? (TA ? a[(size_t)gk * lda + gm] : a[(size_t)gm * lda + gk]) : 0.f;
        }
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            int kk, nn;
            if (TB) { nn = e >> 4; kk = e & 15; } else { kk = e >> 6; nn = e & 63; }
            const int gn = n0 + nn, gk = k0 + kk;
            Bs[kk][nn] = (gn < N && gk < K) ? (TB ? b[(size_t)gn * ldb + gk] : b[(size_t)gk * ldb + gn]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = As[kk][ty * 4 + i]; bv[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gm = m0 + ty * 4 + i, gn = n0 + tx * 4 + j;
            if (gm < M && gn < N) c[(size_t)gm * ldc + gn] = alpha * acc[i][j];
        }
}
static Status bgemm(hipStream_t s, bool ta, bool tb, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                    size_t sA, size_t sB, size_t sC, int batch, float alpha) {
    dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
    if (ta && tb) hipLaunchKernelGGL((bgemm_kernel<true, true>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha);
    else if (ta) hipLaunchKernelGGL((bgemm_kernel<true, false>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha);
    else if (tb) hipLaunchKernelGGL((bgemm_kernel<false, true>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha);
    else hipLaunchKernelGGL((bgemm_kernel<false, false>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// rows of P = softmax(logits) in place (one wave per row of T <= 1024 entries)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* P, int T, size_t rows) {
    const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float* row = P + r * T;
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int s = lane + 64 * i; v[i] = s < T ? row[s] : -INFINITY; mx = fmaxf(mx, v[i]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = lane + 64 * i < T ? expf(v[i] - mx) : 0.f; sum += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) if (lane + 64 * i < T) row[lane + 64 * i] = v[i] * inv;
}
// dS = P * (dP - sum_s P dP) row by row, in place over dP
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* P, float* dP, int T, size_t rows) {
    const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* pr = P + r * T;
    float* dr = dP + r * T;
    float pv[16], dv[16];
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int s = lane + 64 * i;
        pv[i] = s < T ? pr[s] : 0.f; dv[i] = s < T ? dr[s] : 0.f;
        d = fmaf(pv[i], dv[i], d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) if (lane + 64 * i < T) dr[lane + 64 * i] = pv[i] * (dv[i] - d);
}

// QKVAttentionLegacy backward (unet.py:337-354): qkv [B, heads * 3 * 64, T] (per head q | k | v), dAtt [B, heads * 64, T] -> dqkv.
// P / dP: scratch [B * heads, T, T] each.
Status launch_attention_bwd(hipStream_t s, const float* qkv, const float* dAtt, float* dqkv, float* P, float* dP, int B, int C, int T) {
    if (C % 64 || T > 1024) return invalid("attention_bwd: head channels must be 64 and T <= 1024");
    const int heads = C / 64, nb = B * heads;
    const size_t sQ = (size_t)192 * T, sA = (size_t)64 * T, sP = (size_t)T * T;
    const float* q = qkv; const float* k = qkv + (size_t)64 * T; const float* v = qkv + (size_t)128 * T;
    float* dq = dqkv; float* dk = dqkv + (size_t)64 * T; float* dv = dqkv + (size_t)128 * T;
    const float sc = 0.125f;                                    // (64^-1/4)^2, the two-sided scaling of the forward
    const size_t rows = (size_t)nb * T;
    DPIR_TRY(bgemm(s, true, false, q, k, P, T, T, 64, T, T, T, sQ, sQ, sP, nb, sc));            // logits[t][s] = sc * sum_c q[c][t] k[c][s]
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, P, T, rows);
    DPIR_TRY(bgemm(s, false, false, dAtt, P, dv, 64, T, T, T, T, T, sA, sP, sQ, nb, 1.0f));      // dv[c][s] = sum_t dA[c][t] P[t][s]
    DPIR_TRY(bgemm(s, true, false, dAtt, v, dP, T, T, 64, T, T, T, sA, sQ, sP, nb, 1.0f));       // dP[t][s] = sum_c dA[c][t] v[c][s]
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, P, dP, T, rows);
    DPIR_TRY(bgemm(s, false, true, k, dP, dq, 64, T, T, T, T, T, sQ, sP, sQ, nb, sc));           // dq[c][t] = sc * sum_s k[c][s] dS[t][s]
    DPIR_TRY(bgemm(s, false, false, q, dP, dk, 64, T, T, T, T, T, sQ, sP, sQ, nb, sc));          // dk[c][s] = sc * sum_t q[c][t] dS[t][s]
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------------------------------------------------------- DPS step
// p_mean_variance + p_sample for LEARNED_RANGE variance (gaussian_diffusion.py:232-326, 395-439):
//   x0 = clamp(c1 x - c2 eps, -1, 1);  mean = pc1 x0 + pc2 x;  logvar = frac max_log + (1 - frac) min_log, frac = (v + 1) / 2
//   x_prev = mean + (t != 0) exp(0.5 logvar) noise.          inside[i] = 1 where the clamp passes the gradient (-1 <= u <= 1).
__global__ void psample_kernel(const float* x, const float* out6, int out_ch, const float* noise, PSampleCoef cf, float* x0, float* xprev,
                               unsigned char* inside, int HW, size_t total) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / ((size_t)3 * HW), r = i - n * (size_t)3 * HW;
        const size_t c = r / HW, p = r - c * HW;
        const float eps = out6[(n * out_ch + c) * HW + p];
        const float vv = out6[(n * out_ch + 3 + c) * HW + p];
        const float u = cf.c1 * x[i] - cf.c2 * eps;
        const float xs = fminf(fmaxf(u, -1.0f), 1.0f);
        x0[i] = xs;
        if (inside) inside[i] = (u >= -1.0f && u <= 1.0f) ? 1 : 0;
        if (xprev) {
            const float frac = (vv + 1.0f) / 2.0f;
            const float lv = frac * cf.max_log + (1.0f - frac) * cf.min_log;
            if (cf.ddim) {      // eta = 0: eps re-derived from the clamped x0 (gaussian_diffusion.py:345-349, 566), sigma * noise == 0
                const float e2 = (cf.c1 * x[i] - xs) / cf.c2;
                xprev[i] = xs * cf.sa_prev + cf.s1m_prev * e2;
            } else {
                const float mean = cf.pc1 * xs + cf.pc2 * x[i];
                xprev[i] = mean + cf.nonzero * expf(0.5f * lv) * noise[i];
            }
        }
    }
}
Status launch_psample(hipStream_t s, const float* x, const float* out6, int out_ch, const float* noise, const PSampleCoef& cf, float* x0,
                      float* xprev, unsigned char* inside, int B, int HW) {
    const size_t total = (size_t)B * 3 * HW;
    hipLaunchKernelGGL(psample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out6, out_ch, noise, cf, x0, xprev, inside, HW, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// diff = measurement - down, measurement = sa (ma y + mb) + s1m noise (DPS_yt: y_t of main_ddpir.py:440; sa = 1, no noise otherwise);
// partial sums of diff^2 (fp64, one slot per workgroup, folded in order by the next kernel).  lp: y re-read from the loop block.
__global__ __launch_bounds__(256) void diff_norm_kernel(const float* y, float ma, float mb, float sa, float s1m, const float* noise, const float* down,
                                                        float* diff, size_t total, double* part, const LoopDev* lp) {
#pragma clang fp contract(off)
    if (lp) y = lp->y;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float m = y[i] * ma + mb;
        if (noise) m = sa * m + s1m * noise[i];
        const float d = m - down[i];
        diff[i] = d;
        s += (double)d * (double)d;
    }
    __shared__ double red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void norm_fold_kernel(const double* part, int n, float* norm_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += part[i];
        norm_out[0] = (float)sqrt(s);
    }
}
__global__ void norm_fold_ssq_kernel(const double* part, int n, double* ssq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += part[i];
        ssq[0] = s;
    }
}
__global__ void norm_sqrt_kernel(const double* ssq, float* norm_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) norm_out[0] = (float)sqrt(ssq[0]);
}
Status launch_norm_fold_ssq(hipStream_t s, const double* part, int n, double* ssq) {
    hipLaunchKernelGGL(norm_fold_ssq_kernel, dim3(1), dim3(64), 0, s, part, n, ssq);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_norm_sqrt(hipStream_t s, const double* ssq, float* norm_out) {
    hipLaunchKernelGGL(norm_sqrt_kernel, dim3(1), dim3(64), 0, s, ssq, norm_out);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
__global__ void eps_from_xstart_kernel(const float* x, const float* x0, float sa, float s1m, int score, float* out, size_t total) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = (x[i] - sa * x0[i]) / s1m;
        if (score) v = -v / s1m;
        out[i] = v;
    }
}
Status launch_eps_from_xstart(hipStream_t s, const float* x, const float* x0, float sa, float s1m, int score, float* out, size_t total) {
    hipLaunchKernelGGL(eps_from_xstart_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, x0, sa, s1m, score, out, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_diff_norm(hipStream_t s, const float* y, float ma, float mb, const float* down, float* diff, size_t total, double* part, int nparts,
                        float* norm_out, float sa, float s1m, const float* noise, const LoopDev* lp) {
    hipLaunchKernelGGL(diff_norm_kernel, dim3(nparts), dim3(256), 0, s, y, ma, mb, sa, s1m, noise, down, diff, total, part, lp);
    if (norm_out) hipLaunchKernelGGL(norm_fold_kernel, dim3(1), dim3(64), 0, s, part, nparts, norm_out);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// Transposed band resampling (adjoint of elem.hip's band_resample_kernel along one axis):
//   forward  out[p, o, q] = sum_t w[t, o] in[p, idx[t, o], q]      ->      gin[p, i, q] = sum_{(o, t): idx[t, o] = i} w[t, o] gout[p, o, q]
// (tables stored [taps][L_out] as elem.hip's resizer_band writes them)
// evaluated as a gather over the (few) outputs that touch input i (fixed order: o ascending, t ascending).
__global__ void band_resample_T_kernel(const float* gout, const float* w, const int* idx, int taps, int L_in, int L_out, int inner, float scale,
                                       float* gin, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t q = e % inner, r = e / inner;
        const int i = (int)(r % L_in);
        const size_t p = r / L_in;
        // outputs whose support can contain i: the forward kernel is centred at (o + 0.5) * L_in / L_out, support <= taps inputs wide
        const int sf = L_in / L_out;
        int o_lo = (i - taps) / sf - 1, o_hi = (i + taps) / sf + 1;
        if (o_lo < 0) o_lo = 0;
        if (o_hi > L_out - 1) o_hi = L_out - 1;
        float acc = 0.f;
        for (int o = o_lo; o <= o_hi; ++o)
            for (int t = 0; t < taps; ++t)
                if (idx[t * L_out + o] == i) acc = fmaf(w[t * L_out + o], gout[(p * L_out + o) * inner + q], acc);
        gin[e] = acc * scale;
    }
}
Status launch_band_resample_T(hipStream_t s, const float* gout, const float* w, const int* idx, int taps, int P, int L_in, int L_out, int inner,
                              float scale, float* gin) {
    const size_t total = (size_t)P * L_in * inner;
    hipLaunchKernelGGL(band_resample_T_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gout, w, idx, taps, L_in, L_out, inner, scale, gin, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// g0 = -gup / norm (d norm / d x0);  d_out6[:, 0:3] = -c2 inside g0,  d_out6[:, 3:] = 0;  direct[i] = c1 inside g0
__global__ void dps_seed_kernel(const float* gup, const float* norm, const unsigned char* inside, float c1, float c2, int out_ch, float* dout6,
                                float* direct, int HW, size_t total6) {
#pragma clang fp contract(off)
    const float inv = norm[0] > 0.f ? 1.0f / norm[0] : 0.f;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total6; e += (size_t)gridDim.x * blockDim.x) {
        const size_t n = e / ((size_t)out_ch * HW), r = e - n * (size_t)out_ch * HW;
        const size_t c = r / HW, p = r - c * HW;
        if (c >= 3) { dout6[e] = 0.f; continue; }
        const size_t i = (n * 3 + c) * HW + p;
        const float g0 = inside[i] ? -(gup[i] * inv) : 0.f;
        dout6[e] = -(c2 * g0);
        direct[i] = c1 * g0;
    }
}
Status launch_dps_seed(hipStream_t s, const float* gup, const float* norm, const unsigned char* inside, float c1, float c2, int out_ch, float* dout6,
                       float* direct, int B, int HW) {
    const size_t total6 = (size_t)B * out_ch * HW;
    hipLaunchKernelGGL(dps_seed_kernel, dim3((unsigned)((total6 + 255) / 256)), dim3(256), 0, s, gup, norm, inside, c1, c2, out_ch, dout6, direct, HW, total6);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// x <- xprev - (direct + dx_net) * step_scale      (main_ddpir.py:437: x = xt - norm_grad * 1.)
__global__ void dps_update_kernel(const float* xprev, const float* direct, const float* dx_net, float step_scale, float* x, float* grad_out, size_t total) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float g = direct[i] + dx_net[i];
        if (grad_out) grad_out[i] = g;
        if (x) x[i] = xprev[i] - g * step_scale;
    }
}
Status launch_dps_update(hipStream_t s, const float* xprev, const float* direct, const float* dx_net, float step_scale, float* x, float* grad_out,
                         size_t total) {
    hipLaunchKernelGGL(dps_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xprev, direct, dx_net, step_scale, x, grad_out, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

__global__ void neg_scale_by_norm_kernel(const float* gup, const float* norm, float* out, size_t total) {
#pragma clang fp contract(off)
    const float nv = norm[0];
    const float inv = nv > 0.f ? 1.0f / nv : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) out[i] = -(gup[i] * inv);
}
Status launch_neg_scale_by_norm(hipStream_t s, const float* gup, const float* norm, float* out, size_t total) {
    hipLaunchKernelGGL(neg_scale_by_norm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gup, norm, out, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// x <- x - norm_grad * coef, norm_grad = -gup / norm (gradient of || m - A(x) || w.r.t. x), coef = norm * scale / rho:
//   first-order data step  (main_ddpir.py:428):  x0 = x0 - norm_grad * norm / rhos[t_i]                        (scale = 1)
//   DPS_yt                 (main_ddpir.py:444):  x  = xt - norm_grad * lambda * norm / rhos[t_i] * 0.35        (evaluated in that order)
// rho from the device step block when sp != null.
__global__ void grad_step_kernel(const float* src, const float* gup, const float* norm, float lam, float rho, float tail, float* dst, size_t total,
                                 const StepDev* sp) {
#pragma clang fp contract(off)
    if (sp) rho = sp->tau;
    const float nv = norm[0];
    const float inv = nv > 0.f ? 1.0f / nv : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float ng = -(gup[i] * inv);
        dst[i] = src[i] - ng * lam * nv / rho * tail;
    }
}
Status launch_grad_step(hipStream_t s, const float* src, const float* gup, const float* norm, float lam, float rho, float tail, float* dst, size_t total,
                        const StepDev* sp) {
    hipLaunchKernelGGL(grad_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, gup, norm, lam, rho, tail, dst, total, sp);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
