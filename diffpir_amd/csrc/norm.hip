// GroupNorm statistics, GroupNorm/FiLM parameter folding, timestep-embedding MLP and the per-ResBlock
// FiLM projections.
//
// Replaces: GroupNorm32 (guided_diffusion/nn.py:17-19,93-100: 32 groups, eps 1e-5, fp32),
// timestep_embedding (nn.py:103-121), time_embed (unet.py:471-475, 648), emb_layers
// (unet.py:199-205, 245).  The normalisation itself is NOT a pass over the tensor: only the statistics
// are computed here (one read of the tensor, fp64 accumulation), and the affine + FiLM + SiLU are
// folded into the consumer convolution's prologue (conv.hip) through the per-(image,channel)
// parameter table written by gn_prm_kernel.
#include "common.h"
#include "elem.h"

namespace dpir {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Pass 1: one workgroup per (image, channel) plane -> fp64 {sum, sum of squares}; pass 2 (gn_prm_kernel) folds the
// channels of a group in a fixed order.  B*C workgroups keep every CU busy even at batch 1 (B*32 group-blocks did not).
__global__ __launch_bounds__(256) void gn_stats_kernel(CatSrc src, int HW, double2* part) {
    const int C = src.ca + src.cb;
    const int n = blockIdx.x / C, c = blockIdx.x - n * C;
    const float* plane = c < src.ca ? src.a + ((size_t)n * src.ca + c) * HW
                                    : src.b + ((size_t)n * src.cb + (c - src.ca)) * HW;
    double s = 0.0, ss = 0.0;
    if ((HW & 3) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(plane);
        const int n4 = HW >> 2;
        int i = threadIdx.x;
        for (; i + 768 < n4; i += 1024) {
            float4 v0 = p4[i], v1 = p4[i + 256], v2 = p4[i + 512], v3 = p4[i + 768];
            s += ((double)v0.x + (double)v0.y) + ((double)v0.z + (double)v0.w);
            ss += ((double)v0.x * v0.x + (double)v0.y * v0.y) + ((double)v0.z * v0.z + (double)v0.w * v0.w);
            s += ((double)v1.x + (double)v1.y) + ((double)v1.z + (double)v1.w);
            ss += ((double)v1.x * v1.x + (double)v1.y * v1.y) + ((double)v1.z * v1.z + (double)v1.w * v1.w);
            s += ((double)v2.x + (double)v2.y) + ((double)v2.z + (double)v2.w);
            ss += ((double)v2.x * v2.x + (double)v2.y * v2.y) + ((double)v2.z * v2.z + (double)v2.w * v2.w);
            s += ((double)v3.x + (double)v3.y) + ((double)v3.z + (double)v3.w);
            ss += ((double)v3.x * v3.x + (double)v3.y * v3.y) + ((double)v3.z * v3.z + (double)v3.w * v3.w);
        }
        for (; i < n4; i += 256) {
            float4 v = p4[i];
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) {
            float v = plane[i];
            s += v; ss += (double)v * v;
        }
    }
    __shared__ double red[2][4];
    s = wave_sum(s); ss = wave_sum(ss);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s; red[1][w] = ss; }
    __syncthreads();
    if (threadIdx.x == 0)
        part[blockIdx.x] = make_double2((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
}

Status launch_gn_stats(hipStream_t s, CatSrc src, int B, int HW, double2* part) {
    int C = src.ca + src.cb;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * C), dim3(256), 0, s, src, HW, part);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// One workgroup per (image, group): folds the group's per-channel statistics in fp64 - conv6 epilogue slots
// (float2 per 64-pixel wave) or gn_stats partials (double2 per plane), per source tensor of the virtual concat - in a
// fixed order (thread-strided, then the usual wave/LDS tree), and writes the per-channel affine parameters.
__global__ __launch_bounds__(256) void gn_prm_kernel(GnStatSrc sa, GnStatSrc sb, int HW, const float* gamma, const float* beta,
                                                     const float* film, int film_stride, int film_off, int C, float act, float4* prm,
                                                     const StepDev* fstep, int frows, float2* stats_out) {
    const int n = blockIdx.x >> 5, g = blockIdx.x & 31;
    // hoisted FiLM (dpir_run_loop): the projections of ALL steps were evaluated before the loop; row = current step, shared by the batch
    if (film && fstep) film += (size_t)fstep->i * frows;
    const int cg = C / 32;
    double S = 0.0, SS = 0.0;
    for (int k = 0; k < cg; ++k) {
        const int c = g * cg + k;
        const bool in_a = c < sa.c;
        const GnStatSrc& src = in_a ? sa : sb;
        const int cl = in_a ? c : c - sa.c;
        if (src.slots) {
            // four slots of a thread are requested together (clamped index, no branch around the loads) and then added in the SAME order as a one-at-a-time
            // loop would add them: the sums are bit-identical, the serialised memory round trips per channel drop from nslots / 256 to nslots / 1024
            const float2* sp = src.slots + ((size_t)n * src.c + cl) * src.nslots;
            for (int base = threadIdx.x; base < src.nslots; base += 1024) {
                float2 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = sp[min(base + 256 * j, src.nslots - 1)];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (base + 256 * j < src.nslots) { S += (double)v[j].x; SS += (double)v[j].y; }
            }
        } else if (threadIdx.x == 0) {
            double2 v = src.part[(size_t)n * src.c + cl];
            S += v.x; SS += v.y;
        }
    }
    __shared__ double red[2][4];
    __shared__ float2 st_sh;
    S = wave_sum(S); SS = wave_sum(SS);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = S; red[1][w] = SS; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        double ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        double cnt = (double)cg * HW;
        double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        if (var < 0) var = 0;
        st_sh = make_float2((float)mean, (float)(1.0 / sqrt(var + 1e-5)));
        if (stats_out) stats_out[blockIdx.x] = st_sh;          // {mean, rstd} per (image, group): the backward pass needs rstd alone
    }
    __syncthreads();
    const float2 st = st_sh;
    for (int k = threadIdx.x; k < cg; k += 256) {
        const int c = g * cg + k;
        float a = st.y * gamma[c];
        float b = beta[c];
        if (film) {   // h = GN(h) * (1 + scale) + shift   (unet.py:250-251)
            float sc = 1.0f + film[(size_t)n * film_stride + film_off + c];
            float sh = film[(size_t)n * film_stride + film_off + C + c];
            a = a * sc;
            b = b * sc + sh;
        }
        prm[(size_t)n * C + c] = make_float4(st.x, a, b, act);
    }
}

Status launch_gn_prm(hipStream_t s, GnStatSrc sa, GnStatSrc sb, int HW, const float* gamma, const float* beta,
                     const float* film, int film_stride, int film_off, int B, int C, bool silu, float4* prm,
                     const StepDev* fstep, int frows, float2* stats_out) {
    if (C % 32 || sa.c + sb.c != C) return invalid("gn_prm: channel bookkeeping");
    hipLaunchKernelGGL(gn_prm_kernel, dim3(B * 32), dim3(256), 0, s, sa, sb, HW, gamma, beta, film, film_stride,
                       film_off, C, silu ? 1.0f : 0.0f, prm, fstep, frows, stats_out);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// out[n, r] = act_out( dot(W[r,:], in[n,:]) + bias[r] (+ extra[idx[n], r]) ); one wave per (r, n)
template <int ACT_OUT>
__global__ __launch_bounds__(256) void rows_gemv_kernel(const float* W, const float* bias, const float* x, int R, int K,
                                                         float* out, const float* extra, const int* extra_idx) {
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    int n = blockIdx.y;
    int lane = threadIdx.x & 63;
    if (r >= R) return;
    const float* wr = W + (size_t)r * K;
    const float* xn = x + (size_t)n * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(wr[k], xn[k], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
        float v = acc + bias[r];
        if (extra) v += extra[(size_t)extra_idx[n] * R + r];
        if (ACT_OUT) v = v / (1.0f + expf(-v));
        out[(size_t)n * R + r] = v;
    }
}

// out[n, r] = dot(W[r,:], x[n,:]) + bias[r] for ALL n: one wave per weight row (the row is read from HBM/L2 once, kept in
// registers for K <= 1024, and dotted with every image's vector, which stays in L1/L2)
__global__ __launch_bounds__(256) void rows_gemv_all_kernel(const float* W, const float* bias, const float* x, int B, int R, int K, float* out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    const float* wr = W + (size_t)r * K;
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = lane + 64 * i < K ? wr[lane + 64 * i] : 0.f;
    const float bv = bias[r];
    for (int n = 0; n < B; ++n) {
        const float* xn = x + (size_t)n * K;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (lane + 64 * i < K) acc = fmaf(w[i], xn[lane + 64 * i], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) out[(size_t)n * R + r] = acc + bv;
    }
}

Status launch_rows_gemv(hipStream_t s, const float* W, const float* bias, const float* x, int B, int R, int K, float* out) {
    if (K <= 1024) {
        // same per-lane summation order as rows_gemv_kernel (k = lane, lane + 64, ...), so results are bit-identical
        hipLaunchKernelGGL(rows_gemv_all_kernel, dim3((R + 3) / 4), dim3(256), 0, s, W, bias, x, B, R, K, out);
    } else {
        hipLaunchKernelGGL(rows_gemv_kernel<0>, dim3((R + 3) / 4, B), dim3(256), 0, s, W, bias, x, R, K, out,
                           (const float*)nullptr, (const int*)nullptr);
    }
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// e[n, :] = [cos(t f) | sin(t f)], f_i = exp(-ln(10000) i / half)      (nn.py:103-121)
// (freqs are tabulated on the host, correctly rounded, at load time)
__global__ void timestep_embedding_kernel(const int* t, const float* freqs, int mc, float* e) {
    int n = blockIdx.x;
    int half = mc / 2;
    float tv = (float)t[n];
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        float a = tv * freqs[i];
        e[(size_t)n * mc + i] = cosf(a);
        e[(size_t)n * mc + half + i] = sinf(a);
    }
    if ((mc & 1) && threadIdx.x == 0) e[(size_t)n * mc + mc - 1] = 0.f;
}

Status launch_time_embed(hipStream_t s, const int* t_dev, const int* y_dev, const float* freqs, const float* w0, const float* b0,
                         const float* w2, const float* b2, const float* label_emb, int B, int mc, float* tmp, float* semb) {
    int ted = 4 * mc;
    float* e = tmp;                 // [B, mc]
    float* h = tmp + (size_t)B * mc;  // [B, ted]
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(B), dim3(64), 0, s, t_dev, freqs, mc, e);
    // h = silu(W0 e + b0)
    hipLaunchKernelGGL(rows_gemv_kernel<1>, dim3((ted + 3) / 4, B), dim3(256), 0, s, w0, b0, e, ted, mc, h,
                       (const float*)nullptr, (const int*)nullptr);
    // semb = silu(W2 h + b2 (+ label_emb[y]))  -- every consumer of emb applies SiLU first (unet.py:200)
    hipLaunchKernelGGL(rows_gemv_kernel<1>, dim3((ted + 3) / 4, B), dim3(256), 0, s, w2, b2, h, ted, ted, semb,
                       label_emb, y_dev);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
