// Fused QKV attention on fp32 MFMA (v_mfma_f32_32x32x2_f32), flash-style online softmax.
//
// Replaces QKVAttentionLegacy.forward (guided_diffusion/unet.py:337-354): qkv [B, heads*3*ch, T] with
// the LEGACY order -- heads are split first, then q|k|v inside each head -- and the two-sided
// scaling (q*s)·(k*s), s = ch^-1/4, softmax in fp32.  T = 64 / 256 (FFHQ), up to 1024 (ImageNet).
//
// One wave owns 32 queries of one (image, head).  Scores are computed TRANSPOSED,
//   S^T[s][t] = sum_c K[c][s] Q[c][t]      (A = K tile from LDS, B = Q held in 32 VGPRs),
// so that a lane's accumulator column is one query t: the softmax row statistics are per-lane scalars
// (16 in-register values + one cross-half shuffle) and P^T is already in B-operand position for
//   O[c][t] += sum_s V[c][s] P^T[s][t]     (A = V tile from LDS, B = P^T straight from the accumulator),
// with the k index of the second product walking the accumulator's own row order (no data movement).
#include "common.h"

namespace dpir {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int HC = 64;        // head channels (num_head_channels=64, utils_model.py:361)
constexpr int KT = 32;        // keys per tile
constexpr int VS = KT + 1;    // padded V row stride (conflict-free column reads)

__global__ __launch_bounds__(256) void attention_kernel(const float* qkv, float* out, int C, int T, int n_heads,
                                                          int q_tiles_per_block, float scale) {
    __shared__ float lds_k[HC * KT];
    __shared__ float lds_v[HC * VS];
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y;
    const int b = bh / n_heads, h = bh - b * n_heads;
    const float* qb = qkv + ((size_t)b * 3 * C + (size_t)h * 3 * HC) * T;
    const float* kb = qb + (size_t)HC * T;
    const float* vb = kb + (size_t)HC * T;
    const int t0 = (blockIdx.x * q_tiles_per_block + wave) * 32;
    const int tq = t0 + l31;
    const bool q_ok = tq < T;

    // Q as B operand: lane (c = 2kk+half, t = l31).  The loads are unconditional (clamped index, value selected afterwards): as
    // `q_ok ? load : 0` every one of the 32 loads sat in its own predicated block with an `s_waitcnt vmcnt(0)` behind it -- 32
    // serialised HBM latencies before the first MFMA (tools/isa_audit.py, "after_load").
    float qreg[HC / 2];
    const int tqc = q_ok ? tq : T - 1;
#pragma unroll
    for (int kk = 0; kk < HC / 2; ++kk) qreg[kk] = qb[(size_t)(2 * kk + half) * T + tqc];
#pragma unroll
    for (int kk = 0; kk < HC / 2; ++kk) qreg[kk] = q_ok ? qreg[kk] * scale : 0.f;

    floatx16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    for (int s0 = 0; s0 < T; s0 += KT) {
        __syncthreads();
        for (int i = tid; i < HC * KT; i += nthr) {
            int c = i >> 5, s = i & 31;
            const bool ok = s0 + s < T;
            const int sc = ok ? s0 + s : T - 1;                  // unconditional loads, as above
            const float kv = kb[(size_t)c * T + sc], vv = vb[(size_t)c * T + sc];
            lds_k[c * KT + s] = ok ? kv * scale : 0.f;
            lds_v[c * VS + s] = ok ? vv : 0.f;
        }
        __syncthreads();
        floatx16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < HC / 2; ++kk)
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_k[(2 * kk + half) * KT + l31], qreg[kk], st, 0, 0, 0);
        // mask keys beyond T, tile max per query
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int s = s0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (s >= T) st[r] = -INFINITY;
            mt = fmaxf(mt, st[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        float m_new = fmaxf(m_run, mt);
        float alpha = expf(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = expf(st[r] - m_new); ps += st[r]; }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        // O[c][t] += V[c][s] * P^T[s][t], k-step r pairs s = sidx(r) (half 0) with sidx(r)+4 (half 1)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int sidx = (r & 3) + 8 * (r >> 2) + 4 * half;
            float pv = st[r];
            o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_v[l31 * VS + sidx], pv, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_v[(32 + l31) * VS + sidx], pv, o[1], 0, 0, 0);
        }
    }
    if (q_ok) {
        float inv = 1.0f / l_run;
        float* ob = out + ((size_t)b * C + (size_t)h * HC) * T;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                ob[(size_t)c * T + tq] = o[ct][r] * inv;
            }
    }
}

Status launch_attention(hipStream_t s, const float* qkv, float* out, int B, int C, int T, int head_ch) {
    if (head_ch != HC) return Status{DPIR_ERR_UNSUPPORTED, "attention: only num_head_channels=64 is built"};
    if (C % HC) return invalid("attention: channels not divisible by head channels");
    int n_heads = C / HC;
    int q_tiles = (T + 31) / 32;
    int qpb = q_tiles < 4 ? q_tiles : 4;
    int blocks_x = (q_tiles + qpb - 1) / qpb;
    float scale = (float)(1.0 / sqrt(sqrt((double)HC)));
    hipLaunchKernelGGL(attention_kernel, dim3(blocks_x, B * n_heads), dim3(64 * qpb), 0, s, qkv, out, C, T, n_heads, qpb, scale);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
