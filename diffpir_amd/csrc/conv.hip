// Implicit-GEMM convolution (3x3 pad 1 / 1x1) on the gfx950 matrix cores, exact fp32:
// v_mfma_f32_32x32x2_f32 (bit-for-bit a k-ordered fmaf chain, 157.3 TF/s peak).
//
// Replaces the ATen conv2d / conv1d(k=1) calls issued by guided_diffusion/unet.py:185,211,222,
// 286,294,482,615 -- 99 % of the reference's step time -- and folds into the
// same kernel everything the reference runs as separate elementwise passes around them:
//   prologue : GroupNorm affine (+FiLM scale/shift) + SiLU (nn.py:17-19, unet.py:184,200,251),
//              avg_pool2d 2x2 / nearest x2 resampling (unet.py:107,136), channel concat (unet.py:660)
//   epilogue : bias, residual add (unet.py:256, 305) incl. the resampled identity skip.
//
// GEMM view (per image): D[co][p] = sum_{ci,tap} Wt[co][ci,tap] * X[ci,tap][p]
//   A operand = weights  (rows i = output channel), B operand = activations (cols j = pixel),
//   so that in the accumulator layout the 32 lanes of a half-wave hold 32 consecutive pixels of
//   one output channel -> 128-byte coalesced NCHW stores.
// Block = 4 waves, tile = BCO output channels x 128 pixels (a TI x TH x TW patch), K loop over
// chunks of KC input channels: weights chunk [taps][KC][BCO] and the activation halo patch
// [KC][TI][TH+2][TW+2] are staged in LDS once per chunk and reused by all 9 taps.
#include "common.h"

// Ablation switches exist only in -DDPIR_ABLATE builds; the product kernels contain none of them.
#ifdef DPIR_ABLATE
#define ABL(bit) ((p.dbg & (bit)) != 0)
#else
#define ABL(bit) false
#endif

namespace dpir {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct ConvK {
    const float* sa; const float* sb; int ca, cb; int Hs, Ws; int mode; const float4* prm;
    const float* w; const float* bias; float* out; const float* res; int res_mode;
    int B, Cin, Cout, CoutP, H, W;
    int ltw, lth;          // log2 of pixel-tile width / height
    int ti;                // images per pixel tile (<= 8); TW*TH*TI <= 128
    int tiles_x, tiles_y;  // tiles per image
    int n_ptiles;          // total pixel tiles
    int n_co_blocks;
    int chs;               // LDS activation channel stride (floats)
    int ksplit;            // >1: split the Cin chunks over ksplit workgroups, raw partial sums go to `partial`
    int chunks_per_split;
    float* partial;        // [ksplit][B*Cout*H*W]
    size_t partial_cap;    // host-side: floats available in `partial`
    int dbg;               // ablation bits for dpir_debug_conv_bench (0 in production): 1 no MFMA, 2 no prologue
                           // transform, 4 no global loads, 8 no LDS stores, 16 no epilogue stores
};

// SiLU with the hardware transcendental units: v * rcp(1 + exp2(-v*log2e)).  v_exp_f32 / v_rcp_f32 are each good to
// ~1 ulp, so the result is within ~3e-7 relative of the correctly rounded x*sigmoid(x) -- an order of magnitude below
// the layer-level summation-order noise (3e-6) -- at a quarter of the instruction count of expf() + IEEE division.
__device__ __forceinline__ float silu_f(float v) {
    float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int KS, int KC, int WAVES_CO, int WCO, int WPX, int MODE>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvK p) {
    constexpr int TAPS = KS * KS;
    constexpr int BCO = WAVES_CO * WCO * 32;
    constexpr int NP = (KS == 3) ? 2 : 1;   // activation-tile positions per thread (3x3 patch <= 512, 1x1 patch <= 128)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds_w = smem;                       // [TAPS][KC][BCO]
    float4* lds_prm = reinterpret_cast<float4*>(smem + TAPS * KC * BCO);   // [2][KC][8] GroupNorm/FiLM params
    float* lds_x = smem + TAPS * KC * BCO + 2 * KC * 8 * 4;                   // [KC][chs]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int wave_co = wave % WAVES_CO;
    const int wave_px = wave / WAVES_CO;

    int bid = blockIdx.x;
    const int split = bid % p.ksplit;
    bid /= p.ksplit;
    const int co_blk = bid % p.n_co_blocks;
    const int ptile = bid / p.n_co_blocks;
    const int co0 = co_blk * BCO;
    const int TW = 1 << p.ltw, TH = 1 << p.lth;
    const int TI = p.ti;
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int img_grp = ptile / tiles_per_img;
    const int trem = ptile - img_grp * tiles_per_img;
    const int ty0 = (trem / p.tiles_x) * TH;
    const int tx0 = (trem % p.tiles_x) * TW;
    const int n0 = img_grp * TI;
    const int LW = (KS == 3) ? TW + 2 : TW;
    const int LH = (KS == 3) ? TH + 2 : TH;
    const int HsWs = p.Hs * p.Ws;
    const int C = p.ca + p.cb;

    // ---- per-thread staging positions (chunk invariant)
    int pos_lds[NP], pos_src[NP], pos_n[NP], pos_ti[NP];
    bool pos_ok[NP];
    const float* pos_pa[NP];   // &A[n, 0, src] and &B[n, 0, src]: channel c adds c*HsWs (no 64-bit multiplies in the loop)
    const float* pos_pb[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        int r = tid + q * 256;
        pos_lds[q] = r;
        bool in = r < p.chs;
        int ti = r / (LH * LW);
        int rr = r - ti * (LH * LW);
        int hy = rr / LW, hx = rr - hy * LW;
        int gy = ty0 + hy - (KS == 3 ? 1 : 0);
        int gx = tx0 + hx - (KS == 3 ? 1 : 0);
        int n = n0 + ti;
        bool ok = in && ti < TI && n < p.B && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        int so;
        if (MODE == 0) so = gy * p.Ws + gx;
        else if (MODE == 1) so = (gy >> 1) * p.Ws + (gx >> 1);
        else so = (gy * 2) * p.Ws + gx * 2;
        pos_src[q] = ok ? so : 0;
        pos_n[q] = ok ? n : 0;
        pos_ti[q] = ok ? ti : 0;
        pos_ok[q] = ok;
        if (!in) pos_lds[q] = -1;
        pos_pa[q] = p.sa + (size_t)pos_n[q] * p.ca * HsWs + pos_src[q];
        pos_pb[q] = p.sb ? p.sb + (size_t)pos_n[q] * p.cb * HsWs + pos_src[q] : p.sa;
    }

    // ---- per-lane B-operand (pixel) offsets inside the LDS patch
    int boff[WPX];
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        int pp = (wave_px * WPX + j) * 32 + l31;
        int px = pp & (TW - 1);
        int py = (pp >> p.ltw) & (TH - 1);
        int ti = pp >> (p.ltw + p.lth);
        if (ti >= TI) ti = 0;   // padding lanes of a short tile: any valid address, masked at the store
        boff[j] = ti * (LH * LW) + py * LW + px + half * p.chs;
    }
    const int aoff = half * BCO + wave_co * WCO * 32 + l31;

    floatx16 acc[WCO][WPX];
#pragma unroll
    for (int i = 0; i < WCO; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int c_begin = split * p.chunks_per_split * KC;
    const int c_end = min(p.Cin, c_begin + p.chunks_per_split * KC);
    if (p.prm && tid < KC * 8) {   // parameters of the first chunk (visible after the loop-top barrier)
        int k = tid >> 3, ti = tid & 7;
        int c = c_begin + k, n = n0 + ti;
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < p.Cin && ti < TI && n < p.B) m = p.prm[(size_t)n * C + c];
        lds_prm[k * 8 + ti] = m;
    }
    // ---- software pipeline: the NEXT chunk's global loads are issued before the MFMA phase of the current
    // chunk and land in registers while the matrix pipe is busy (weights: NWV float4, activations KC x NP)
    constexpr int NV = TAPS * KC * BCO / 4;
    constexpr int NWV = (NV + 255) / 256;
    float4 wreg[NWV];
    float vals[KC][NP];
    auto load_chunk = [&](int c0) {
        if (ABL(4)) {
#pragma unroll
            for (int u = 0; u < NWV; ++u) wreg[u] = make_float4(0.5f, 0.25f, 0.125f, 1.f);
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int q = 0; q < NP; ++q) vals[k][q] = 0.3f;
            return;
        }
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            int v = tid + u * 256;
            int co4 = v % (BCO / 4);
            int t2 = v / (BCO / 4);
            int tap = t2 % TAPS;
            int k = t2 / TAPS;
            int c = c0 + k;
            int co = co0 + co4 * 4;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < NV && c < p.Cin && co < p.CoutP)
                val = *reinterpret_cast<const float4*>(p.w + ((size_t)c * TAPS + tap) * p.CoutP + co);
            wreg[u] = val;
        }
        if (MODE != 2) {
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                int c = c0 + k;
                bool cok = c < p.Cin;
                const bool in_a = c < p.ca;
                const int coff = (in_a ? c : c - p.ca) * HsWs;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    bool ok = cok && pos_ok[q];
                    const float* src = (in_a ? pos_pa[q] : pos_pb[q]) + coff;
                    vals[k][q] = ok ? *src : 0.f;
                }
            }
        }
    };
    load_chunk(c_begin);
    for (int c0 = c_begin; c0 < c_end; c0 += KC) {
        __syncthreads();
        // ---- registers -> LDS: weights [TAPS][KC][BCO]
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            int v = tid + u * 256;
            if (v < NV && !ABL(8)) {
                int co4 = v % (BCO / 4);
                int t2 = v / (BCO / 4);
                int tap = t2 % TAPS;
                int k = t2 / TAPS;
                *reinterpret_cast<float4*>(lds_w + (tap * KC + k) * BCO + co4 * 4) = wreg[u];
            }
        }
        // ---- prefetch the NEXT chunk's GroupNorm/FiLM parameters into the other LDS buffer
        const int pbuf = ((c0 - c_begin) / KC) & 1;
        if (p.prm && tid < KC * 8) {
            int k = tid >> 3, ti = tid & 7;
            int c = c0 + KC + k, n = n0 + ti;
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < p.Cin && ti < TI && n < p.B) m = p.prm[(size_t)n * C + c];
            lds_prm[((pbuf ^ 1) * KC + k) * 8 + ti] = m;
        }
        // ---- registers -> LDS: activation patch with the fused prologue
        if (MODE != 2) {
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                bool cok = (c0 + k) < p.Cin;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (pos_lds[q] < 0) continue;
                    float v = vals[k][q];
                    if (p.prm && !ABL(2)) {
                        float4 m = lds_prm[(pbuf * KC + k) * 8 + pos_ti[q]];
                        v = (v - m.x) * m.y + m.z;
                        if (m.w != 0.f) v = silu_f(v);
                    }
                    if (!ABL(8)) lds_x[k * p.chs + pos_lds[q]] = (cok && pos_ok[q]) ? v : 0.f;
                }
            }
        } else {
#pragma unroll 2
            for (int k = 0; k < KC; ++k) {
                int c = c0 + k;
                bool cok = c < p.Cin;
                const float* plane; int cc, cs;
                if (c < p.ca) { plane = p.sa; cc = c; cs = p.ca; } else { plane = p.sb; cc = c - p.ca; cs = p.cb; }
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (pos_lds[q] < 0) continue;
                    bool ok = cok && pos_ok[q];
                    float v = 0.f;
                    if (ok) {
                        const float* src = plane + ((size_t)pos_n[q] * cs + cc) * HsWs + pos_src[q];
                        float v0 = src[0], v1 = src[1], v2 = src[p.Ws], v3 = src[p.Ws + 1];
                        if (p.prm) {
                            float4 m = lds_prm[(pbuf * KC + k) * 8 + pos_ti[q]];
                            v0 = (v0 - m.x) * m.y + m.z; v1 = (v1 - m.x) * m.y + m.z;
                            v2 = (v2 - m.x) * m.y + m.z; v3 = (v3 - m.x) * m.y + m.z;
                            if (m.w != 0.f) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
                        }
                        v = ((v0 + v1) + (v2 + v3)) * 0.25f;
                    }
                    lds_x[k * p.chs + pos_lds[q]] = v;
                }
            }
        }
        __syncthreads();
        if (c0 + KC < c_end) load_chunk(c0 + KC);
        // ---- MFMA over taps x channel pairs.  Operands are read from LDS one STAGE (2 k-steps = 8 MFMAs per
        // wave, ~500 cycles) ahead into a second register set, so ds_read latency never sits in front of an MFMA
        // (the compiler's own schedule re-used one A register pair and waited lgkmcnt(0) every 4 MFMAs).
        if (!ABL(1)) {
            constexpr int KSTEPS = KC / 2;
            constexpr int NSTEP = TAPS * KSTEPS;
            constexpr int SG = 2;                       // k-steps per stage (8 MFMAs per wave, ~500 cycles of cover)
            constexpr int NSTAGE = NSTEP / SG;
            static_assert(NSTEP % SG == 0, "stage size must divide the step count");
            float a_op[2][SG][WCO], b_op[2][SG][WPX];
            auto load_stage = [&](int stage, int buf) {
#pragma unroll
                for (int q = 0; q < SG; ++q) {
                    const int f = stage * SG + q;
                    const int tap = f / KSTEPS, kk = f % KSTEPS;
                    const int toff = (KS == 3) ? (tap / 3) * LW + (tap % 3) : 0;
#pragma unroll
                    for (int i = 0; i < WCO; ++i) a_op[buf][q][i] = lds_w[(tap * KC + 2 * kk) * BCO + aoff + i * 32];
#pragma unroll
                    for (int j = 0; j < WPX; ++j) b_op[buf][q][j] = lds_x[(2 * kk) * p.chs + boff[j] + toff];
                }
            };
            load_stage(0, 0);
#pragma unroll
            for (int st = 0; st < NSTAGE; ++st) {
                if (st + 1 < NSTAGE) load_stage(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < SG; ++q)
#pragma unroll
                    for (int i = 0; i < WCO; ++i)
#pragma unroll
                        for (int j = 0; j < WPX; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_op[st & 1][q][i], b_op[st & 1][q][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: bias + residual, coalesced NCHW stores (32 consecutive pixels per half-wave)
    const int HW = p.H * p.W;
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        int pp = (wave_px * WPX + j) * 32 + l31;
        int px = pp & (TW - 1);
        int py = (pp >> p.ltw) & (TH - 1);
        int ti = pp >> (p.ltw + p.lth);
        int n = n0 + ti, y = ty0 + py, x = tx0 + px;
        bool pok = ti < TI && n < p.B && y < p.H && x < p.W;
#pragma unroll
        for (int i = 0; i < WCO; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = co0 + (wave_co * WCO + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (ABL(16)) {
                    if (acc[i][j][r] == 1.2345e33f) p.out[0] = 1.f;   // keep the accumulators live
                } else if (pok && co < p.Cout && p.ksplit > 1) {
                    p.partial[(size_t)split * ((size_t)p.B * p.Cout * HW) + ((size_t)n * p.Cout + co) * HW + y * p.W + x] = acc[i][j][r];
                } else if (pok && co < p.Cout) {
                    float v = acc[i][j][r] + p.bias[co];
                    if (p.res) {
                        float rv;
                        if (p.res_mode == 0) {
                            rv = p.res[((size_t)n * p.Cout + co) * HW + y * p.W + x];
                        } else if (p.res_mode == 1) {
                            int Hr = p.H >> 1, Wr = p.W >> 1;
                            rv = p.res[((size_t)n * p.Cout + co) * (Hr * Wr) + (y >> 1) * Wr + (x >> 1)];
                        } else {
                            int Wr = p.W * 2;
                            const float* rp = p.res + ((size_t)n * p.Cout + co) * (4 * HW) + (2 * y) * Wr + 2 * x;
                            rv = ((rp[0] + rp[1]) + (rp[Wr] + rp[Wr + 1])) * 0.25f;
                        }
                        v = rv + v;
                    }
                    p.out[((size_t)n * p.Cout + co) * HW + y * p.W + x] = v;
                }
            }
        }
    }
}

// deterministic split-K combine: out = sum_s partial[s] (in order) + bias + residual
__global__ void conv_splitk_reduce_kernel(const float* partial, int ksplit, const float* bias, const float* res, int res_mode,
                                          float* out, int Cout, int H, int W, size_t total) {
    const int HW = H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < ksplit; ++s) v += partial[(size_t)s * total + i];
        size_t nc = i / HW;
        int co = (int)(nc % Cout);
        v += bias[co];
        if (res) {
            int r = (int)(i - nc * HW);
            int y = r / W, x = r - y * W;
            float rv;
            if (res_mode == 0) rv = res[i];
            else if (res_mode == 1) { int Hr = H >> 1, Wr = W >> 1; rv = res[nc * (size_t)(Hr * Wr) + (y >> 1) * Wr + (x >> 1)]; }
            else { int Wr = W * 2; const float* rp = res + nc * (size_t)(4 * HW) + (2 * y) * Wr + 2 * x; rv = ((rp[0] + rp[1]) + (rp[Wr] + rp[Wr + 1])) * 0.25f; }
            v = rv + v;
        }
        out[i] = v;
    }
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

template <int KS, int KC, int WAVES_CO, int WCO, int WPX, int MODE>
static Status launch_mode(hipStream_t s, ConvK k) {
    constexpr int TAPS = KS * KS;
    constexpr int BCO = WAVES_CO * WCO * 32;
    k.n_co_blocks = (k.Cout + BCO - 1) / BCO;
    size_t lds = (size_t)(TAPS * KC * BCO + 2 * KC * 8 * 4 + KC * k.chs) * sizeof(float);
    auto fn = conv_mfma_kernel<KS, KC, WAVES_CO, WCO, WPX, MODE>;
    static LdsAttrOnce attr_set;
    DPIR_HIP(attr_set.set(reinterpret_cast<const void*>(fn), 160 * 1024));
    // split-K for launches that cannot fill 256 CUs x 2 workgroups (low-resolution layers): partial slabs +
    // an ordered reduce, so results stay bitwise reproducible
    const int chunks = (k.Cin + KC - 1) / KC;
    const int blocks = k.n_ptiles * k.n_co_blocks;
    int S = 1;
    if (k.partial && blocks < 384) {
        S = (512 + blocks - 1) / blocks;
        if (S > chunks / 4) S = chunks / 4;
        if (S > 16) S = 16;
        if (S < 1) S = 1;
        size_t need = (size_t)S * k.B * k.Cout * k.H * k.W;
        if (need > k.partial_cap) S = 1;
    }
    k.ksplit = S;
    k.chunks_per_split = (chunks + S - 1) / S;
    if (S == 1) k.partial = nullptr;
    dim3 grid((unsigned)(blocks * S));
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, s, k);
    if (S > 1) {
        size_t total = (size_t)k.B * k.Cout * k.H * k.W;
        unsigned nb = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(nb), dim3(256), 0, s, k.partial, S, k.bias, k.res, k.res_mode, k.out,
                           k.Cout, k.H, k.W, total);
    }
    DPIR_HIP(hipGetLastError());
    return Status{};
}

template <int KS, int KC, int WAVES_CO, int WCO, int WPX>
static Status launch_cfg(hipStream_t s, ConvK k) {
    if (k.mode == 0) return launch_mode<KS, KC, WAVES_CO, WCO, WPX, 0>(s, k);
    if (k.mode == 1) return launch_mode<KS, KC, WAVES_CO, WCO, WPX, 1>(s, k);
    return launch_mode<KS, KC, WAVES_CO, WCO, WPX, 2>(s, k);
}

Status launch_conv2(hipStream_t s, const ConvArgs& a);

Status launch_conv(hipStream_t s, const ConvArgs& a) {
    if (a.ks != 1 && a.ks != 3) return invalid("conv: ks must be 1 or 3");
    if (a.src.ca + a.src.cb != a.Cin) return invalid("conv: Cin mismatch");
    if (a.CoutP % 4 != 0 || a.CoutP < a.Cout) return invalid("conv: CoutP must be a multiple of 4 and >= Cout");
    ConvK k;
    k.sa = a.src.a; k.sb = a.src.b; k.ca = a.src.ca; k.cb = a.src.cb; k.Hs = a.src.Hs; k.Ws = a.src.Ws;
    k.mode = a.src.mode; k.prm = a.src.prm;
    k.w = a.w; k.bias = a.bias; k.out = a.out; k.res = a.res; k.res_mode = a.res_mode;
    k.B = a.B; k.Cin = a.Cin; k.Cout = a.Cout; k.CoutP = a.CoutP; k.H = a.H; k.W = a.W;
    k.partial = a.partial; k.ksplit = 1; k.chunks_per_split = 0;
    k.partial_cap = a.partial_capacity;
    k.dbg = a.dbg;
    // expected source resolution for the resampling mode
    int eh = a.src.mode == 1 ? a.H / 2 : (a.src.mode == 2 ? a.H * 2 : a.H);
    int ew = a.src.mode == 1 ? a.W / 2 : (a.src.mode == 2 ? a.W * 2 : a.W);
    if (eh != a.src.Hs || ew != a.src.Ws) return invalid("conv: source resolution does not match mode");
    if (a.src.mode == 1 && ((a.H | a.W) & 1)) return invalid("conv: up mode needs even output size");
    // generation-2 kernel (conv2.hip) for plain / up-sampled sources; the pooled-source variant stays on v1
    if (a.src.mode != 2 && a.CoutP % 64 == 0) return launch_conv2(s, a);
    // pixel tile: TW x TH x TI = 128
    int tw = a.W >= 32 ? 32 : (a.W >= 16 ? 16 : (a.W >= 8 ? 8 : 4));
    int th = 128 / tw;
    int hp2 = 1 << ilog2(a.H);
    if (th > hp2) th = hp2;
    int ti = 128 / (tw * th);
    if (ti > 8) ti = 8;
    k.ti = ti;
    k.ltw = ilog2(tw); k.lth = ilog2(th);
    k.tiles_x = (a.W + tw - 1) / tw;
    k.tiles_y = (a.H + th - 1) / th;
    k.n_ptiles = k.tiles_x * k.tiles_y * ((a.B + ti - 1) / ti);
    k.chs = a.ks == 3 ? ti * (th + 2) * (tw + 2) : ti * th * tw;
    if (k.chs > 512) return invalid("conv: activation patch too large");
    if (a.ks == 3) {
        if (a.Cout > 64) return launch_cfg<3, 8, 2, 2, 2>(s, k);
        if (a.Cout > 32) return launch_cfg<3, 8, 1, 2, 1>(s, k);
        return launch_cfg<3, 8, 1, 1, 1>(s, k);
    } else {
        if (a.Cout > 64) return launch_cfg<1, 32, 2, 2, 2>(s, k);
        if (a.Cout > 32) return launch_cfg<1, 32, 1, 2, 1>(s, k);
        return launch_cfg<1, 32, 1, 1, 1>(s, k);
    }
}

}  // namespace dpir
