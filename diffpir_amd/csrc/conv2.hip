// conv2: second-generation implicit-GEMM convolution (3x3 pad 1 / 1x1), exact fp32 MFMA
// (v_mfma_f32_32x32x2_f32).  Same math, fusions and operand orientation as conv.hip (see its header); what
// changes is the staging pipeline, after the round-1 ablation (profiles/r01: the v1 kernel issues ~2000
// non-MFMA instructions per wave per K chunk, which do not fit in the shadow of its 144 MFMAs):
//   * tile 64 output channels x 256 pixels (8 x 32 patch) instead of 128 x 128: the weight chunk per K step
//     halves (18 KB) and the halo overhead drops from 1.59x to 1.33x;
//   * EVERYTHING that comes from global memory arrives by LDS-DMA: weights (global_load_lds_dwordx4, 4 rows of 64
//     channels per instruction), the RAW activation patch (global_load_lds_dword, one element per lane;
//     out-of-image lanes are sourced from a zero page) and the GroupNorm/FiLM parameters (one private copy per
//     wave).  The K loop contains no VGPR-destination global load: no per-element address arithmetic, no staging
//     registers, no compiler-inserted vmcnt(0) drains, no ds_write of weights;
//   * weights and patch are double-buffered in LDS (2 x 35 KB, 2 workgroups per CU): ONE barrier per K chunk.
//     Per chunk a wave issues its DMAs for chunk c+1, runs the 144 MFMAs of chunk c, waits for its own DMAs and
//     transforms IN PLACE (GroupNorm affine + FiLM + SiLU) exactly the patch elements its own lanes fetched.
// Weight layout expected here: [CinP][taps][CoutP] with CinP a multiple of 16 (zero rows) and CoutP a multiple
// of 64 (zero columns), so the weight DMA needs no bounds check.
#include "common.h"

// Ablation switches exist only in -DDPIR_ABLATE builds; the product kernels contain none of them.
#ifdef DPIR_ABLATE
#define ABL(bit) ((p.dbg & (bit)) != 0)
#else
#define ABL(bit) false
#endif

namespace dpir {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct Conv2K {
    const float* sa; const float* sb; int ca, cb; int Hs, Ws; const float4* prm;
    const float* w; const float* bias; float* out; const float* res; int res_mode;
    int B, Cin, Cout, CoutP, H, W;
    int ltw, lth, ti;
    int tiles_x, tiles_y, n_ptiles, n_co_blocks;
    int chs;               // patch floats per channel (<= NP*256)
    int ksplit, chunks_per_split;
    float* partial;
    const float* zeros;    // >= 64 B of zeros in global memory (DMA source of padding lanes)
    int dbg;
};

__device__ __forceinline__ float silu2_f(float v) {
    float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

#define GLDS(src, dst, bytes) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), bytes, 0, 0)

template <int KS, int KC, int MODE>   // MODE 0 plain, 1 nearest-up source
__global__ __launch_bounds__(256, 2) void conv2_mfma_kernel(Conv2K p) {
    constexpr int TAPS = KS * KS;
    constexpr int BCO = 64;
    constexpr int WCO = 2, WPX = 2;
    constexpr int NP = (KS == 3) ? 2 : 1;          // patch positions per thread
    constexpr int XS = NP * 256;                   // LDS channel stride of the patch (DMA writes whole 64-lane pieces)
    constexpr int WROWS = KC * TAPS;               // weight rows (of 64 floats) per chunk
    constexpr int WCHUNK = WROWS * BCO;            // floats
    constexpr int NDMA = WCHUNK * 4 / 1024;        // 1 KiB LDS-DMA pieces per chunk
    constexpr int PRM = KC * 8;                    // parameter entries (float4) per chunk
    static_assert((WCHUNK * 4) % 1024 == 0, "weight chunk must be whole DMA pieces");
    static_assert(PRM % 64 == 0, "parameter table must be whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds_w = smem;                                              // [2][WROWS][64]
    float4* lds_prm = reinterpret_cast<float4*>(smem + 2 * WCHUNK);    // [2][4 waves][PRM]
    float* lds_x = smem + 2 * WCHUNK + 2 * 4 * PRM * 4;               // [2][KC][XS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    int bid = blockIdx.x;
    const int split = bid % p.ksplit;
    bid /= p.ksplit;
    const int co_blk = bid % p.n_co_blocks;
    const int ptile = bid / p.n_co_blocks;
    const int co0 = co_blk * BCO;
    const int TW = 1 << p.ltw, TH = 1 << p.lth, TI = p.ti;
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

    // ---- per-thread patch positions (chunk invariant): DMA source pointers and their channel stride (0 for padding
    // lanes, which read the zero page)
    int pos_ti[NP], pos_cs[NP];
    bool pos_ok[NP];
    const float* pos_pa[NP];
    const float* pos_pb[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        int r = tid + q * 256;
        bool in = r < p.chs;
        int ti = r / (LH * LW);
        int rr = r - ti * (LH * LW);
        int hy = rr / LW, hx = rr - hy * LW;
        int gy = ty0 + hy - (KS == 3 ? 1 : 0);
        int gx = tx0 + hx - (KS == 3 ? 1 : 0);
        int n = n0 + ti;
        bool ok = in && ti < TI && n < p.B && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        int so = (MODE == 0) ? gy * p.Ws + gx : (gy >> 1) * p.Ws + (gx >> 1);
        pos_ti[q] = ok ? ti : 0;
        pos_ok[q] = ok;
        pos_cs[q] = ok ? HsWs : 0;
        pos_pa[q] = ok ? p.sa + (size_t)n * p.ca * HsWs + so : p.zeros;
        pos_pb[q] = (ok && p.sb) ? p.sb + (size_t)n * p.cb * HsWs + so : p.zeros;
    }

    // ---- per-lane MFMA operand offsets
    int boff[WPX];
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        int pp = (wave * WPX + j) * 32 + l31;
        int px = pp & (TW - 1);
        int py = (pp >> p.ltw) & (TH - 1);
        int ti = pp >> (p.ltw + p.lth);
        if (ti >= TI) ti = 0;
        boff[j] = ti * (LH * LW) + py * LW + px + half * XS;
    }
    const int aoff = half * TAPS * BCO + l31;      // lds_w[(k*TAPS + tap)*64 + co]

    floatx16 acc[WCO][WPX];
#pragma unroll
    for (int i = 0; i < WCO; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int c_begin = split * p.chunks_per_split * KC;
    const int c_end = min(p.Cin, c_begin + p.chunks_per_split * KC);

    // all DMAs of one chunk (starting at input channel c0) into LDS buffer `buf`
    auto issue_dma = [&](int c0, int buf) {
        if (ABL(4)) return;
        // weights: piece = 4 rows x 256 B, pieces round-robin over the 4 waves
#pragma unroll
        for (int u = 0; u < (NDMA + 3) / 4; ++u) {
            int piece = wave + u * 4;
            if (piece < NDMA) {
                int row = piece * 4 + (lane >> 4);     // row inside the chunk = k*TAPS + tap
                GLDS(p.w + ((size_t)c0 * TAPS + row) * p.CoutP + co0 + (lane & 15) * 4, lds_w + buf * WCHUNK + piece * 256, 16);
            }
        }
        // GroupNorm/FiLM parameters: a private copy per wave (read back after this wave's own vmcnt(0), no barrier)
        if (p.prm) {
#pragma unroll
            for (int u = 0; u < PRM / 64; ++u) {
                int ent = u * 64 + lane;
                int k = ent >> 3, ti = ent & 7;
                int c = min(c0 + k, p.Cin - 1), n = min(n0 + ti, p.B - 1);
                GLDS(p.prm + (size_t)n * C + c, lds_prm + (buf * 4 + wave) * PRM + u * 64, 16);
            }
        }
        // raw activation patch: lane = its own patch position, KC x NP dwords per lane
        float* xb = lds_x + buf * KC * XS + wave * 64;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            int c = min(c0 + k, p.Cin - 1);            // rows >= Cin are zeroed by the transform
            const bool in_a = c < p.ca;
            const int cc = in_a ? c : c - p.ca;
#pragma unroll
            for (int q = 0; q < NP; ++q)
                GLDS((in_a ? pos_pa[q] : pos_pb[q]) + (size_t)cc * pos_cs[q], xb + k * XS + q * 256, 4);
        }
    };
    // in-place prologue transform of the elements this thread's own DMAs delivered
    auto transform_acts = [&](int buf, int cbase) {
        if (ABL(8)) return;
        float* dst = lds_x + buf * KC * XS;
        const float4* prm = lds_prm + (buf * 4 + wave) * PRM;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const bool cok = (cbase + k) < p.Cin;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int idx = k * XS + tid + q * 256;
                float v = dst[idx];
                if (p.prm && !ABL(2)) {
                    float4 m = prm[k * 8 + pos_ti[q]];
                    v = (v - m.x) * m.y + m.z;
                    if (m.w != 0.f) v = silu2_f(v);
                }
                dst[idx] = (cok && pos_ok[q]) ? v : 0.f;
            }
        }
    };

    // ---- prologue: chunk 0 into buffer 0
    issue_dma(c_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    transform_acts(0, c_begin);

    int it = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += KC, ++it) {
        const int cur = it & 1;
        __syncthreads();                   // chunk c0 complete in LDS: every wave drained its DMAs and transformed its elements
        const bool more = c0 + KC < c_end;
        if (more) issue_dma(c0 + KC, cur ^ 1);

        if (!ABL(1)) {
            constexpr int KSTEPS = KC / 2;
            constexpr int NSTEP = TAPS * KSTEPS;
            constexpr int SG = 2;
            constexpr int NSTAGE = NSTEP / SG;
            static_assert(NSTEP % SG == 0, "stage size must divide the step count");
            const float* lw = lds_w + cur * WCHUNK;
            const float* lx = lds_x + cur * KC * XS;
            float a_op[2][SG][WCO], b_op[2][SG][WPX];
            auto load_stage = [&](int stage, int buf) {
#pragma unroll
                for (int q = 0; q < SG; ++q) {
                    const int f = stage * SG + q;
                    const int tap = f / KSTEPS, kk = f % KSTEPS;
                    const int toff = (KS == 3) ? (tap / 3) * LW + (tap % 3) : 0;
#pragma unroll
                    for (int i = 0; i < WCO; ++i) a_op[buf][q][i] = lw[(2 * kk * TAPS + tap) * BCO + aoff + i * 32];
#pragma unroll
                    for (int j = 0; j < WPX; ++j) b_op[buf][q][j] = lx[(2 * kk) * XS + boff[j] + toff];
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
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMAs of chunk c0+KC have landed
            transform_acts(cur ^ 1, c0 + KC);
        }
    }

    // ---- epilogue: bias + residual, 128-byte coalesced NCHW stores
    const int HW = p.H * p.W;
    const bool full_co = co0 + BCO <= p.Cout;      // workgroup-uniform: no per-element channel test on the fast path
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        int pp = (wave * WPX + j) * 32 + l31;
        int px = pp & (TW - 1);
        int py = (pp >> p.ltw) & (TH - 1);
        int ti = pp >> (p.ltw + p.lth);
        int n = n0 + ti, y = ty0 + py, x = tx0 + px;
        bool pok = ti < TI && n < p.B && y < p.H && x < p.W;
        if (ABL(16)) {
#pragma unroll
            for (int i = 0; i < WCO; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc[i][j][r] == 1.2345e33f) p.out[0] = 1.f;
            continue;
        }
        if (!pok) continue;
        const size_t pix = (size_t)y * p.W + x;
        if (p.ksplit > 1) {
            float* pb = p.partial + (size_t)split * ((size_t)p.B * p.Cout * HW) + (size_t)n * p.Cout * HW + pix;
#pragma unroll
            for (int i = 0; i < WCO; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (full_co || co < p.Cout) pb[(size_t)co * HW] = acc[i][j][r];
                }
            continue;
        }
        float* ob = p.out + (size_t)n * p.Cout * HW + pix;
        if (p.res && p.res_mode == 0) {
            const float* rb = p.res + (size_t)n * p.Cout * HW + pix;
            float rv[WCO][16];
#pragma unroll
            for (int i = 0; i < WCO; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    rv[i][r] = rb[(size_t)(full_co ? co : min(co, p.Cout - 1)) * HW];
                }
#pragma unroll
            for (int i = 0; i < WCO; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (full_co || co < p.Cout) ob[(size_t)co * HW] = rv[i][r] + (acc[i][j][r] + p.bias[co]);
                }
            continue;
        }
#pragma unroll
        for (int i = 0; i < WCO; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (full_co || co < p.Cout) {
                    float v = acc[i][j][r] + p.bias[co];
                    if (p.res) {
                        float rv;
                        if (p.res_mode == 1) {
                            int Hr = p.H >> 1, Wr = p.W >> 1;
                            rv = p.res[((size_t)n * p.Cout + co) * (Hr * Wr) + (y >> 1) * Wr + (x >> 1)];
                        } else {
                            int Wr = p.W * 2;
                            const float* rp = p.res + ((size_t)n * p.Cout + co) * (4 * HW) + (2 * y) * Wr + 2 * x;
                            rv = ((rp[0] + rp[1]) + (rp[Wr] + rp[Wr + 1])) * 0.25f;
                        }
                        v = rv + v;
                    }
                    ob[(size_t)co * HW] = v;
                }
            }
    }
}

__global__ void conv_splitk_reduce_kernel(const float* partial, int ksplit, const float* bias, const float* res, int res_mode,
                                          float* out, int Cout, int H, int W, size_t total);

static int ilog2c(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

const float* conv_zero_page() {
    static std::map<int, float*> pages;        // one read-only zero page per device
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    auto it = pages.find(dev);
    if (it != pages.end()) return it->second;
    float* z = nullptr;
    if (hipMalloc((void**)&z, 256) != hipSuccess) return nullptr;
    (void)hipMemset(z, 0, 256);
    pages[dev] = z;
    return z;
}

template <int KS, int KC, int MODE>
static Status launch2(hipStream_t s, Conv2K k, size_t partial_cap) {
    constexpr int TAPS = KS * KS;
    constexpr int NP = (KS == 3) ? 2 : 1;
    constexpr int WCHUNK = KC * TAPS * 64;
    size_t lds = (size_t)(2 * WCHUNK + 2 * 4 * KC * 8 * 4 + 2 * KC * NP * 256) * sizeof(float);
    auto fn = conv2_mfma_kernel<KS, KC, MODE>;
    static LdsAttrOnce attr_set;
    DPIR_HIP(attr_set.set(reinterpret_cast<const void*>(fn), 160 * 1024));
    const int chunks = (k.Cin + KC - 1) / KC;
    const int blocks = k.n_ptiles * k.n_co_blocks;
    int S = 1;
    if (k.partial && blocks < 384) {
        S = (512 + blocks - 1) / blocks;
        if (S > chunks / 4) S = chunks / 4;
        if (S > 16) S = 16;
        if (S < 1) S = 1;
        if ((size_t)S * k.B * k.Cout * k.H * k.W > partial_cap) S = 1;
    }
    k.ksplit = S;
    k.chunks_per_split = (chunks + S - 1) / S;
    if (S == 1) k.partial = nullptr;
    hipLaunchKernelGGL(fn, dim3((unsigned)(blocks * S)), dim3(256), lds, s, k);
    if (S > 1) {
        size_t total = (size_t)k.B * k.Cout * k.H * k.W;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k.partial, S, k.bias,
                           k.res, k.res_mode, k.out, k.Cout, k.H, k.W, total);
    }
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// Requirements checked by the caller (launch_conv): a.src.mode in {0,1}; weights packed with CinP % 16 == 0 rows
// and CoutP % 64 == 0 (see unet.hip load_conv).
Status launch_conv2(hipStream_t s, const ConvArgs& a) {
    Conv2K k;
    k.sa = a.src.a; k.sb = a.src.b; k.ca = a.src.ca; k.cb = a.src.cb; k.Hs = a.src.Hs; k.Ws = a.src.Ws; k.prm = a.src.prm;
    k.w = a.w; k.bias = a.bias; k.out = a.out; k.res = a.res; k.res_mode = a.res_mode;
    k.B = a.B; k.Cin = a.Cin; k.Cout = a.Cout; k.CoutP = a.CoutP; k.H = a.H; k.W = a.W;
    k.partial = a.partial; k.ksplit = 1; k.chunks_per_split = 0; k.dbg = a.dbg;
    k.zeros = conv_zero_page();
    if (!k.zeros) return Status{DPIR_ERR_NOMEM, "conv2: cannot allocate the zero page"};
    int tw = a.W >= 32 ? 32 : (a.W >= 16 ? 16 : (a.W >= 8 ? 8 : 4));
    int th = 256 / tw;
    int hp2 = 1 << ilog2c(a.H);
    if (th > hp2) th = hp2;
    int ti = 256 / (tw * th);
    if (ti > 8) ti = 8;
    k.ti = ti; k.ltw = ilog2c(tw); k.lth = ilog2c(th);
    k.tiles_x = (a.W + tw - 1) / tw;
    k.tiles_y = (a.H + th - 1) / th;
    k.n_ptiles = k.tiles_x * k.tiles_y * ((a.B + ti - 1) / ti);
    k.n_co_blocks = (a.Cout + 63) / 64;
    k.chs = a.ks == 3 ? ti * (th + 2) * (tw + 2) : ti * th * tw;
    if (k.chs > (a.ks == 3 ? 512 : 256)) return invalid("conv2: activation patch too large");
    if (a.CoutP % 64) return invalid("conv2: CoutP must be a multiple of 64");
    if (a.ks == 3) return a.src.mode == 0 ? launch2<3, 8, 0>(s, k, a.partial_capacity) : launch2<3, 8, 1>(s, k, a.partial_capacity);
    return a.src.mode == 0 ? launch2<1, 16, 0>(s, k, a.partial_capacity) : launch2<1, 16, 1>(s, k, a.partial_capacity);
}

}  // namespace dpir
