// conv3: implicit-GEMM convolution on the gfx950 MATRIX pipe with fp32-equivalent accuracy: every fp32 operand is
// split into two f16 halves (x = hi + lo, hi = f16(x), lo = f16(x - hi)) and each product is evaluated as
//     hi_w*hi_x + hi_w*lo_x + lo_w*hi_x                (3 x v_mfma_f32_32x32x16_f16, fp32 accumulate)
// i.e. a 22-bit-mantissa product; the dropped lo*lo term is 2^-22 relative.  Measured on this chip
// (tools/micro/mfma_f16x3_probe): max error of a K=16 dot product 2.7e-7 vs 4.5e-7 for the exact-fp32 MFMA chain
// (the f16 MFMA sums its 16 products in a wider internal accumulator), f16 subnormal inputs are NOT flushed, and
// the operand mapping is A[i][8g+j] / B[8g+j][i'] for lane (i = l%32, g = l/32), element j.  Weights are pre-scaled
// by a per-layer power of two (exactly undone in the epilogue) so that their low halves stay normal.
// Why: fp32 MFMA runs on the vector ALU (157 TF/s peak = the VALU peak; the round-1 ablation shows zero overlap between
// it and the staging VALU work), while f16 MFMA has its own pipe at 2.5 PF/s: 3 MFMAs per product = 833 TF/s
// fp32-equivalent, 5.3x the fp32-MFMA rate, with the GroupNorm/SiLU prologue running in its shadow.
//
// Same fusions and tile (64 output channels x 256 pixels, 4 waves x (64 x 64)) as conv2.hip.  Per K chunk of 16
// input channels (KB k16-blocks per chunk: 1 for 3x3, 2 for 1x1):
//   weights   : pre-split, pre-swizzled at load time into [chunk][co-block][hi|lo][tap][k-half][co][8 halves] so that one
//               LDS-DMA stream (global_load_lds_dwordx4) lands them in MFMA A-operand order (ds_read_b128 per lane);
//   activations: raw fp32 loads (8 channels of one patch position per thread-task) issued before the MFMA phase,
//               GroupNorm affine + FiLM + SiLU + hi/lo split + pack after it, two ds_write_b128 into the other LDS
//               buffer laid out [k-half][patch position][8 halves] = MFMA B-operand order.
#include "common.h"
#include <string.h>
#include <math.h>

namespace dpir {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct Conv3K {
    const float* sa; const float* sb; int ca, cb; int Hs, Ws; const float4* prm;
    const _Float16* w16; const float* bias; float* out; const float* res; int res_mode;
    int B, Cin, Cout, H, W;
    int n_chunks_total;    // CinP / KC
    int ltw, lth, ti;
    int tiles_x, tiles_y, n_ptiles, n_co_blocks;
    int chs;               // patch positions
    int ksplit, chunks_per_split;
    float* partial;
    const float* zeros;
    float out_scale;       // 1 / weight scale (power of two)
    int dbg;
};

__device__ __forceinline__ float silu3_f(float v) {
    float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

#define GLDS3(src, dst, bytes) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), bytes, 0, 0)

template <int KS, int KB, int MODE>   // KB = k16-blocks per chunk; MODE 0 plain, 1 nearest-up source
__global__ __launch_bounds__(256, 2) void conv3_mfma_kernel(Conv3K p) {
    constexpr int TAPS = KS * KS;
    constexpr int KC = 16 * KB;
    constexpr int BCO = 64;
    constexpr int WCO = 2, WPX = 2;
    constexpr int PMAX = (KS == 3) ? 344 : 256;            // patch stride (positions)
    constexpr int NT = (PMAX * 2 * KB + 255) / 256;         // thread-tasks (position, k-half group) per chunk
    constexpr int WPLANE = TAPS * 2 * KB * BCO * 8;         // halves per weight plane (hi or lo) per chunk
    constexpr int WBYTES = 2 * WPLANE * 2;                  // bytes per weight chunk (hi + lo)
    constexpr int NDMA = WBYTES / 1024;
    constexpr int XPLANE = 2 * KB * PMAX * 8;               // halves per activation plane
    static_assert(WBYTES % 1024 == 0, "weight chunk must be whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    _Float16* lds_w = reinterpret_cast<_Float16*>(smem3);                          // [2][hi|lo][TAPS][2KB][64][8]
    _Float16* lds_x = reinterpret_cast<_Float16*>(smem3 + 2 * WBYTES);             // [2][hi|lo][2KB][PMAX][8]
    float4* lds_prm = reinterpret_cast<float4*>(smem3 + 2 * WBYTES + 2 * 2 * XPLANE * 2);   // [2][KC][8]

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

    // ---- per-thread tasks (chunk invariant): task = (patch position e, k-half group kg): 8 channels c0 + 8 kg + j
    int tk_kg[NT], tk_e[NT], tk_ti[NT], tk_cs[NT];
    bool tk_ok[NT], tk_in[NT];
    const float* tk_pa[NT];
    const float* tk_pb[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        int idx = tid + q * 256;
        int kg = idx / p.chs;
        int e = idx - kg * p.chs;
        bool in = kg < 2 * KB;
        int ti = e / (LH * LW);
        int rr = e - ti * (LH * LW);
        int hy = rr / LW, hx = rr - hy * LW;
        int gy = ty0 + hy - (KS == 3 ? 1 : 0);
        int gx = tx0 + hx - (KS == 3 ? 1 : 0);
        int n = n0 + ti;
        bool ok = in && ti < TI && n < p.B && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        int so = (MODE == 0) ? gy * p.Ws + gx : (gy >> 1) * p.Ws + (gx >> 1);
        tk_kg[q] = in ? kg : 0;
        tk_e[q] = e;
        tk_ti[q] = ok ? ti : 0;
        tk_in[q] = in;
        tk_ok[q] = ok;
        tk_cs[q] = ok ? HsWs : 0;
        tk_pa[q] = ok ? p.sa + (size_t)n * p.ca * HsWs + so : p.zeros;
        tk_pb[q] = (ok && p.sb) ? p.sb + (size_t)n * p.cb * HsWs + so : p.zeros;
    }

    // ---- per-lane MFMA operand offsets (in 16-byte entries)
    int boff[WPX];
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        int pp = (wave * WPX + j) * 32 + l31;
        int px = pp & (TW - 1);
        int py = (pp >> p.ltw) & (TH - 1);
        int ti = pp >> (p.ltw + p.lth);
        if (ti >= TI) ti = 0;
        boff[j] = ti * (LH * LW) + py * LW + px + half * PMAX;      // + kb*2*PMAX + tap shift
    }
    const int aoff = half * BCO + l31;                              // + (tap*2KB + kb*2)*64 + i*32

    floatx16 acc[WCO][WPX];
#pragma unroll
    for (int i = 0; i < WCO; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ch_begin = split * p.chunks_per_split;
    const int ch_end = min(p.n_chunks_total, ch_begin + p.chunks_per_split);

    auto dma_weights = [&](int chunk, int buf) {
        if (p.dbg & 4) return;
        const char* src = reinterpret_cast<const char*>(p.w16) + ((size_t)chunk * p.n_co_blocks + co_blk) * WBYTES + lane * 16;
        char* dst = smem3 + buf * WBYTES;
#pragma unroll
        for (int u = 0; u < (NDMA + 3) / 4; ++u) {
            int piece = wave + u * 4;
            if (piece < NDMA) GLDS3(src + piece * 1024, dst + piece * 1024, 16);
        }
    };
    float4 prm_reg = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch_prm = [&](int c0) {
        if (p.prm && tid < KC * 8) {
            int k = tid >> 3, ti = tid & 7;
            int c = min(c0 + k, p.Cin - 1), n = min(n0 + ti, p.B - 1);
            prm_reg = p.prm[(size_t)n * C + c];
        }
    };
    auto commit_prm = [&](int buf) {
        if (p.prm && tid < KC * 8) lds_prm[buf * KC * 8 + tid] = prm_reg;
    };
    float vals[NT][8];
    auto load_acts = [&](int c0) {
        if (p.dbg & 4) {
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) vals[q][j] = 0.3f;
            return;
        }
        const bool in_a = c0 < p.ca;                        // a chunk never straddles the concat boundary (checked on the host)
        const int cb0 = in_a ? c0 : c0 - p.ca;
        const int cmax = (in_a ? p.ca : p.cb) - 1;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const float* base = (in_a ? tk_pa[q] : tk_pb[q]);
            const int cfirst = cb0 + 8 * tk_kg[q];
#pragma unroll
            for (int j = 0; j < 8; ++j) vals[q][j] = base[min(cfirst + j, cmax) * tk_cs[q]];   // 32-bit offsets (< 2^31 floats per tensor slice)
        }
    };
    auto store_acts = [&](int buf, int c0) {
        if (p.dbg & 8) return;
        _Float16* xh = lds_x + (buf * 2 + 0) * XPLANE;
        _Float16* xl = lds_x + (buf * 2 + 1) * XPLANE;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            if (!tk_in[q]) continue;
            half8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = vals[q][j];
                int kl = 8 * tk_kg[q] + j;
                if (p.prm && !(p.dbg & 2)) {
                    float4 m = lds_prm[(buf * KC + kl) * 8 + tk_ti[q]];
                    v = (v - m.x) * m.y + m.z;
                    if (m.w != 0.f) v = silu3_f(v);
                }
                v = (tk_ok[q] && (c0 + kl) < p.Cin) ? v : 0.f;
                v = fminf(fmaxf(v, -65000.f), 65000.f);
                _Float16 h = (_Float16)v;
                hi[j] = h;
                lo[j] = (_Float16)(v - (float)h);
            }
            int ent = tk_kg[q] * PMAX + tk_e[q];
            *reinterpret_cast<half8*>(xh + ent * 8) = hi;
            *reinterpret_cast<half8*>(xl + ent * 8) = lo;
        }
    };

    // ---- prologue: first chunk into buffer 0
    dma_weights(ch_begin, 0);
    fetch_prm(ch_begin * KC); commit_prm(0);
    load_acts(ch_begin * KC);
    __syncthreads();
    store_acts(0, ch_begin * KC);
    fetch_prm((ch_begin + 1) * KC); commit_prm(1);

    int it = 0;
    for (int chunk = ch_begin; chunk < ch_end; ++chunk, ++it) {
        const int cur = it & 1;
        __syncthreads();
        const bool more = chunk + 1 < ch_end;
        if (more) {
            dma_weights(chunk + 1, cur ^ 1);
            load_acts((chunk + 1) * KC);
        }
        fetch_prm((chunk + 2) * KC);

        if (!(p.dbg & 1)) {
            const half8* wh = reinterpret_cast<const half8*>(lds_w + (size_t)cur * 2 * WPLANE);
            const half8* wl = wh + WPLANE / 8;
            const half8* xh = reinterpret_cast<const half8*>(lds_x + (size_t)(cur * 2) * XPLANE);
            const half8* xl = xh + XPLANE / 8;
            // operands of step s+1 (one (tap, k16-block) pair = 12 MFMAs) are read from LDS while step s computes
            constexpr int NSTEP = TAPS * KB;
            half8 ah[2][WCO], al[2][WCO], bh[2][WPX], bl[2][WPX];
            auto load_step = [&](int st, int buf) {
                const int tap = st / KB, kb = st % KB;
                const int toff = (KS == 3) ? (tap / 3) * LW + (tap % 3) : 0;
#pragma unroll
                for (int i = 0; i < WCO; ++i) {
                    int o = (tap * 2 * KB + kb * 2) * BCO + aoff + i * 32;
                    ah[buf][i] = wh[o]; al[buf][i] = wl[o];
                }
#pragma unroll
                for (int j = 0; j < WPX; ++j) {
                    int o = kb * 2 * PMAX + boff[j] + toff;
                    bh[buf][j] = xh[o]; bl[buf][j] = xl[o];
                }
            };
            load_step(0, 0);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + 1 < NSTEP) load_step(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < WCO; ++i)
#pragma unroll
                    for (int j = 0; j < WPX; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st & 1][i], bh[st & 1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st & 1][i], bl[st & 1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st & 1][i], bh[st & 1][j], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) store_acts(cur ^ 1, (chunk + 1) * KC);
        commit_prm(cur);
    }

    // ---- epilogue: un-scale, bias + residual, 128-byte coalesced NCHW stores
    const int HW = p.H * p.W;
    const bool full_co = co0 + BCO <= p.Cout;
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        int pp = (wave * WPX + j) * 32 + l31;
        int px = pp & (TW - 1);
        int py = (pp >> p.ltw) & (TH - 1);
        int ti = pp >> (p.ltw + p.lth);
        int n = n0 + ti, y = ty0 + py, x = tx0 + px;
        bool pok = ti < TI && n < p.B && y < p.H && x < p.W;
        if (p.dbg & 16) {
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
                    if (full_co || co < p.Cout) pb[(size_t)co * HW] = acc[i][j][r] * p.out_scale;
                }
            continue;
        }
        float* ob = p.out + (size_t)n * p.Cout * HW + pix;
#pragma unroll
        for (int i = 0; i < WCO; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (full_co || co < p.Cout) {
                    float v = acc[i][j][r] * p.out_scale + p.bias[co];
                    if (p.res) {
                        float rv;
                        if (p.res_mode == 0) {
                            rv = p.res[((size_t)n * p.Cout + co) * HW + pix];
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
                    ob[(size_t)co * HW] = v;
                }
            }
    }
}

__global__ void conv_splitk_reduce_kernel(const float* partial, int ksplit, const float* bias, const float* res, int res_mode,
                                          float* out, int Cout, int H, int W, size_t total);
const float* conv_zero_page();

static int ilog2d(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

template <int KS, int KB, int MODE>
static Status launch3(hipStream_t s, Conv3K k, size_t partial_cap) {
    constexpr int TAPS = KS * KS;
    constexpr int KC = 16 * KB;
    constexpr int PMAX = (KS == 3) ? 344 : 256;
    constexpr int WBYTES = 2 * TAPS * 2 * KB * 64 * 8 * 2;
    constexpr int XPLANE = 2 * KB * PMAX * 8;
    size_t lds = (size_t)2 * WBYTES + (size_t)2 * 2 * XPLANE * 2 + (size_t)2 * KC * 8 * 16;
    auto fn = conv3_mfma_kernel<KS, KB, MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        DPIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int chunks = k.n_chunks_total;
    const int blocks = k.n_ptiles * k.n_co_blocks;
    int S = 1;
    if (k.partial && blocks < 384) {
        S = (512 + blocks - 1) / blocks;
        if (S > chunks / 2) S = chunks / 2;
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

// a.w16 / a.w16_scale must have been prepared by pack_weights_f16x3 (same KB as used here: 1 for 3x3, 2 for 1x1).
Status launch_conv3(hipStream_t s, const ConvArgs& a) {
    Conv3K k;
    const int KC = a.ks == 3 ? 16 : 32;
    if (a.src.cb > 0 && (a.src.ca % KC)) return Status{DPIR_ERR_UNSUPPORTED, "conv3: concat boundary must be a multiple of the K chunk"};
    k.sa = a.src.a; k.sb = a.src.b; k.ca = a.src.ca; k.cb = a.src.cb; k.Hs = a.src.Hs; k.Ws = a.src.Ws; k.prm = a.src.prm;
    k.w16 = reinterpret_cast<const _Float16*>(a.w16); k.bias = a.bias; k.out = a.out; k.res = a.res; k.res_mode = a.res_mode;
    k.B = a.B; k.Cin = a.Cin; k.Cout = a.Cout; k.H = a.H; k.W = a.W;
    k.n_chunks_total = (a.Cin + KC - 1) / KC;
    k.partial = a.partial; k.ksplit = 1; k.chunks_per_split = 0; k.dbg = a.dbg;
    k.out_scale = 1.0f / a.w16_scale;
    k.zeros = conv_zero_page();
    if (!k.zeros) return Status{DPIR_ERR_NOMEM, "conv3: cannot allocate the zero page"};
    int tw = a.W >= 32 ? 32 : (a.W >= 16 ? 16 : (a.W >= 8 ? 8 : 4));
    int th = 256 / tw;
    int hp2 = 1 << ilog2d(a.H);
    if (th > hp2) th = hp2;
    int ti = 256 / (tw * th);
    if (ti > 8) ti = 8;
    k.ti = ti; k.ltw = ilog2d(tw); k.lth = ilog2d(th);
    k.tiles_x = (a.W + tw - 1) / tw;
    k.tiles_y = (a.H + th - 1) / th;
    k.n_ptiles = k.tiles_x * k.tiles_y * ((a.B + ti - 1) / ti);
    k.n_co_blocks = (a.Cout + 63) / 64;
    k.chs = a.ks == 3 ? ti * (th + 2) * (tw + 2) : ti * th * tw;
    if (k.chs > (a.ks == 3 ? 344 : 256)) return Status{DPIR_ERR_UNSUPPORTED, "conv3: activation patch too large for this tile"};
    if (a.ks == 3) return a.src.mode == 0 ? launch3<3, 1, 0>(s, k, a.partial_capacity) : launch3<3, 1, 1>(s, k, a.partial_capacity);
    return a.src.mode == 0 ? launch3<1, 2, 0>(s, k, a.partial_capacity) : launch3<1, 2, 1>(s, k, a.partial_capacity);
}

// Host: OIHW fp32 -> [chunk][co-block][hi|lo][tap][k-half (2KB)][64 co][8] f16, scaled by a power of two so that
// max|w|*scale is in [512, 1024).  Returns the scale.
float pack_weights_f16x3(const float* w, int cout, int cin, int ks, std::vector<uint16_t>& out) {
    const int taps = ks * ks, KB = ks == 3 ? 1 : 2, KC = 16 * KB;
    const int chunks = (cin + KC - 1) / KC, cblocks = (cout + 63) / 64;
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin * taps; ++i) mx = fmaxf(mx, fabsf(w[i]));
    float scale = 1.0f;
    if (mx > 0.f) scale = exp2f(floorf(log2f(1024.0f / mx)) - 0.0f);
    while (mx * scale >= 1024.0f) scale *= 0.5f;
    const size_t plane = (size_t)taps * 2 * KB * 64 * 8;
    out.assign((size_t)chunks * cblocks * 2 * plane, 0);
    for (int ch = 0; ch < chunks; ++ch)
        for (int cbk = 0; cbk < cblocks; ++cbk) {
            uint16_t* hi = out.data() + ((size_t)ch * cblocks + cbk) * 2 * plane;
            uint16_t* lo = hi + plane;
            for (int tap = 0; tap < taps; ++tap)
                for (int kh = 0; kh < 2 * KB; ++kh)
                    for (int col = 0; col < 64; ++col)
                        for (int j = 0; j < 8; ++j) {
                            int co = cbk * 64 + col, ci = ch * KC + kh * 8 + j;
                            float v = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * taps + tap] * scale : 0.f;
                            _Float16 h = (_Float16)v;
                            _Float16 l = (_Float16)(v - (float)h);
                            size_t o = (((size_t)tap * 2 * KB + kh) * 64 + col) * 8 + j;
                            __builtin_memcpy(&hi[o], &h, 2);
                            __builtin_memcpy(&lo[o], &l, 2);
                        }
        }
    return scale;
}

}  // namespace dpir
