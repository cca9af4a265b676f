// conv4: 3x3 implicit-GEMM convolution on the f16 matrix pipe with OPERAND SPLITTING: every fp32 operand is split into
// two f16 halves (x = hi + lo, hi = f16(x), lo = f16(x - hi)) and each product is evaluated as
//     lo_w*hi_x + hi_w*lo_x + hi_w*hi_x                (3 x v_mfma_f32_32x32x16_f16, fp32 accumulate, small terms first)
// i.e. a 22-bit-mantissa product; the dropped lo*lo term is 2^-22 relative.  Measured on this chip
// (tools/micro/mfma_f16x3_probe.hip): max error of a K=16 dot product 2.7e-7 vs 4.5e-7 for the exact-fp32 MFMA chain,
// f16 subnormal inputs are NOT flushed, operand mapping A[i][8g+j] / B[8g+j][i'] for lane (i = l%32, g = l/32), element j.
// Weights are pre-scaled by a per-layer power of two (exactly undone in the epilogue) so that their low halves stay normal.
// Why: fp32 MFMA runs on the vector ALU (157 TF/s peak), f16 MFMA has its own pipe at 2.5 PF/s: 3 MFMAs per product =
// 833 TF/s fp32-equivalent.
//
// The kernel is fed ONLY by LDS-DMA: the activation operand has already been normalised,
// activated, resampled, concatenated and split into f16 hi/lo halves by act.hip, in the blocked layout
// [n][C/8][H][W][8] whose 16-byte entries are exactly one lane's MFMA B-operand fragment.  The kernel contains no
// staging arithmetic at all: per K chunk of 16 input channels a workgroup issues
//     36 x 1 KiB weight pieces  + 2 x 21 x 1 KiB activation pieces (hi, lo)     (global_load_lds_dwordx4)
// into the other half of a double buffer and runs 9 taps x 4 tiles x 3 = 108 MFMAs per wave out of the current one.
// Tile: 64 output channels x 512 pixels (16 x 32 patch), 8 waves x (64 x 64): two waves per SIMD, weight traffic per
// MFMA half of the 256-pixel tile's.  LDS: 2 x 36 KiB weights + 2 x 2 x 21 KiB activations = 156 KiB, one workgroup per CU.
#include "common.h"
#include <math.h>
#include <vector>

namespace dpir {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// Ablation switches (tools/conv_ablation.py) exist only in -DDPIR_ABLATE builds; the product kernel has none of them.
#ifdef DPIR_ABLATE
#define ABL(bit) ((p.dbg & (bit)) != 0)
#else
#define ABL(bit) false
#endif

#ifndef DMA_SLOTS_PER_TAP
#define DMA_SLOTS_PER_TAP 2
#endif

struct Conv4K {
    const char* xhi; const char* xlo;      // blocked split activations [n][C8][H][W][16 B]
    int C8;
    const char* w16; const float* bias; float* out; const float* res; int res_mode;
    int B, Cout, H, W;
    int n_chunks_total;
    int ltw, lth, ti;
    int tiles_x, tiles_y, n_ptiles, n_co_blocks;
    int chs;
    int ksplit, chunks_per_split;
    float* partial;
    const float* zeros;
    float out_scale;
    int dbg;
    float2* stat; int stat_slots;          // per-(image, channel, slot) {sum, sum of squares} of the stored values, or null
};

#define GLDS4(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

template <int CTRL>
__device__ __forceinline__ float dpp_row_shr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__global__ __launch_bounds__(512, 2) void conv4_mfma_kernel(Conv4K p) {
    constexpr int TAPS = 9, BCO = 64, WCO = 2, WPX = 2;
    constexpr int PMAX = 648;                       // patch positions per k-half plane
    constexpr int XPIECES = (2 * PMAX + 63) / 64;   // 21 DMA pieces of 64 entries per plane (hi or lo)
    constexpr int XBYTES = XPIECES * 1024;          // bytes per plane buffer
    constexpr int WBYTES = 2 * TAPS * 2 * BCO * 16; // hi + lo weight chunk: 36864
    constexpr int WPIECES = WBYTES / 1024;          // 36
    constexpr int NXT = (XPIECES + 7) / 8;          // activation pieces per wave (per plane)
    constexpr int NWT = (WPIECES + 7) / 8;          // weight pieces per wave
    extern __shared__ __attribute__((aligned(16))) char smem4[];
    char* lds_w = smem4;                            // [2][WBYTES]
    char* lds_x = smem4 + 2 * WBYTES;               // [2][hi|lo][XBYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2); renumber them so that
    // one XCD owns a contiguous range - the co-blocks of a pixel tile and neighbouring tiles then share an L2
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
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
    const int LW = TW + 2, LH = TH + 2;
    const int HW = p.H * p.W;

    // ---- per-lane DMA source offsets of this wave's activation pieces (chunk invariant, in 16-byte entries)
    int x_off[NXT];      // entry offset of (n, kg, gy, gx) inside the blocked tensor for chunk 0; -1 = zero page
#pragma unroll
    for (int u = 0; u < NXT; ++u) {
        int piece = wave + u * 8;
        int f = piece * 64 + lane;
        int kg = f / PMAX;
        int e = f - kg * PMAX;
        int ti = e / (LH * LW);
        int rr = e - ti * (LH * LW);
        int hy = rr / LW, hx = rr - hy * LW;
        int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        int n = n0 + ti;
        bool ok = piece < XPIECES && kg < 2 && e < p.chs && ti < TI && n < p.B && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        x_off[u] = ok ? ((n * p.C8 + kg) * HW + gy * p.W + gx) : -1;
    }

    // ---- per-lane MFMA operand offsets (16-byte entries)
    int boff[WPX];
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        int pp = (wave * WPX + j) * 32 + l31;
        int px = pp & (TW - 1);
        int py = (pp >> p.ltw) & (TH - 1);
        int ti = pp >> (p.ltw + p.lth);
        if (ti >= TI) ti = 0;
        boff[j] = ti * (LH * LW) + py * LW + px + half * PMAX;
    }
    const int aoff = half * BCO + l31;

    floatx16 acc[WCO][WPX];
#pragma unroll
    for (int i = 0; i < WCO; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ch_begin = split * p.chunks_per_split;
    const int ch_end = min(p.n_chunks_total, ch_begin + p.chunks_per_split);

    // One DMA "slot" = one 1 KiB piece of this wave's share of a chunk: slots [0, NWT) are weight pieces, then (hi, lo)
    // pairs of activation pieces.  The prologue issues all slots at once; in steady state they are spread over the first
    // taps of the previous chunk (two per tap), so the vector-memory queue never fills and stalls the MFMA issue.
    constexpr int NSLOT = NWT + 2 * NXT;
    auto issue_slot = [&](int chunk, int buf, int slot) {
        if (ABL(4)) return;
        if (slot < NWT) {
            if (ABL(1024)) return;
            int piece = wave + slot * 8;
            const char* wsrc = p.w16 + ((size_t)chunk * p.n_co_blocks + co_blk) * WBYTES + lane * 16;
            if (piece < WPIECES) GLDS4(wsrc + piece * 1024, lds_w + buf * WBYTES + piece * 1024);
        } else {
            if (ABL(2048)) return;
            const int u = (slot - NWT) >> 1, plane = (slot - NWT) & 1;
            int piece = wave + u * 8;
            if (piece < XPIECES) {
                const bool ok = x_off[u] >= 0;
                const size_t ent = ok ? (size_t)x_off[u] + (size_t)chunk * 2 * HW : 0;      // two C8 groups per chunk
                const char* base = plane ? p.xlo : p.xhi;
                const char* src = ok ? base + ent * 16 : reinterpret_cast<const char*>(p.zeros);
                GLDS4(src, lds_x + buf * 2 * XBYTES + plane * XBYTES + piece * 1024);
            }
        }
    };

#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) issue_slot(ch_begin, 0, sl);
    // Software pipeline across K chunks.  The only workgroup barrier of a chunk sits INSIDE its last tap, between the first
    // MFMA group and the other two: at that point every wave has issued (and, because __syncthreads drains lgkmcnt,
    // completed) all LDS reads of the current buffer, and has waited for its own DMA pieces of the next one, so after the
    // barrier (a) the next chunk's tap-0 operands can be requested while 8 MFMAs are still queued - the MFMA pipe never
    // drains at a chunk boundary - and (b) the current buffer is free for the DMA of the chunk after next.
    const half8* wbase = reinterpret_cast<const half8*>(lds_w);
    const half8* xbase = reinterpret_cast<const half8*>(lds_x);
    half8 ah[2][WCO], al[2][WCO], bh[2][WPX], bl[2][WPX];
    auto load_step = [&](int lbuf, int tap, int rbuf) {
        const half8* wh = wbase + lbuf * (WBYTES / 16);
        const half8* wl = wh + WBYTES / 32;
        const half8* xh = xbase + lbuf * (2 * XBYTES / 16);
        const half8* xl = xh + XBYTES / 16;
        const int toff = (tap / 3) * LW + (tap % 3);
#pragma unroll
        for (int i = 0; i < WCO; ++i) {
            int o = tap * 2 * BCO + aoff + i * 32;
            ah[rbuf][i] = wh[o]; al[rbuf][i] = wl[o];
        }
#pragma unroll
        for (int j = 0; j < WPX; ++j) {
            int o = boff[j] + toff;
            bh[rbuf][j] = xh[o]; bl[rbuf][j] = xl[o];
        }
    };
    __syncthreads();                        // chunk ch_begin has landed (vmcnt(0) precedes the barrier)
    if (!ABL(1)) load_step(0, 0, 0);
    int it = 0;
    for (int chunk = ch_begin; chunk < ch_end; ++chunk, ++it) {
        const int cur = it & 1;
        const bool more = chunk + 1 < ch_end;

        if (ABL(1)) {                       // ablation: DMA only
            if (more) {
#pragma unroll
                for (int sl = 0; sl < NSLOT; ++sl) issue_slot(chunk + 1, cur ^ 1, sl);
                __syncthreads();
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if (more) {
                    constexpr int SPT = DMA_SLOTS_PER_TAP;
#pragma unroll
                    for (int q = 0; q < SPT; ++q)
                        if (SPT * tap + q < NSLOT) issue_slot(chunk + 1, cur ^ 1, SPT * tap + q);
                }
                __builtin_amdgcn_sched_barrier(0);
                // the three partial products of one accumulator are issued four MFMAs apart (no back-to-back dependency);
                // small terms first, as in conv3
#pragma unroll
                for (int i = 0; i < WCO; ++i)
#pragma unroll
                    for (int j = 0; j < WPX; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tap & 1][i], bh[tap & 1][j], acc[i][j], 0, 0, 0);
                // The next tap's operands are requested only now, with 8 MFMAs still to issue in front of their first use:
                // hipcc waits lgkmcnt(0) before a tap's first MFMA, so nothing younger may be in flight at that point.
                __builtin_amdgcn_sched_barrier(0);
                if (tap + 1 < TAPS) {
                    load_step(cur, tap + 1, (tap + 1) & 1);
                } else if (more) {
                    __syncthreads();
                    load_step(cur ^ 1, 0, 1);            // tap 8 computes from register set 0; copied to set 0 below
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < WCO; ++i)
#pragma unroll
                    for (int j = 0; j < WPX; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap & 1][i], bl[tap & 1][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < WCO; ++i)
#pragma unroll
                    for (int j = 0; j < WPX; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap & 1][i], bh[tap & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < WCO; ++i) { ah[0][i] = ah[1][i]; al[0][i] = al[1][i]; }
#pragma unroll
                for (int j = 0; j < WPX; ++j) { bh[0][j] = bh[1][j]; bl[0][j] = bl[1][j]; }
            }
        }
    }

    // ---- epilogue.  Vector path (W % 4 == 0): each wave transposes its 64co x 64px accumulators through its own 17 KiB
    // LDS slab so that every lane owns 4 consecutive pixels of one channel: float4 residual loads and float4 NCHW stores
    // (4 channel rows x 256 B per instruction) instead of 64 scalar stores per lane.
    if ((p.W & 3) == 0 && !ABL(16)) {
        constexpr int TS = 68;                                   // slab row stride in floats (16-byte aligned, bank-skewed)
        // Same-resolution residual: all 16 float4 loads of this lane are requested up front, so their latency is paid
        // once (and overlaps the transpose) instead of once per unrolled batch of the store loop.
        const bool pre_res = p.res != nullptr && p.res_mode == 0 && p.ksplit == 1;
        float4 rpre[16];
        {
            const int q4 = lane & 15, rsub = lane >> 4;
            const int pp = wave * 64 + q4 * 4;
            const int ti = pp >> (p.ltw + p.lth);
            const int n = n0 + ti, y = ty0 + ((pp >> p.ltw) & (TH - 1)), x = tx0 + (pp & (TW - 1));
            const bool ok = pre_res && ti < TI && n < p.B && y < p.H && x < p.W;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int co = co0 + it * 4 + rsub;
                rpre[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok && co < p.Cout) rpre[it] = *reinterpret_cast<const float4*>(p.res + ((size_t)n * p.Cout + co) * HW + (size_t)y * p.W + x);
            }
        }
        __syncthreads();                                         // operand buffers are dead from here on
        float* tr = reinterpret_cast<float*>(smem4) + wave * (64 * TS);
#pragma unroll
        for (int i = 0; i < WCO; ++i)
#pragma unroll
            for (int j = 0; j < WPX; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tr[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * TS + j * 32 + l31] = acc[i][j][r] * p.out_scale;
        __syncthreads();
        const int q4 = lane & 15, rsub = lane >> 4;
        const int pp = wave * 64 + q4 * 4;
        const int px = pp & (TW - 1);
        const int py = (pp >> p.ltw) & (TH - 1);
        const int ti = pp >> (p.ltw + p.lth);
        const int n = n0 + ti, y = ty0 + py, x = tx0 + px;
        const bool img_ok = ti < TI && n < p.B;                       // wave uniform: a wave's 64 pixels share one image
        const bool pok = img_ok && y < p.H && x < p.W;
        if (!img_ok) return;
        const size_t pix = (size_t)y * p.W + x;
        float* dst = p.ksplit > 1 ? p.partial + (size_t)split * ((size_t)p.B * p.Cout * HW) : p.out;
        // GroupNorm statistics of the tensor being written, fused: this wave's 64-pixel partial sums go to its own slot
        const bool do_stat = p.stat != nullptr && p.ksplit == 1;
        const int wpi = (TW * TH) >> 6;                                // waves per image inside one tile
        const int slot = trem * wpi + (wave & (wpi - 1));
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int co_l = it * 4 + rsub;
            const int co = co0 + co_l;
            const bool cok = co < p.Cout;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cok && pok) {
                v = *reinterpret_cast<const float4*>(tr + co_l * TS + q4 * 4);
                const size_t plane = (size_t)n * p.Cout + co;
                if (p.ksplit == 1) {
                    const float bv = p.bias[co];
                    v.x += bv; v.y += bv; v.z += bv; v.w += bv;
                    if (p.res) {
                        if (p.res_mode == 0) {
                            const float4 rv = rpre[it];
                            v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w;
                        } else if (p.res_mode == 1) {
                            int Hr = p.H >> 1, Wr = p.W >> 1;
                            float2 rv = *reinterpret_cast<const float2*>(p.res + plane * (Hr * Wr) + (y >> 1) * Wr + (x >> 1));
                            v.x = rv.x + v.x; v.y = rv.x + v.y; v.z = rv.y + v.z; v.w = rv.y + v.w;
                        } else {
                            int Wr = p.W * 2;
                            const float* rp = p.res + plane * (4 * (size_t)HW) + (size_t)(2 * y) * Wr + 2 * x;
                            float4 a0 = *reinterpret_cast<const float4*>(rp), a1 = *reinterpret_cast<const float4*>(rp + 4);
                            float4 b0 = *reinterpret_cast<const float4*>(rp + Wr), b1 = *reinterpret_cast<const float4*>(rp + Wr + 4);
                            v.x = ((a0.x + a0.y) + (b0.x + b0.y)) * 0.25f + v.x;
                            v.y = ((a0.z + a0.w) + (b0.z + b0.w)) * 0.25f + v.y;
                            v.z = ((a1.x + a1.y) + (b1.x + b1.y)) * 0.25f + v.z;
                            v.w = ((a1.z + a1.w) + (b1.z + b1.w)) * 0.25f + v.w;
                        }
                    }
                }
                *reinterpret_cast<float4*>(dst + plane * HW + pix) = v;
            }
            if (do_stat) {
                float s1 = (v.x + v.y) + (v.z + v.w);
                float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                // inclusive scan over the 16-lane DPP row (row_shr 1, 2, 4, 8, zero fill): lane 15 of the row ends with the total
                s1 += dpp_row_shr<0x111>(s1); s2 += dpp_row_shr<0x111>(s2);
                s1 += dpp_row_shr<0x112>(s1); s2 += dpp_row_shr<0x112>(s2);
                s1 += dpp_row_shr<0x114>(s1); s2 += dpp_row_shr<0x114>(s2);
                s1 += dpp_row_shr<0x118>(s1); s2 += dpp_row_shr<0x118>(s2);
                if (q4 == 15 && cok) p.stat[((size_t)n * p.Cout + co) * p.stat_slots + slot] = make_float2(s1, s2);
            }
        }
        return;
    }
    // scalar path (odd widths, and the "no epilogue" ablation)
    const bool full_co = co0 + BCO <= p.Cout;
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

static int ilog2e(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// Tile geometry: a 512-pixel tile is TI images x TH rows x TW columns (all powers of two; lanes beyond the image or beyond
// TI are masked).  TI is cut down when the TI halo patches would not fit the 648-entry plane (8x8 layers: 6 images).
struct Conv4Geo { int tw, th, ti, slots; bool ok; };
static Conv4Geo conv4_geometry(int H, int W) {
    Conv4Geo g{0, 0, 0, 0, false};
    if (W < 8 || H < 8) return g;
    g.tw = W >= 32 ? 32 : (W >= 16 ? 16 : 8);
    g.th = 512 / g.tw;
    int hp2 = 1 << ilog2e(H);
    if (g.th > hp2) g.th = hp2;
    if (g.tw * g.th < 64) return g;                   // a wave's 64 pixels must belong to one image
    g.ti = 512 / (g.tw * g.th);
    const int patch = (g.th + 2) * (g.tw + 2);
    if (g.ti * patch > 648) g.ti = 648 / patch;
    if (g.ti < 1 || g.ti > 8) return g;
    g.slots = ((W + g.tw - 1) / g.tw) * ((H + g.th - 1) / g.th) * ((g.tw * g.th) >> 6);
    g.ok = true;
    return g;
}

// geometry shared with the executor: does conv4 tile this output shape?
bool conv4_supported(int H, int W) { return conv4_geometry(H, W).ok; }

// number of statistics slots per (image, channel) plane that launch_conv4 fills when Conv4Args::stat is set
int conv4_stat_slots(int H, int W) {
    Conv4Geo g = conv4_geometry(H, W);
    return (g.ok && (W & 3) == 0) ? g.slots : 0;
}

Status launch_conv4(hipStream_t s, const Conv4Args& a, bool* stat_written) {
    if (stat_written) *stat_written = false;
    if (!conv4_supported(a.H, a.W)) return Status{DPIR_ERR_UNSUPPORTED, "conv4: shape not tiled"};
    Conv4K k;
    k.xhi = reinterpret_cast<const char*>(a.xhi); k.xlo = reinterpret_cast<const char*>(a.xlo); k.C8 = (a.Cin + 7) / 8;
    k.w16 = reinterpret_cast<const char*>(a.w16); k.bias = a.bias; k.out = a.out; k.res = a.res; k.res_mode = a.res_mode;
    k.B = a.B; k.Cout = a.Cout; k.H = a.H; k.W = a.W;
    k.n_chunks_total = (a.Cin + 15) / 16;
    k.C8 = 2 * k.n_chunks_total;          // act.hip pads the blocked tensor to whole 16-channel chunks
    k.partial = a.partial; k.ksplit = 1; k.chunks_per_split = 0; k.dbg = a.dbg;
    k.out_scale = 1.0f / a.w16_scale;
    k.zeros = conv_zero_page();
    if (!k.zeros) return Status{DPIR_ERR_NOMEM, "conv4: cannot allocate the zero page"};
    const Conv4Geo geo = conv4_geometry(a.H, a.W);
    const int tw = geo.tw, th = geo.th, ti = geo.ti;
    k.ti = ti; k.ltw = ilog2e(tw); k.lth = ilog2e(th);
    k.tiles_x = (a.W + tw - 1) / tw;
    k.tiles_y = (a.H + th - 1) / th;
    k.n_ptiles = k.tiles_x * k.tiles_y * ((a.B + ti - 1) / ti);
    k.n_co_blocks = (a.Cout + 63) / 64;
    k.chs = ti * (th + 2) * (tw + 2);
    constexpr size_t LDS = 2 * 36864 + 2 * 2 * 21 * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        DPIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv4_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int chunks = k.n_chunks_total;
    const int blocks = k.n_ptiles * k.n_co_blocks;
    int S = 1;
    if (k.partial && blocks < 192) {
        S = (256 + blocks - 1) / blocks;
        if (S > chunks / 2) S = chunks / 2;
        if (S > 16) S = 16;
        if (S < 1) S = 1;
        if ((size_t)S * k.B * k.Cout * k.H * k.W > a.partial_capacity) S = 1;
    }
    k.ksplit = S;
    k.stat = nullptr; k.stat_slots = 0;
    if (a.stat && S == 1 && (a.dbg & 16) == 0) {
        k.stat_slots = conv4_stat_slots(a.H, a.W);
        if (k.stat_slots > 0) { k.stat = a.stat; if (stat_written) *stat_written = true; }
    }
    k.chunks_per_split = (chunks + S - 1) / S;
    if (S == 1) k.partial = nullptr;
    hipLaunchKernelGGL(conv4_mfma_kernel, dim3((unsigned)(blocks * S)), dim3(512), LDS, s, k);
    if (S > 1) {
        size_t total = (size_t)k.B * k.Cout * k.H * k.W;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k.partial, S, k.bias,
                           k.res, k.res_mode, k.out, k.Cout, k.H, k.W, total);
    }
    DPIR_HIP(hipGetLastError());
    return Status{};
}


// Host: OIHW fp32 -> [chunk][co-block][hi|lo][tap][k-half (2KB)][64 co][8] f16, scaled by a power of two so that
// max|w|*scale is in [512, 1024).  Returns the scale.
float pack_weights_f16x3(const float* w, int cout, int cin, int ks, std::vector<uint16_t>& out) {
    const int taps = ks * ks, KB = 1, KC = 16;   // ks == 3 (the 1x1 layout is pack_weights_f16x3_1x1, conv5.hip)
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
