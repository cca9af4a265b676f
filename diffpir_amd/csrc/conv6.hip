// conv6: 3x3 implicit-GEMM convolution on the f16 matrix pipe with OPERAND SPLITTING: every fp32 operand is split into two f16
// halves (x = hi + lo, hi = f16(x), lo = f16(x - hi)) and each product is evaluated as
//     lo_w*hi_x + hi_w*lo_x + hi_w*hi_x                (3 x v_mfma_f32_32x32x16_f16, fp32 accumulate, small terms first)
// i.e. a 22-bit-mantissa product; the dropped lo*lo term is 2^-22 relative.  Measured on this chip
// (tools/micro/mfma_f16x3_probe.hip): max error of a K=16 dot product 2.7e-7 vs 4.5e-7 for the exact-fp32 MFMA chain, f16
// subnormal inputs are NOT flushed, operand mapping A[i][8g+j] / B[8g+j][i'] for lane (i = l%32, g = l/32), element j.  Weights
// are pre-scaled by a per-layer power of two (exactly undone in the epilogue) so that their low halves stay normal.
// Why: fp32 MFMA runs at the vector rate (157 TF/s), f16 MFMA has its own pipe at 2.5 PF/s: 3 MFMAs per product = 833 TF/s
// fp32-equivalent.  The activation operand arrives already normalised, activated, resampled, concatenated and split
// (act.hip), so the kernel is pure LDS-DMA + MFMA.
//
// Structure: TWO WORKGROUPS PER CU.  The previous generation (round 1's conv4: 64 co x 512 px, 8 waves, weights of a K chunk
// staged once for all waves) kept one 156 KiB workgroup per CU: its prologue (first operand DMA, ~2 us), its epilogue
// (64 co x 512 px fp32 = 128 KiB of stores + as much residual, bound by the per-CU store issue rate: 6-9 us) and its
// barrier bubbles are all serialised with the MFMA stream -- at Cin = 128 the matrix pipe is busy 27.6 us of a 55 us
// workgroup lifetime.  Here a workgroup needs 76 KiB of LDS and <= 256 VGPRs, so the hardware keeps TWO resident per CU
// (two waves per SIMD, one from each) and one workgroup's prologue / epilogue / barrier waits run under the other's MFMAs.
// Measured in the network (FFHQ, B = 16): 3x3 class 17.1 -> 16.3 ms per forward, isolated layers -8 ... -17 %.
//
// What made the LDS fit.  The 36 KiB weight stage of conv4 (64 co x 16 ci x 9 taps x hi/lo, shared by 8 waves, double
// buffered = 72 KiB) is gone: the workgroup tile is 128 output channels x 256 pixels and each of the 4 waves owns
// 32 output channels x ALL 256 pixels, so no two waves of a workgroup need the same weights.  Every wave streams its own
// A operands (one 2 KiB piece per tap: hi + lo fragments of 32 co x 16 ci, stored by the host in lane order) through a
// PRIVATE ring of R slots by LDS-DMA, ordered by nothing but the wave's own counted s_waitcnt vmcnt -- no barrier, no
// sharing.  Only the activation patch (hi + lo, 11-13 KiB per plane, the blocked [n][C/8][H][W][8] tensor of act.hip) is
// shared and double buffered: one workgroup barrier per 16-channel K chunk, as before.  Per unit of work the activation
// DMA traffic is a quarter of conv4's (one patch feeds 128 output channels instead of 64, 256-pixel patch), the weight
// DMA traffic is equal.
//
// Per K chunk and wave: 9 taps x 8 pixel tiles x 3 = 216 MFMAs (accumulators 8 x 16 = 128 registers), 9 x 18
// ds_read_b128 (2 A + 16 B fragments per tap, B in four groups of two tiles so the next operands are requested while
// the previous group's MFMAs are queued), 18 weight pieces + 6-8 activation pieces of LDS-DMA.
//
// Tile geometries (256 pixels): 8 rows x 32 columns (W >= 32), 16 x 16 (W >= 16), 4 images x 8 x 8 (W >= 8).
#include "common.h"
#include "lds_dma.h"
#include "conv6_params.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

namespace dpir {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));


template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_row_shr6(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <int GEO> struct Geo6;
template <> struct Geo6<0> { static constexpr int LTW = 5, LTH = 3, TI = 1; };   // 8 rows x 32 columns
template <> struct Geo6<1> { static constexpr int LTW = 4, LTH = 4, TI = 1; };   // 16 x 16
template <> struct Geo6<2> { static constexpr int LTW = 3, LTH = 3, TI = 4; };   // 4 images x 8 x 8

// A/B switch of the chunk-boundary barrier (lds_dma.h barrier_lds_only against __syncthreads with its vmcnt(0))
#ifndef DPIR_C6_LDS_BARRIER
#define DPIR_C6_LDS_BARRIER 1
#endif
[[maybe_unused]] constexpr bool LDS_BARRIER = DPIR_C6_LDS_BARRIER != 0;

// DMA instructions issued by the group of tap i of a chunk (i < 0: tap i + 9 of the previous chunk, always a MORE body):
// [one activation piece of the next chunk if i < nact] + [the two weight pieces of tap i + d]
// wp: weight pieces per tap (2 = hi + lo; 1 in the single-product f16x1 mode)
constexpr int c6_gsize(bool more, int nact, int d, int i, int wp) {
    const int j = i < 0 ? i + 9 : i;
    const bool m = i < 0 ? true : more;
    return ((m && j < nact) ? 1 : 0) + ((m || j + d < 9) ? wp : 0);
}
// number of DMA instructions issued AFTER the weights of tap + 1 at the point where they are read (after the group of `tap`)
constexpr int c6_wait_n(bool more, int nact, int d, int tap, int wp) {
    int s = 0;
    for (int i = tap + 2 - d; i <= tap; ++i) s += c6_gsize(more, nact, d, i, wp);
    return s;
}

// X1: single-product mode (f16x1): only the hi halves of both operands exist / are moved / are multiplied -- f16 operands with
// fp32 accumulation, the reference's own fp16 recipe (fp16_util.py:15-32, unet.py:618-632) -- one MFMA per product.
template <int GEO, int R, bool X1>
__global__ __launch_bounds__(256, 2) void conv6_mfma_kernel(Conv6K p) {
    // The body uses the buffer-descriptor builtin type, which only exists in the DEVICE pass of hipcc; in the host pass an
    // (ill-formed) template body would silently drop the kernel's launch stub, so the host pass sees an empty body.
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Geo6<GEO>;
    constexpr int TW = 1 << G::LTW, TH = 1 << G::LTH, TI = G::TI;
    constexpr int LW = TW + 2, LH = TH + 2;
    constexpr int PATCH = TI * LH * LW;                 // patch positions = entries per k-half
    constexpr int NPIECE = (2 * PATCH + 63) / 64;       // 1 KiB DMA pieces per plane (hi or lo)
    constexpr int NXT = (NPIECE + 3) / 4;               // pieces per wave per plane
    constexpr int NPL = X1 ? 1 : 2;                     // operand planes (hi [, lo])
    constexpr int NACT = NPL * NXT;                     // activation DMA instructions per wave per chunk
    constexpr int XB = NPIECE * 1024;                   // bytes per plane buffer
    constexpr int D = R - 1;                            // weight prefetch distance in taps
    constexpr int TAPS = 9;
    constexpr int GPI = (TW * TH) / 64;                 // 64-pixel groups per image inside one tile
    static_assert(NACT <= TAPS, "one activation piece per tap");
    extern __shared__ __attribute__((aligned(16))) char smem6[];
    char* lds_x = smem6;                                // [2 buffers][hi|lo][XB]
    char* lds_w = smem6 + 4 * XB;                       // [4 waves][R slots][hi 1 KiB | lo 1 KiB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2); renumber them so that one XCD
    // owns a contiguous range of tiles (the co-blocks of a pixel tile and neighbouring tiles share an L2)
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int split = bid % p.ksplit;
    bid /= p.ksplit;
    const int co_blk = bid % p.n_co_blocks;
    const int ptile = bid / p.n_co_blocks;
    const int co0 = co_blk * 128 + wave * 32;           // this wave's first output channel
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int img_grp = ptile / tiles_per_img;
    const int trem = ptile - img_grp * tiles_per_img;
    const int ty0 = (trem / p.tiles_x) * TH;
    const int tx0 = (trem % p.tiles_x) * TW;
    const int n0 = img_grp * TI;
    const int HW = p.H * p.W;
    const bool wave_live = co0 < p.Cout;

    // ---- per-lane DMA source offsets of this wave's activation pieces (chunk invariant, BYTES from the first entry of the
    // chunk's first channel group); out-of-image positions get kOutOfRange (hardware zero fill).
    // Piece indices beyond the plane are clamped onto its last piece (same data written twice) so that every wave issues the
    // same number of DMA instructions: the counted vmcnt waits below rely on it.
    unsigned x_off[NXT];
#pragma unroll
    for (int u = 0; u < NXT; ++u) {
        int piece = wave + u * 4;
        if (piece > NPIECE - 1) piece = NPIECE - 1;
        const int f = piece * 64 + lane;
        const int kg = f / PATCH;
        const int e = f - kg * PATCH;
        const int ti = e / (LH * LW);
        const int rr = e - ti * (LH * LW);
        const int hy = rr / LW, hx = rr - hy * LW;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const int n = n0 + ti;
        const bool ok = kg < 2 && n < p.B && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        x_off[u] = ok ? ((unsigned)((n * p.C8 + kg) * HW + gy * p.W + gx) << 4) : kOutOfRange;
    }
    // ---- MFMA operand addressing.  B fragment of pixel tile j, tap (dy, dx): entry lane_b + tile_off(j) + dy * LW + dx
    const int lane_b = (GEO == 0 ? l31 : (GEO == 1 ? (l31 >> 4) * LW + (l31 & 15) : (l31 >> 3) * LW + (l31 & 7))) + half * PATCH;
    auto tile_off = [](int j) constexpr -> int { return GEO == 0 ? j * LW : (GEO == 1 ? 2 * j * LW : (j >> 1) * (LH * LW) + (j & 1) * 4 * LW); };
    const half8* xbase = reinterpret_cast<const half8*>(lds_x) + lane_b;
    char* ring = lds_w + wave * (R * 2048);             // this wave's private weight ring
    const half8* abase = reinterpret_cast<const half8*>(ring) + lane;

    floatx16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int ch_begin = split * p.chunks_per_split;
    const int ch_end = min(p.n_chunks_total, ch_begin + p.chunks_per_split);
    const int n_taps_total = (ch_end - ch_begin) * TAPS;

    // weights of global tap g (counted from this block's first chunk) -> ring slot g % R: two 1 KiB pieces (hi, lo).
    // This wave's records of one chunk are 18 KiB contiguous; the descriptor base moves with the chunk, the tap is the scalar offset.
    const unsigned lane16 = (unsigned)lane * 16u;
    const char* wsrc0 = p.w16 + ((((size_t)ch_begin * p.n_co_blocks + co_blk) * 4 + wave) * TAPS) * 2048;
    const size_t wchunk_stride = (size_t)p.n_co_blocks * 4 * TAPS * 2048;
    auto dma_w = [&](int chunk_rel, int tap, int slot) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rw = rsrc_uniform(wsrc0 + (size_t)chunk_rel * wchunk_stride, TAPS * 2048);
        char* dst = ring + slot * 2048;
        BLDS6(rw, dst, lane16, tap * 2048);
        if (!X1) BLDS6(rw, dst + 1024, lane16, tap * 2048 + 1024);
    };
    // activation piece q (0 .. NACT-1: u = q / 2, plane = q & 1) of chunk `chunk` -> buffer buf.  One chunk = two channel groups
    // of HW entries; the descriptor base moves with the chunk, num_records = the plane (valid offsets never leave it).
    const size_t xplane_bytes = (size_t)p.B * p.C8 * HW * 16;
    auto dma_x = [&](int chunk, int buf, int q) __attribute__((always_inline)) {
        const int u = X1 ? q : q >> 1, plane = X1 ? 0 : q & 1;
        int piece = wave + u * 4;
        if (piece > NPIECE - 1) piece = NPIECE - 1;
        const size_t coff = (size_t)chunk * 2 * HW * 16;
        const __amdgpu_buffer_rsrc_t rx = rsrc_uniform((plane ? p.xlo : p.xhi) + coff, (unsigned)(xplane_bytes - coff));
        BLDS6(rx, lds_x + buf * 2 * XB + plane * XB + piece * 1024, x_off[u], 0);
    };

    // Waves whose 32 output channels lie beyond Cout (Cout = 6 of the output layer: three of four waves) only carry their share
    // of the activation DMA and keep the barrier count: prologue, one per chunk boundary, epilogue.  No weights, no MFMAs.
    if (!wave_live) {
#pragma unroll
        for (int q = 0; q < NACT; ++q) dma_x(ch_begin, 0, q);
        wait_vmcnt<0>();
        __syncthreads();
        int it = 0;
        for (int chunk = ch_begin; chunk + 1 < ch_end; ++chunk, ++it) {
#pragma unroll
            for (int q = 0; q < NACT; ++q) dma_x(chunk + 1, (it & 1) ^ 1, q);
            wait_vmcnt<0>();
            __syncthreads();
        }
        __syncthreads();
        return;
    }

    // ---- prologue: the first chunk's patch and the first D taps of weights, all waited for
#pragma unroll
    for (int q = 0; q < NACT; ++q) dma_x(ch_begin, 0, q);
#pragma unroll
    for (int g = 0; g < D; ++g)
        if (g < n_taps_total) dma_w(g / TAPS, g % TAPS, g % R);
    wait_vmcnt<0>();            // explicit: the compiler does not know that the ds_reads below depend on the LDS-DMA
    __syncthreads();

    half8 a_h[2], a_l[2];            // A fragments of the current / next tap
    half8 b_h[2][2], b_l[2][2];      // B fragments of two pixel tiles, two register sets (one in use, one being filled)
    auto read_a = [&](int slot, int rs) __attribute__((always_inline)) {
        const half8* w = abase + slot * 128;           // 2048 B per slot = 128 entries
        a_h[rs] = w[0];
        if (!X1) a_l[rs] = w[64];
    };
    auto read_b = [&](int buf, int tap, int grp, int set) __attribute__((always_inline)) {      // pixel tiles 2 grp, 2 grp + 1
        const half8* xh = xbase + buf * (2 * XB / 16);
        const half8* xl = xh + XB / 16;
        const int toff = (tap / 3) * LW + (tap % 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = tile_off(grp * 2 + j) + toff;
            b_h[set][j] = xh[o];
            if (!X1) b_l[set][j] = xl[o];
        }
    };
    auto mfma_group = [&](int grp, int set, int rs) __attribute__((always_inline)) {
        // the three partial products of one accumulator are issued two MFMAs apart; small terms first
        if (!X1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[grp * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[rs], b_h[set][j], acc[grp * 2 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[grp * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[rs], b_l[set][j], acc[grp * 2 + j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[grp * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[rs], b_h[set][j], acc[grp * 2 + j], 0, 0, 0);
    };

    read_a(0, 0); read_b(0, 0, 0, 0);

    // One K chunk.  MORE: a further chunk follows (its patch and the weights D taps ahead are prefetched).
    // A tap = four groups of two pixel tiles (6 MFMAs each); the operands of the next group are requested before the MFMAs of
    // the current one are issued, into the other register set.
    // DMA group of tap i: [activation piece i of the next chunk, if i < NACT] + [the 2 weight pieces of tap i + D]; the weights
    // of tap t+1 are therefore followed, in issue order, by the groups of taps t+2-D .. t, whose sizes are compile-time
    // constants: that sum is the vmcnt that proves the tap t+1 pieces have landed (vector-memory loads complete in order).
    auto chunk_body = [&](auto more_c, int chunk, int it) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_c)::value;
        const int cur = it & 1;
        const int rbase = (it * TAPS) % R;              // ring slot of this chunk's tap 0
        static_for<0, TAPS>([&](auto tap_c) __attribute__((always_inline)) {
            constexpr int tap = decltype(tap_c)::value;
            constexpr int rs = tap & 1;
            read_b(cur, tap, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (MORE && tap < NACT) dma_x(chunk + 1, cur ^ 1, tap);
            if (MORE || tap + D < TAPS) {
                const int g = tap + D;
                dma_w(it + g / TAPS, g % TAPS, (rbase + g) % R);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(0, 0, rs);
            __builtin_amdgcn_sched_barrier(0);
            read_b(cur, tap, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(1, 1, rs);
            __builtin_amdgcn_sched_barrier(0);
            read_b(cur, tap, 3, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(2, 0, rs);
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < TAPS) {
                wait_vmcnt<c6_wait_n(MORE, NACT, D, tap, NPL)>();
                read_a((rbase + tap + 1) % R, (tap + 1) & 1);
                read_b(cur, tap + 1, 0, 0);
            } else if (MORE) {
                // chunk boundary.  Before the barrier: this wave's activation pieces of the next chunk have landed (they
                // are followed by their group's weights and the groups of taps NACT .. 8).  The barrier (with its lgkmcnt(0))
                // then says every wave is done reading the current patch and writing / receiving the next one.  After it: the
                // weights of the next chunk's tap 0 (followed by the groups of taps 11-D .. 8) -- an EXPLICIT wait: the compiler
                // does not know that a ds_read depends on an LDS-DMA and may leave vmcnt out of the barrier's wait.
                wait_vmcnt<NPL + NPL * (TAPS - NACT)>();
                if (LDS_BARRIER) barrier_lds_only(); else __syncthreads();
                wait_vmcnt<c6_wait_n(true, NACT, D, TAPS - 1, NPL)>();
                read_a((rbase + TAPS) % R, 1);
                read_b(cur ^ 1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(3, 1, rs);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (MORE) { a_h[0] = a_h[1]; if (!X1) a_l[0] = a_l[1]; }     // 9 taps: the next chunk's tap 0 uses register set 0 again
    };
    {   // all chunks but the last in a loop with ONE body (a branch between two bodies inside the loop keeps the register
        // coalescer from unifying the accumulators across the back edge: two live copies of 128 registers), then the last
        int it = 0, chunk = ch_begin;
        for (; chunk + 1 < ch_end; ++chunk, ++it) chunk_body(std::true_type{}, chunk, it);
        chunk_body(std::false_type{}, chunk, it);
    }

    // ---- epilogue: per wave, four passes of 32 co x 64 px through a private LDS slab (transposition to float4 rows of 4
    // consecutive pixels), bias / residual / GroupNorm partial sums fused.  The other workgroup on this CU keeps the matrix
    // pipe busy meanwhile.
    // STRAIGHT-LINE code (r3).  The first version predicated every residual load and every store with a branch and loaded the bias
    // inside the store loop; the compiler's waitcnt pass gives up counting across those joins and put `s_waitcnt vmcnt(0)` behind
    // every single residual load and in front of every store: 8 serialised load latencies + 8 store round trips per pass, 64 per
    // tile, as long as the tile's whole MFMA phase.  Here every access goes through a buffer descriptor with the predicate folded
    // into the offset (out of range: loads return 0, stores are dropped), the bias is loaded once, and the residual rows of pass
    // q + 1 are requested BEFORE the stores of pass q, so that waiting for them never waits for a store.
    __syncthreads();                                         // all waves are done with the operand buffers
    constexpr int TS = 68;                                   // slab row stride in floats (16-byte aligned, bank-skewed)
    float* tr = reinterpret_cast<float*>(smem6) + wave * (32 * TS);
    const int q4 = lane & 15, rsub = lane >> 4;
    const bool single = p.ksplit == 1;
    float* dst = single ? p.out : p.partial + (size_t)split * ((size_t)p.B * p.Cout * HW);
    const float osc = p.out_scale_dev ? p.out_scale * p.out_scale_dev[0] : p.out_scale;
    const bool do_stat = p.stat != nullptr && single;
    const int res_mode = (single && p.res) ? p.res_mode : -1;
    const size_t img0 = (size_t)n0 * p.Cout;                 // first (image, channel) plane of the tile
    const size_t res_plane = res_mode == 1 ? (size_t)(HW >> 2) : (res_mode == 2 ? (size_t)HW * 4 : (size_t)HW);
    // per-lane offsets are relative to the tile's first image and stay far below 4 GiB; the descriptors only exist for the
    // out-of-range semantics, so their size is "everything" (enabled) or 0 (that operand does not exist: zeros / no store)
    // (the descriptors are rebuilt from scalars right where they are used: one that lives across the branches below ends up in
    // vector registers and every buffer operation in a readfirstlane "waterfall" loop)
    float* const out_base = dst + img0 * HW;
    const float* const res_base = res_mode >= 0 ? p.res + img0 * res_plane : p.bias;
    const unsigned res_bytes = res_mode >= 0 ? 0xFFFFFFFFu : 0u;
    const void* const stat_base = do_stat ? (const void*)(p.stat + img0 * p.stat_slots) : (const void*)p.bias;
    const unsigned stat_bytes = do_stat ? 0xFFFFFFFFu : 0u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    auto f4 = [](u32x4 v) { return make_float4(as_f32(v.x), as_f32(v.y), as_f32(v.z), as_f32(v.w)); };

    float bv[8];
    const __amdgpu_buffer_rsrc_t r_bias = rsrc_uniform(p.bias, single ? (unsigned)p.Cout * 4u : 0u);
#pragma unroll
    for (int it = 0; it < 8; ++it) bv[it] = as_f32(__builtin_amdgcn_raw_buffer_load_b32(r_bias, (unsigned)(co0 + it * 4 + rsub) * 4u, 0, 0));

    // geometry of pass q for this lane: 4 consecutive pixels of one row
    struct PassGeo { int ti, y, x; bool pok; unsigned pix; };
    auto geo = [&](int q) {
        const int pp = q * 64 + q4 * 4;
        PassGeo g;
        g.ti = pp >> (G::LTW + G::LTH);
        g.y = ty0 + ((pp >> G::LTW) & (TH - 1));
        g.x = tx0 + (pp & (TW - 1));
        g.pok = n0 + g.ti < p.B && g.y < p.H && g.x < p.W;
        g.pix = (unsigned)(g.y * p.W + g.x);
        return g;
    };
    // residual rows of pass q (all three forms reduced to one float4 per channel row)
    auto load_res = [&](int q, float4 (&rv)[8]) __attribute__((always_inline)) {
        const PassGeo g = geo(q);
        const __amdgpu_buffer_rsrc_t r_res = rsrc_uniform(res_base, res_bytes);
        if (res_mode <= 0) {                               // same shape as the output (or none: zero-sized descriptor)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int co = co0 + it * 4 + rsub;
                const unsigned off = (g.pok && co < p.Cout) ? ((unsigned)(g.ti * p.Cout + co) * (unsigned)HW + g.pix) * 4u : kOutOfRange;
                rv[it] = f4(__builtin_amdgcn_raw_buffer_load_b128(r_res, off, 0, 0));
            }
        } else if (res_mode == 1) {                        // residual at half resolution, nearest up-sampling (unet.py:107)
            const unsigned Wr = (unsigned)(p.W >> 1), HWr = (unsigned)(HW >> 2);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int co = co0 + it * 4 + rsub;
                const unsigned off = (g.pok && co < p.Cout) ? ((unsigned)(g.ti * p.Cout + co) * HWr + (unsigned)(g.y >> 1) * Wr + (unsigned)(g.x >> 1)) * 4u : kOutOfRange;
                const u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(r_res, off, 0, 0);
                const float a = as_f32(r2.x), b = as_f32(r2.y);
                rv[it] = make_float4(a, a, b, b);
            }
        } else {                                           // residual at double resolution, 2x2 average pooling (unet.py:136)
            const unsigned Wr = (unsigned)p.W * 2u, HWr = (unsigned)HW * 4u;
#pragma unroll
            for (int h = 0; h < 4; ++h) {                  // two channel rows (8 loads) at a time: register pressure
                float4 a0[2], a1[2], b0[2], b1[2];
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    const int co = co0 + (h * 2 + i2) * 4 + rsub;
                    const bool ok = g.pok && co < p.Cout;
                    const unsigned off = ((unsigned)(g.ti * p.Cout + co) * HWr + (unsigned)(2 * g.y) * Wr + (unsigned)(2 * g.x)) * 4u;
                    a0[i2] = f4(__builtin_amdgcn_raw_buffer_load_b128(r_res, ok ? off : kOutOfRange, 0, 0));
                    a1[i2] = f4(__builtin_amdgcn_raw_buffer_load_b128(r_res, ok ? off + 16u : kOutOfRange, 0, 0));
                    b0[i2] = f4(__builtin_amdgcn_raw_buffer_load_b128(r_res, ok ? off + Wr * 4u : kOutOfRange, 0, 0));
                    b1[i2] = f4(__builtin_amdgcn_raw_buffer_load_b128(r_res, ok ? off + Wr * 4u + 16u : kOutOfRange, 0, 0));
                }
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2)
                    rv[h * 2 + i2] = make_float4(((a0[i2].x + a0[i2].y) + (b0[i2].x + b0[i2].y)) * 0.25f, ((a0[i2].z + a0[i2].w) + (b0[i2].z + b0[i2].w)) * 0.25f,
                                                 ((a1[i2].x + a1[i2].y) + (b1[i2].x + b1[i2].y)) * 0.25f, ((a1[i2].z + a1[i2].w) + (b1[i2].z + b1[i2].w)) * 0.25f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // (one straight-line instance of the passes per residual form would let the waitcnt pass count exactly -- the join of the three
    // forms makes the first store of a pass wait for the previous pass's stores -- but it costs > 30 spilled registers: not taken)
    float4 rv[2][8];
    load_res(0, rv[0]);
    static_for<0, 4>([&](auto q_c) __attribute__((always_inline)) {
        constexpr int q = decltype(q_c)::value;
        const PassGeo g = geo(q);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tr[((r & 3) + 8 * (r >> 2) + 4 * half) * TS + jj * 32 + l31] = acc[q * 2 + jj][r] * osc;
        // wave-private slab: program order + the compiler's lgkmcnt waits are all the synchronisation needed
        if (q + 1 < 4) load_res(q + 1, rv[(q + 1) & 1]);   // requested before this pass's stores
        const int slot = trem * GPI + (q % GPI);
        const __amdgpu_buffer_rsrc_t r_out = rsrc_uniform(out_base, 0xFFFFFFFFu);
        const __amdgpu_buffer_rsrc_t r_stat = rsrc_uniform(stat_base, stat_bytes);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int co_l = it * 4 + rsub;
            const int co = co0 + co_l;
            const bool ok = g.pok && co < p.Cout;
            float4 v = *reinterpret_cast<const float4*>(tr + co_l * TS + q4 * 4);
            if (single) {
                v.x += bv[it]; v.y += bv[it]; v.z += bv[it]; v.w += bv[it];
                const float4 r4 = rv[q & 1][it];
                v.x = r4.x + v.x; v.y = r4.y + v.y; v.z = r4.z + v.z; v.w = r4.w + v.w;
            }
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            const unsigned plane_l = (unsigned)(g.ti * p.Cout + co);
            u32x4 sv;
            sv.x = as_u32(v.x); sv.y = as_u32(v.y); sv.z = as_u32(v.z); sv.w = as_u32(v.w);
            __builtin_amdgcn_raw_buffer_store_b128(sv, r_out, ok ? (plane_l * (unsigned)HW + g.pix) * 4u : kOutOfRange, 0, 0);
            if (do_stat) {
                float s1 = (v.x + v.y) + (v.z + v.w);
                float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                // inclusive scan over the 16-lane DPP row (row_shr 1, 2, 4, 8, zero fill): lane 15 of the row ends with the total
                s1 += dpp_row_shr6<0x111>(s1); s2 += dpp_row_shr6<0x111>(s2);
                s1 += dpp_row_shr6<0x112>(s1); s2 += dpp_row_shr6<0x112>(s2);
                s1 += dpp_row_shr6<0x114>(s1); s2 += dpp_row_shr6<0x114>(s2);
                s1 += dpp_row_shr6<0x118>(s1); s2 += dpp_row_shr6<0x118>(s2);
                // a 64-pixel group belongs to ONE image (GPI groups per image and tile); groups wholly outside the image store 0
                u32x2 st;
                st.x = as_u32(s1); st.y = as_u32(s2);
                const bool wr = q4 == 15 && co < p.Cout && n0 + g.ti < p.B;
                __builtin_amdgcn_raw_buffer_store_b64(st, r_stat, wr ? (plane_l * (unsigned)p.stat_slots + (unsigned)slot) * 8u : kOutOfRange, 0, 0);
            }
        }
    });
#endif
}

// Deterministic split-K combine, one wave per (image, channel) plane: out = sum_s partial[s] (in order) + bias + residual,
// and -- fused -- the fp64 GroupNorm statistics of the plane just written (the same {sum, sum of squares} record that
// gn_stats_kernel produces), so split-K layers need no separate statistics pass.
__global__ __launch_bounds__(256) void conv6_reduce_kernel(const float* partial, int ksplit, const float* bias, const float* res, int res_mode,
                                                           float* out, int Cout, int H, int W, int planes, double2* stat) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const int HW = H * W;
    const size_t total = (size_t)planes * HW;
    const int co = plane % Cout;
    const float bv = bias[co];
    double s = 0.0, ss = 0.0;
    for (int i4 = lane; i4 < (HW >> 2); i4 += 64) {
        const size_t o = (size_t)plane * HW + (size_t)i4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; k0 < ksplit; k0 += 4) {       // four slabs requested together, added in slab order: same bits, a quarter of the round trips
            float4 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = *reinterpret_cast<const float4*>(partial + (size_t)min(k0 + j, ksplit - 1) * total + o);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j < ksplit) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
        }
        v.x += bv; v.y += bv; v.z += bv; v.w += bv;
        if (res) {
            const int r = i4 * 4;
            const int y = r / W, x = r - y * W;
            if (res_mode == 0) {
                const float4 t = *reinterpret_cast<const float4*>(res + o);
                v.x = t.x + v.x; v.y = t.y + v.y; v.z = t.z + v.z; v.w = t.w + v.w;
            } else if (res_mode == 1) {
                const int Hr = H >> 1, Wr = W >> 1;
                const float2 t = *reinterpret_cast<const float2*>(res + (size_t)plane * (Hr * Wr) + (y >> 1) * Wr + (x >> 1));
                v.x = t.x + v.x; v.y = t.x + v.y; v.z = t.y + v.z; v.w = t.y + v.w;
            } else {
                const int Wr = W * 2;
                const float* rp = res + (size_t)plane * (4 * (size_t)HW) + (size_t)(2 * y) * Wr + 2 * x;
                const float4 a0 = *reinterpret_cast<const float4*>(rp), a1 = *reinterpret_cast<const float4*>(rp + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(rp + Wr), b1 = *reinterpret_cast<const float4*>(rp + Wr + 4);
                v.x = ((a0.x + a0.y) + (b0.x + b0.y)) * 0.25f + v.x;
                v.y = ((a0.z + a0.w) + (b0.z + b0.w)) * 0.25f + v.y;
                v.z = ((a1.x + a1.y) + (b1.x + b1.y)) * 0.25f + v.z;
                v.w = ((a1.z + a1.w) + (b1.z + b1.w)) * 0.25f + v.w;
            }
        }
        *reinterpret_cast<float4*>(out + o) = v;
        s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    if (stat) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
        if (lane == 0) stat[plane] = make_double2(s, ss);
    }
}

const float* conv_zero_page();

static int conv6_geo(int H, int W) {
    if ((W & 3) || W < 8 || H < 8) return -1;
    return W >= 32 ? 0 : (W >= 16 ? 1 : 2);
}
bool conv6_supported(int H, int W) { return conv6_geo(H, W) >= 0; }
// Conv6Emit: whole K (>= 384 workgroups, launch_conv6's rule), 8 x 32 tiles inside the image, full 128-channel blocks, and at most 256
// workgroups per (image, co-block) -- the set that waits for one another, consecutive in dispatch order -- so that it is resident
// together with room to spare (the chip holds 512)
bool conv7_emit_supported(int B, int Cout, int H, int W) {
    if (conv6_geo(H, W) != 0 || (W & 31) || (H & 7) || (Cout & 127)) return false;
    // the emission folds per-group sums inside one wave: channels per group (Cout / 32) must divide the 64 channels a wave owns, i.e. Cout in
    // {128, 256, 512, 1024, 2048}; 384 / 640 / 768 / 896 (channel_mult x3, x5, x6, x7) have groups that straddle waves -> unfused path
    if (64 % (Cout / 32) != 0) return false;
    const int tiles = (W / 32) * (H / 8);
    return tiles <= conv7_emit_capacity() / 2 && tiles * (Cout / 128) * B >= 384;
}

// statistics slots per (image, channel) plane written by the epilogue when no split-K is used
int conv6_stat_slots(int H, int W) {
    const int g = conv6_geo(H, W);
    if (g < 0) return 0;
    const int tw = g == 0 ? 32 : (g == 1 ? 16 : 8), th = g == 0 ? 8 : (g == 1 ? 16 : 8);
    return ((W + tw - 1) / tw) * ((H + th - 1) / th) * ((tw * th) / 64);
}

template <int GEO, int R, bool X1>
static Status launch6(hipStream_t s, Conv6K k, int blocks) {
    using G = Geo6<GEO>;
    constexpr int PATCH = G::TI * ((1 << G::LTH) + 2) * ((1 << G::LTW) + 2);
    constexpr int NPIECE = (2 * PATCH + 63) / 64;
    constexpr size_t LDS = (size_t)4 * NPIECE * 1024 + (size_t)4 * R * 2048;
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
    auto fn = conv6_mfma_kernel<GEO, R, X1>;
    static LdsAttrOnce attr_set;
    DPIR_HIP(attr_set.set(reinterpret_cast<const void*>(fn), (int)LDS));
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), LDS, s, k);
    return Status{};
}

// stat_kind_out: 0 none, 1 epilogue slots (a.stat filled, conv6_stat_slots entries per plane), 2 per-plane fp64 records (a.stat_plane)
Status launch_conv6_resolve(hipStream_t s, const PendingConv& p) {
    const int planes = p.B * p.Cout;
    hipLaunchKernelGGL(conv6_reduce_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, s, p.partial, p.ksplit, p.bias, p.res, p.res_mode, p.out,
                       p.Cout, p.H, p.W, planes, p.stat_plane);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

Status launch_conv6(hipStream_t s, const Conv6Args& a, int* stat_kind_out, PendingConv* pend_out) {
    if (stat_kind_out) *stat_kind_out = 0;
    if (pend_out) *pend_out = PendingConv{};
    const int geo = conv6_geo(a.H, a.W);
    if (geo < 0) return Status{DPIR_ERR_UNSUPPORTED, "conv6: shape not tiled"};
    Conv6K k;
    k.xhi = reinterpret_cast<const char*>(a.xhi); k.xlo = reinterpret_cast<const char*>(a.xlo);
    k.w16 = reinterpret_cast<const char*>(a.w16); k.bias = a.bias; k.out = a.out; k.res = a.res; k.res_mode = a.res_mode;
    k.B = a.B; k.Cout = a.Cout; k.H = a.H; k.W = a.W;
    k.n_chunks_total = (a.Cin + 15) / 16;
    k.C8 = 2 * k.n_chunks_total;          // act.hip pads the blocked tensor to whole 16-channel chunks
    k.partial = a.partial; k.ksplit = 1; k.chunks_per_split = 0;
    k.out_scale = 1.0f / a.w16_scale;
    k.out_scale_dev = a.out_scale_dev;
    k.zeros = conv_zero_page();
    if (!k.zeros) return Status{DPIR_ERR_NOMEM, "conv6: cannot allocate the zero page"};
    const int tw = geo == 0 ? 32 : (geo == 1 ? 16 : 8), th = geo == 0 ? 8 : (geo == 1 ? 16 : 8), ti = geo == 2 ? 4 : 1;
    k.tiles_x = (a.W + tw - 1) / tw;
    k.tiles_y = (a.H + th - 1) / th;
    const int n_ptiles = k.tiles_x * k.tiles_y * ((a.B + ti - 1) / ti);
    k.n_co_blocks = (a.Cout + 127) / 128;
    const int chunks = k.n_chunks_total;
    const int blocks = n_ptiles * k.n_co_blocks;
    // split-K when the launch cannot give every CU ONE workgroup (low-resolution layers); deterministic slabs.  Every slice writes a full fp32
    // slab, so splitting trades co-resident workgroups for slab traffic.  Rounds 2-4 split below 384 workgroups up to ~512 ("two per CU");
    // measured in round 5 (profiles/r05/split_rule_layer_roofline_and_forward_ab.log, five settings interleaved in one call): splitting only
    // below 256 up to ~256 is faster on every affected shape -- 512 -> 512 @ 32^2 215 -> 185 us, 256 -> 256 @ 32^2 65 -> 60, 512 -> 512 @ 16^2
    // 64 -> 58.5, 1024 -> 512 @ 16^2 108 -> 99 -- and 18.88 -> 18.73 ms per forward; larger targets (768, 1024) lose 0.7 ms.  The two constants stay
    // read-once environment knobs for A/B runs (tools/layer_roofline.py, tools/forward_time.py).
    static const int split_below = getenv("DPIR_SPLIT_BELOW") ? atoi(getenv("DPIR_SPLIT_BELOW")) : 256;
    static const int split_target = getenv("DPIR_SPLIT_TARGET") ? atoi(getenv("DPIR_SPLIT_TARGET")) : 256;
    int S = 1;
    if (k.partial && blocks < split_below) {
        S = (split_target + blocks - 1) / blocks;
        if (S > chunks / 2) S = chunks / 2;
        if (S > 16) S = 16;
        if (S < 1) S = 1;
        if ((size_t)S * k.B * k.Cout * k.H * k.W > a.partial_capacity) S = 1;
    }
    k.chunks_per_split = (chunks + S - 1) / S;
    S = (chunks + k.chunks_per_split - 1) / k.chunks_per_split;      // no empty slice: every workgroup owns at least one chunk
    k.ksplit = S;
    if (S == 1) k.partial = nullptr;
    k.stat = nullptr; k.stat_slots = 0;
    if (a.emit) {
        if (S != 1 || !conv7_emit_supported(a.B, a.Cout, a.H, a.W) || a.res || a.out_scale_dev) return invalid("conv6: this launch cannot emit the next convolution's planes");
        k.em = *a.emit;
        DPIR_TRY(launch_conv7(s, k, blocks, a.x1));
        DPIR_HIP(hipGetLastError());
        return Status{};
    }
    if (a.stat && S == 1) {
        k.stat = a.stat; k.stat_slots = conv6_stat_slots(a.H, a.W);
        if (stat_kind_out) *stat_kind_out = 1;
    }
    // conv7 (csrc/conv7.hip: same results bit for bit) is the 3x3 kernel; conv6 keeps the two launch classes of the 8 x 32 geometry it
    // is measurably faster at (profiles/r04/conv7x_check.log): split-K launches (x1.07) and a last co-block with at most 64 live
    // channels (x1.05) -- unless the whole launch has at most 32 output channels (the 128 -> 6 output convolution), which conv7's
    // NARROW variant spreads over all four waves.  force_kernel (tests): 6 = conv6 (geometry 0 only), 7 = conv7.
    const bool idle_half = (a.Cout & 127) != 0 && (a.Cout & 127) <= 64 && a.Cout > 32;
    const bool use6 = a.force_kernel == 6 || (a.force_kernel != 7 && geo == 0 && (S > 1 || idle_half));
    if (use6 && geo != 0) return invalid("conv6 is built for the 8 x 32 geometry only (conv7 has the 16 x 16 and 8 x 8 ones)");
    if (!use6) DPIR_TRY(launch_conv7(s, k, blocks * S, a.x1));
    else if (a.x1) DPIR_TRY((launch6<0, 4, true>(s, k, blocks * S)));
    else DPIR_TRY((launch6<0, 4, false>(s, k, blocks * S)));
    if (S > 1) {
        PendingConv pc;
        pc.partial = k.partial; pc.ksplit = S; pc.bias = k.bias; pc.res = k.res; pc.res_mode = k.res_mode; pc.out = k.out;
        pc.B = k.B; pc.Cout = k.Cout; pc.H = k.H; pc.W = k.W; pc.stat_plane = a.stat_plane;
        if (pend_out) {
            *pend_out = pc;
            if (stat_kind_out) *stat_kind_out = 3;
        } else {
            DPIR_TRY(launch_conv6_resolve(s, pc));
            if (a.stat_plane && stat_kind_out) *stat_kind_out = 2;
        }
    }
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// Host: OIHW fp32 -> [chunk (16 ci)][co-block (128)][wave (32 co)][tap][hi|lo][k-half][32 co][8 ci] f16: one 2 KiB record per
// (wave, tap), its two 1 KiB halves in MFMA A-fragment lane order (lane = k-half * 32 + co).  Scaled by a power of two so
// that max|w| * scale is in [512, 1024) (keeps the low halves out of the f16 subnormal range); returns the scale.
float pack_weights_conv6(const float* w, int cout, int cin, std::vector<uint16_t>& out) {
    const int taps = 9;
    const int chunks = (cin + 15) / 16, cblocks = (cout + 127) / 128;
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin * taps; ++i) mx = fmaxf(mx, fabsf(w[i]));
    float scale = 1.0f;
    if (mx > 0.f) scale = exp2f(floorf(log2f(1024.0f / mx)));
    while (mx * scale >= 1024.0f) scale *= 0.5f;
    out.assign((size_t)chunks * cblocks * 4 * taps * 1024, 0);       // 1024 halves = 2 KiB per record
    for (int ch = 0; ch < chunks; ++ch)
        for (int cbk = 0; cbk < cblocks; ++cbk)
            for (int wv = 0; wv < 4; ++wv)
                for (int tap = 0; tap < taps; ++tap) {
                    uint16_t* hi = out.data() + ((((size_t)ch * cblocks + cbk) * 4 + wv) * taps + tap) * 1024;
                    uint16_t* lo = hi + 512;
                    for (int kh = 0; kh < 2; ++kh)
                        for (int col = 0; col < 32; ++col)
                            for (int j = 0; j < 8; ++j) {
                                const int co = cbk * 128 + wv * 32 + col, ci = ch * 16 + kh * 8 + j;
                                const float v = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * taps + tap] * scale : 0.f;
                                const _Float16 h = (_Float16)v;
                                const _Float16 l = (_Float16)(v - (float)h);
                                const size_t o = ((size_t)kh * 32 + col) * 8 + j;
                                __builtin_memcpy(&hi[o], &h, 2);
                                __builtin_memcpy(&lo[o], &l, 2);
                            }
                }
    return scale;
}

}  // namespace dpir
