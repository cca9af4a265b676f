// conv7: the 3x3 convolution of the f16x3 / f16x1 modes (round 3: geometry 0 only; round 4: every case -- the 16 x 16 and 4-image 8 x 8
// geometries, split-K partial slabs, f16x1, the run-time output scale of dgrad, idle co-halves).
//
// Same arithmetic, operand planes, weight pack and workgroup tile as conv6 (128 output channels x 256 pixels; f16x3: al*bh, ah*bl,
// ah*bh per product, in that order per accumulator -> outputs, fused GroupNorm sums and split-K slabs are BIT-IDENTICAL to conv6's:
// 16 shape x residual x mode cases, profiles/r04/conv7x_check.log), but inside the workgroup
//   conv6: a wave owns 32 output channels x all 256 pixels  -> per tap 2 A + 16 B fragment reads from LDS for 24 MFMAs, and the
//          weights (private to the wave) travel L2 -> LDS ring -> registers although no other wave reads them;
//   conv7: a wave owns 64 output channels x 128 pixels      -> per tap 8 B fragment reads from LDS for 24 MFMAs; the 4 A fragments
//          (2 co-tiles x hi / lo) are loaded STRAIGHT into registers (buffer_load_dwordx4: 16 B per lane IS the fragment, the host
//          pack is in lane order already), two taps ahead, in a three-set register ring (9 taps % 3 == 0: the ring index is a
//          compile-time constant of the unrolled chunk body).  LDS reads per MFMA 18/24 -> 8/24, LDS footprint 76 -> 44 KiB.
//   Cost: the two waves that share a co-half request the same weights (L2 -> CU weight traffic doubles, all L2 hits) and the ring
//   takes ~48 VGPRs (214-234 in all, still two workgroups per CU).
// Measured back to back against conv6 on one box (profiles/r04/conv7x_check.log): x1.03-1.06 at 256^2 / 128^2, x1.03 in f16x1 and with
// the dgrad scale, x1.01-1.08 at 16 x 16 / 8 x 8 -- and x0.93 for the split-K launches of the 8 x 32 geometry, x0.95 for the 128 -> 6
// output convolution (one live co-tile of four): those two launch classes stay on conv6 (launch_conv6 decides).  In the loop
// (bench.py, DPIR_CONV7=1/0 interleaved in one call, profiles/r04/bench_ab_conv7_in_one_call.log): 8.37 / 8.32 vs 8.19 / 8.21 images/s (the round-3 kernel on / off).
// PMC (profiles/r04): matrix pipe busy 78.1 % of the cycles at an effective 1.67 GHz (conv6: 74.8 % at 1.62 GHz).
#include "common.h"
#include <atomic>
#include "elem.h"
#include "lds_dma.h"
#include "conv6_params.h"
#include <type_traits>

namespace dpir {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for7(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for7<I + 1, N>(f);
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_row_shr7(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <int GEO> struct Geo7;
template <> struct Geo7<0> { static constexpr int LTW = 5, LTH = 3, TI = 1; };   // 8 rows x 32 columns
template <> struct Geo7<1> { static constexpr int LTW = 4, LTH = 4, TI = 1; };   // 16 x 16
template <> struct Geo7<2> { static constexpr int LTW = 3, LTH = 3, TI = 4; };   // 4 images x 8 x 8

// NARROW (8 x 32 geometry): a launch with at most 32 output channels (the 128 -> 6 output convolution, the 128 -> 3 dgrad of conv_in).  One
// live co-tile: instead of two waves computing a dead second co-tile and two waves idling, all four waves take that co-tile for a
// quarter of the pixels each (wave tile 32 co x 64 px, 6 MFMAs per tap instead of 24 on half the waves).
// EMIT (8 x 32 geometry, whole K, full 128-channel blocks, tiles inside the image): the fused hop to the next convolution of a ResBlock,
// see Conv6Emit.  Everything up to the epilogue is the same kernel; workgroups keep their natural order (image-major), so that the
// <= 256 workgroups of an image are dispatched together and the wait below always ends.
__device__ __forceinline__ float silu7(float v) {          // act.hip's silu_a
    float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int GEO, bool X1, bool NARROW, bool EMIT>
__global__ __launch_bounds__(256, 2) void conv7_mfma_kernel(Conv6K p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Geo7<GEO>;
    constexpr int TW = 1 << G::LTW, TH = 1 << G::LTH, TI = G::TI, LW = TW + 2, LH = TH + 2;
    constexpr int PATCH = TI * LH * LW;                 // entries per k-half
    constexpr int NPIECE = (2 * PATCH + 63) / 64;       // one-KiB DMA pieces per plane
    constexpr int NXT = (NPIECE + 3) / 4;               // per wave and plane
    constexpr int NPL = X1 ? 1 : 2;                     // operand planes (hi [, lo])
    constexpr int NACT = NPL * NXT;                     // activation DMA instructions per wave and chunk
    static_assert(NACT <= 8, "the activation pieces must be older than the weights of taps 7 and 8");
    constexpr int XB = NPIECE * 1024;
    constexpr int TAPS = 9;
    constexpr int CT = NARROW ? 1 : 2;                  // co-tiles (32 channels) per wave
    constexpr int TPG = NARROW ? 1 : 2;                 // pixel tiles (32 pixels) per group; a wave has two groups
    static_assert(!NARROW || GEO == 0, "the narrow variant exists for the 8 x 32 geometry");
    static_assert(!EMIT || (GEO == 0 && !NARROW), "fused emission exists for the 8 x 32 geometry");
    extern __shared__ __attribute__((aligned(16))) char smem7[];      // [2 buffers][hi|lo][XB]; the epilogue slabs alias it

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // co half (64 channels), pixel half (rows 4 pw .. 4 pw + 3); NARROW: one co-tile, pixel quarter (rows 2 pw, 2 pw + 1)
    const int cw = NARROW ? 0 : wave & 1, pw = NARROW ? wave : wave >> 1;
    const int l31 = lane & 31;
    const int half = lane >> 5;

    int bid = blockIdx.x;
    if (!EMIT && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);      // XCD-contiguous tiles, as conv6
    const int split = bid % p.ksplit;
    bid /= p.ksplit;
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    // EMIT: inside an image the co-block is the OUTER index, so that the workgroups that wait for one another -- one (image, co-block):
    // GroupNorm groups never straddle a 128-channel block -- are tiles_per_img (<= 256) CONSECUTIVE ids, whatever the channel count
    const int per_img = tiles_per_img * p.n_co_blocks;
    const int co_blk = EMIT ? (bid % per_img) / tiles_per_img : bid % p.n_co_blocks;
    const int ptile = EMIT ? (bid / per_img) * tiles_per_img + (bid % per_img) % tiles_per_img : bid / p.n_co_blocks;
    const int img_grp = ptile / tiles_per_img;
    const int n0 = img_grp * TI;
    const int trem = ptile - img_grp * tiles_per_img;
    const int ch_begin = split * p.chunks_per_split;
    const int ch_end = min(p.n_chunks_total, ch_begin + p.chunks_per_split);
    const int co_wave = co_blk * 128 + cw * 64;              // this wave's first output channel
    const bool wave_live = co_wave < p.Cout;
    const int ty0 = (trem / p.tiles_x) * TH;
    const int tx0 = (trem % p.tiles_x) * TW;
    const int HW = p.H * p.W;

    // ---- activation DMA: identical to conv6 (pieces dealt to the 4 waves, out-of-image positions out of range = zeros)
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
    const size_t xplane_bytes = (size_t)p.B * p.C8 * HW * 16;
    auto dma_x = [&](int chunk, int buf, int q) __attribute__((always_inline)) {
        const int u = X1 ? q : q >> 1, plane = X1 ? 0 : q & 1;
        int piece = wave + u * 4;
        if (piece > NPIECE - 1) piece = NPIECE - 1;
        const size_t coff = (size_t)chunk * 2 * HW * 16;
        const __amdgpu_buffer_rsrc_t rx = rsrc_uniform((plane ? p.xlo : p.xhi) + coff, (unsigned)(xplane_bytes - coff));
        BLDS6(rx, smem7 + buf * 2 * XB + plane * XB + piece * 1024, x_off[u], 0);
    };

    // ---- B fragments: this wave's pixel tiles are 4 pw .. 4 pw + 3 of conv6's eight (tile_off is additive in the wave part)
    auto tile_off = [](int j) constexpr -> int { return GEO == 0 ? j * LW : (GEO == 1 ? 2 * j * LW : (j >> 1) * (LH * LW) + (j & 1) * 4 * LW); };
    const int wave_off = NARROW ? 2 * pw * LW : (GEO == 0 ? 4 * pw * LW : (GEO == 1 ? 8 * pw * LW : 2 * pw * (LH * LW)));
    const int lane_b = (GEO == 0 ? l31 : (GEO == 1 ? (l31 >> 4) * LW + (l31 & 15) : (l31 >> 3) * LW + (l31 & 7))) + half * PATCH + wave_off;
    const half8* xbase = reinterpret_cast<const half8*>(smem7) + lane_b;

    // ---- A fragments straight from the weight pack: record (chunk, co_blk, co-tile ct, tap) = 2 KiB [hi | lo], 16 B per lane
    const unsigned lane16 = (unsigned)lane * 16u;
    half8 a_h[3][CT], a_l[3][CT];
    auto load_a = [&](int chunk, int tap, int slot) __attribute__((always_inline)) {
        const char* base = p.w16 + ((size_t)chunk * p.n_co_blocks + co_blk) * (4 * TAPS * 2048);
        const __amdgpu_buffer_rsrc_t rw = rsrc_uniform(base, 4 * TAPS * 2048);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const unsigned so = (unsigned)(((2 * cw + i) * TAPS + tap) * 2048);
            a_h[slot][i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, so, 0));
            if (!X1) a_l[slot][i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, so + 1024u, 0));
        }
    };

    floatx16 acc[CT][2 * TPG];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TPG; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    half8 b_h[2][TPG], b_l[2][TPG];  // TPG pixel tiles per set, two sets (one in use, one being filled)
    auto read_b = [&](int buf, int tap, int grp, int set) __attribute__((always_inline)) {      // pixel tiles 2 grp, 2 grp + 1
        const half8* xh = xbase + buf * (2 * XB / 16);
        const half8* xl = xh + XB / 16;
        const int toff = (tap / 3) * LW + (tap % 3);
#pragma unroll
        for (int j = 0; j < TPG; ++j) {
            const int o = tile_off(grp * TPG + j) + toff;
            b_h[set][j] = xh[o];
            if (!X1) b_l[set][j] = xl[o];
        }
    };
    auto mfma_group = [&](int grp, int set, int slot) __attribute__((always_inline)) {
        // per accumulator: al * bh, ah * bl, ah * bh -- conv6's order
        if (!X1) {
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < TPG; ++j) acc[i][grp * TPG + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[slot][i], b_h[set][j], acc[i][grp * TPG + j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < TPG; ++j) acc[i][grp * TPG + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[slot][i], b_l[set][j], acc[i][grp * TPG + j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < TPG; ++j) acc[i][grp * TPG + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[slot][i], b_h[set][j], acc[i][grp * TPG + j], 0, 0, 0);
    };

    // Waves whose 64 output channels lie beyond Cout only carry their share of the activation DMA and keep the barrier count
    // (prologue, one per chunk boundary, epilogue): no weights, no MFMAs.
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

    // ---- prologue: first patch, weights of taps 0 and 1
#pragma unroll
    for (int q = 0; q < NACT; ++q) dma_x(ch_begin, 0, q);
    load_a(ch_begin, 0, 0);
    load_a(ch_begin, 1, 1);
    wait_vmcnt<0>();
    __syncthreads();
    read_b(0, 0, 0, 0);

    // One K chunk: 9 taps x 2 groups of (2 pixel tiles x 2 co-tiles x 3) = 12 MFMAs.  At tap t the weights of tap t + 2 are requested
    // (ring slot (t + 2) % 3; at taps 7 / 8 those are the next chunk's taps 0 / 1) and, for t < NACT, one activation piece of the next
    // chunk.  The compiler counts the register loads itself; the activation pieces are older than the weights of taps 7 and 8, so
    // "at most those 8 loads outstanding" at the chunk boundary proves that they have landed.
    auto chunk_body = [&](auto more_c, int chunk, int it) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_c)::value;
        const int cur = it & 1;
        static_for7<0, TAPS>([&](auto tap_c) __attribute__((always_inline)) {
            constexpr int tap = decltype(tap_c)::value;
            constexpr int slot = tap % 3;
            read_b(cur, tap, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (MORE && tap < NACT) dma_x(chunk + 1, cur ^ 1, tap);
            if (tap + 2 < TAPS) load_a(chunk, tap + 2, (tap + 2) % 3);
            else if (MORE) load_a(chunk + 1, tap + 2 - TAPS, (tap + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(0, 0, slot);
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < TAPS) {
                read_b(cur, tap + 1, 0, 0);
            } else if (MORE) {
                wait_vmcnt<2 * CT * NPL>();        // the weights of taps 7 and 8 (the next chunk's 0 and 1) may still be in flight
                barrier_lds_only();
                read_b(cur ^ 1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(1, 1, slot);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    {
        int it = 0, chunk = ch_begin;
        for (; chunk + 1 < ch_end; ++chunk, ++it) chunk_body(std::true_type{}, chunk, it);
        chunk_body(std::false_type{}, chunk, it);
    }

    if constexpr (EMIT) {
        // ---- fused emission (Conv6Emit).  Accumulator layout: acc[i][j][r] of lane (l31, half) = channel i*32 + 8*(r>>2) + 4*half + (r&3) of
        // the wave's 64, pixel (row 4 pw + j, column l31) of the tile.
        __syncthreads();
        float* wl = reinterpret_cast<float*>(smem7) + wave * 512;       // per-wave scratch: [0,64) bias, [64,128) S, [128,192) SS, [192,448) table
        const float osc = p.out_scale;
        wl[lane] = p.bias[co_wave + lane];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float b = wl[ch];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[i][j][r] * osc + b;
                    acc[i][j][r] = v;
                    s1 += v; s2 += v * v;
                }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }      // the 32 lanes of this half
                if (l31 == 0) { wl[64 + ch] = s1; wl[128 + ch] = s2; }
            }
        __builtin_amdgcn_wave_barrier();
        // GroupNorm needs GROUP sums only, and with integer accumulation the order of the additions is immaterial: fold the channels of a
        // group and the two pixel halves of the workgroup here, so that ONE atomic instruction per workgroup (<= 32 groups x {S, SS}) is
        // left.  (Per-channel atomics from every wave -- 512 per workgroup, 2 M per launch -- cost 0.8 ms per forward: profiles/r04.)
        const int cg = p.Cout >> 5;                           // channels per group: 4, 8 or 16
        const int gpw = 64 / cg;                              // groups per wave: 16, 8 or 4
        double* gsum = reinterpret_cast<double*>(reinterpret_cast<float*>(smem7) + 4 * 512);     // [cw 2][pw 2][16 groups][2] behind the per-wave areas
        if (lane < gpw) {
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < cg; ++k) { s1 += (double)wl[64 + lane * cg + k]; s2 += (double)wl[128 + lane * cg + k]; }
            gsum[((cw * 2 + pw) * 16 + lane) * 2] = s1;
            gsum[((cw * 2 + pw) * 16 + lane) * 2 + 1] = s2;
        }
        __syncthreads();
        const int gpb = 2 * gpw;                              // groups of this workgroup's 128 channels
        long long* const accb = p.em.acc + ((size_t)n0 * 32 + (size_t)co_blk * gpb) * 2;
        if (wave == 0) {
            const int gl = lane >> 1, t = lane & 1;           // lane = (group of the block, S | SS)
            long long r = 0;
            if (gl < gpb) {
                const int c2 = gl / gpw, g = gl - c2 * gpw;
                const double v = gsum[((c2 * 2 + 0) * 16 + g) * 2 + t] + gsum[((c2 * 2 + 1) * 16 + g) * 2 + t];
                // memory-side atomic that returns: once the result is back it has been performed (an agent-scope release fence instead
                // writes back the XCD's dirty L2 lines, i.e. everybody's plane stores: +1.5 ms per forward, profiles/r04)
                r = __hip_atomic_fetch_add(accb + gl * 2 + t, __double2ll_rn(v * (t ? 4096.0 : 1048576.0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            asm volatile("s_waitcnt vmcnt(0)" : : "v"((int)r) : "memory");
            if (lane == 0) {
                unsigned* cp = p.em.cnt + (size_t)n0 * p.n_co_blocks + co_blk;
                const unsigned old = __hip_atomic_fetch_add(cp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                asm volatile("" : : "v"(old));
                int spins = 0;
                while (__hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)(tiles_per_img + p.em.expect_extra)) {
                    switch (p.em.sleep_sel) {          // the argument of s_sleep is an immediate
                        case 0: __builtin_amdgcn_s_sleep(2); break;
                        case 1: __builtin_amdgcn_s_sleep(8); break;
                        case 3: __builtin_amdgcn_s_sleep(32); break;
                        case 4: __builtin_amdgcn_s_sleep(64); break;
                        case 6: __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); break;
                        case 7: __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); break;
                        case 2: __builtin_amdgcn_s_sleep(16); break;
                        default: __builtin_amdgcn_s_sleep(127); break;
                    }
                    if (++spins > p.em.spin_limit) { atomicAdd(p.em.range_ctr, 1ull << 40); break; }      // never hang the GPU: report through the range guard
                }
            }
        }
        __syncthreads();
        {
            const int c = co_wave + lane;
            const long long* ap = accb + (size_t)((cw * gpw) + lane / cg) * 2;
            const double S = (double)__hip_atomic_load(ap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) * (1.0 / 1048576.0);
            const double SS = (double)__hip_atomic_load(ap + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) * (1.0 / 4096.0);
            const double cntd = (double)cg * HW;
            const double mean = S / cntd;
            double var = SS / cntd - mean * mean;
            if (var < 0) var = 0;
            const float rstd = (float)(1.0 / sqrt(var + 1e-5));
            float a = rstd * p.em.gamma[c];
            float b = p.em.beta[c];
            if (p.em.film) {   // h = GN(h) * (1 + scale) + shift   (unet.py:250-251), gn_prm_kernel's arithmetic
                const float* f = p.em.film + (p.em.fstep ? (size_t)p.em.fstep->i * p.em.frows : 0) + (size_t)n0 * p.em.film_stride + p.em.film_off;
                const float sc = 1.0f + f[c];
                const float sh = f[p.Cout + c];
                a = a * sc;
                b = b * sc + sh;
            }
            reinterpret_cast<float4*>(wl + 192)[lane] = make_float4((float)mean, a, b, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        typedef _Float16 half4v __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x2e __attribute__((ext_vector_type(2)));
        typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
        const float4* tab = reinterpret_cast<const float4*>(wl + 192);
        bool bad = false;
        // A 16-byte plane entry = 8 channels of one pixel; this lane holds 4 of them (4 half .. 4 half + 3), lane ^ 32 the other 4.  For a pair
        // of pixel rows (j, j + 1) v_permlane32_swap hands the lower lanes both halves of row j and the upper lanes both halves of row
        // j + 1: one 16-byte store per lane (1 KiB per instruction) instead of two 8-byte ones.
        auto norm_split = [&](float x, const float4& m, _Float16& h, _Float16& l) __attribute__((always_inline)) {
            float v = (x - m.x) * m.y + m.z;
            v = silu7(v);
            bad |= !(fabsf(v) <= 65000.f);
            v = fminf(fmaxf(v, -65000.f), 65000.f);
            h = (_Float16)v;
            l = (_Float16)(v - (float)h);
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                float4 m[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) m[q] = tab[i * 32 + 8 * jb + 4 * half + q];
                const int c8 = (co_wave + i * 32 + 8 * jb) >> 3;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    half4v h0, l0, h1, l1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        _Float16 th, tl;
                        norm_split(acc[i][2 * jp][jb * 4 + q], m[q], th, tl); h0[q] = th; l0[q] = tl;
                        norm_split(acc[i][2 * jp + 1][jb * 4 + q], m[q], th, tl); h1[q] = th; l1[q] = tl;
                    }
                    const u32x2e a_h = __builtin_bit_cast(u32x2e, h0), b_h = __builtin_bit_cast(u32x2e, h1);
                    const u32x2e a_l = __builtin_bit_cast(u32x2e, l0), b_l = __builtin_bit_cast(u32x2e, l1);
                    u32x4e eh, el;
                    {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(a_h.x, b_h.x, false, false), s1 = __builtin_amdgcn_permlane32_swap(a_h.y, b_h.y, false, false);
                        eh.x = s0[0]; eh.y = s1[0]; eh.z = s0[1]; eh.w = s1[1];
                    }
                    const size_t eo = (((size_t)n0 * p.em.C8 + c8) * HW + (size_t)(ty0 + 4 * pw + 2 * jp + half) * p.W + (tx0 + l31)) << 4;
                    *reinterpret_cast<u32x4e*>(p.em.hi + eo) = eh;
                    if (!X1) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(a_l.x, b_l.x, false, false), s1 = __builtin_amdgcn_permlane32_swap(a_l.y, b_l.y, false, false);
                        el.x = s0[0]; el.y = s1[0]; el.z = s0[1]; el.w = s1[1];
                        *reinterpret_cast<u32x4e*>(p.em.lo + eo) = el;
                    }
                }
            }
        {
            const unsigned long long mk = __ballot(bad);
            if (mk != 0ull && lane == (int)__builtin_ctzll(mk)) atomicAdd(p.em.range_ctr, (unsigned long long)__builtin_popcountll(mk));
        }
        return;
    }

    // ---- epilogue: conv6's (straight-line buffer-descriptor code, see there), four passes q = (co-tile i, pixel-tile pair jp) of
    // 32 co x 64 px: bias, un-scaling, residual in its three forms, GroupNorm partial sums (slot = the 64-pixel group of the tile)
    __syncthreads();
    constexpr int TS = 68;
    constexpr int GPI = (TW * TH) / 64;                      // 64-pixel groups per image inside one tile
    float* tr = reinterpret_cast<float*>(smem7) + wave * (32 * TS);
    const int q4 = lane & 15, rsub = lane >> 4;
    const bool single = p.ksplit == 1;
    const float osc = p.out_scale_dev ? p.out_scale * p.out_scale_dev[0] : p.out_scale;
    const bool do_stat = p.stat != nullptr && single;
    const int res_mode = (single && p.res) ? p.res_mode : -1;
    float* const dst = single ? p.out : p.partial + (size_t)split * ((size_t)p.B * p.Cout * HW);
    const size_t img0 = (size_t)n0 * p.Cout;
    const size_t res_plane = res_mode == 1 ? (size_t)(HW >> 2) : (res_mode == 2 ? (size_t)HW * 4 : (size_t)HW);
    float* const out_base = dst + img0 * HW;
    const float* const res_base = res_mode >= 0 ? p.res + img0 * res_plane : p.bias;
    const unsigned res_bytes = res_mode >= 0 ? 0xFFFFFFFFu : 0u;
    const void* const stat_base = do_stat ? (const void*)(p.stat + img0 * p.stat_slots) : (const void*)p.bias;
    const unsigned stat_bytes = do_stat ? 0xFFFFFFFFu : 0u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    auto f4 = [](u32x4 v) { return make_float4(as_f32(v.x), as_f32(v.y), as_f32(v.z), as_f32(v.w)); };

    float bv[CT][8];
    {
        const __amdgpu_buffer_rsrc_t r_bias = rsrc_uniform(p.bias, single ? (unsigned)p.Cout * 4u : 0u);
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int it = 0; it < 8; ++it) bv[i][it] = as_f32(__builtin_amdgcn_raw_buffer_load_b32(r_bias, (unsigned)(co_wave + i * 32 + it * 4 + rsub) * 4u, 0, 0));
    }
    struct PassGeo { int ti, y, x; bool pok; unsigned pix; };
    auto geo = [&](int q) {
        const int pp = (NARROW ? pw : pw * 2 + (q & 1)) * 64 + q4 * 4;
        PassGeo g;
        g.ti = pp >> (G::LTW + G::LTH);
        g.y = ty0 + ((pp >> G::LTW) & (TH - 1));
        g.x = tx0 + (pp & (TW - 1));
        g.pok = n0 + g.ti < p.B && g.y < p.H && g.x < p.W;
        g.pix = (unsigned)(g.y * p.W + g.x);
        return g;
    };
    auto load_res = [&](int q, float4 (&rv)[8]) __attribute__((always_inline)) {
        const PassGeo g = geo(q);
        const int co0 = co_wave + (q >> 1) * 32;
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

    float4 rv[2][8];
    load_res(0, rv[0]);
    constexpr int NPASS = CT * TPG;                          // passes of 32 co x 64 px
    static_for7<0, NPASS>([&](auto q_c) __attribute__((always_inline)) {
        constexpr int q = decltype(q_c)::value;
        constexpr int i = q >> 1, jp = q & 1;
        const PassGeo g = geo(q);
        const int co0 = co_wave + i * 32;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tr[((r & 3) + 8 * (r >> 2) + 4 * half) * TS + jj * 32 + l31] = acc[i][jp * 2 + jj][r] * osc;
        if (q + 1 < NPASS) load_res(q + 1, rv[(q + 1) & 1]);   // requested before this pass's stores
        const int slot = trem * GPI + ((NARROW ? pw : pw * 2 + jp) % GPI);
        const __amdgpu_buffer_rsrc_t r_out = rsrc_uniform(out_base, 0xFFFFFFFFu);
        const __amdgpu_buffer_rsrc_t r_stat = rsrc_uniform(stat_base, stat_bytes);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int co_l = it * 4 + rsub;
            const int co = co0 + co_l;
            const bool ok = g.pok && co < p.Cout;
            float4 v = *reinterpret_cast<const float4*>(tr + co_l * TS + q4 * 4);
            {
                v.x += bv[i][it]; v.y += bv[i][it]; v.z += bv[i][it]; v.w += bv[i][it];
                const float4 r4 = rv[q & 1][it];
                v.x = r4.x + v.x; v.y = r4.y + v.y; v.z = r4.z + v.z; v.w = r4.w + v.w;
            }
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            u32x4 sv;
            sv.x = as_u32(v.x); sv.y = as_u32(v.y); sv.z = as_u32(v.z); sv.w = as_u32(v.w);
            const unsigned plane_l = (unsigned)(g.ti * p.Cout + co);
            __builtin_amdgcn_raw_buffer_store_b128(sv, r_out, ok ? (plane_l * (unsigned)HW + g.pix) * 4u : kOutOfRange, 0, 0);
            if (do_stat) {
                float s1 = (v.x + v.y) + (v.z + v.w);
                float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                s1 += dpp_row_shr7<0x111>(s1); s2 += dpp_row_shr7<0x111>(s2);
                s1 += dpp_row_shr7<0x112>(s1); s2 += dpp_row_shr7<0x112>(s2);
                s1 += dpp_row_shr7<0x114>(s1); s2 += dpp_row_shr7<0x114>(s2);
                s1 += dpp_row_shr7<0x118>(s1); s2 += dpp_row_shr7<0x118>(s2);
                u32x2 st;
                st.x = as_u32(s1); st.y = as_u32(s2);
                const bool wr = q4 == 15 && co < p.Cout && n0 + g.ti < p.B;
                __builtin_amdgcn_raw_buffer_store_b64(st, r_stat, wr ? (plane_l * (unsigned)p.stat_slots + (unsigned)slot) * 8u : kOutOfRange, 0, 0);
            }
        }
    });
#endif
}

template <int GEO, bool X1, bool NARROW = false, bool EMIT = false>
static Status launch7(hipStream_t s, const Conv6K& k, int blocks) {
    using G = Geo7<GEO>;
    constexpr int PATCH = G::TI * ((1 << G::LTH) + 2) * ((1 << G::LTW) + 2);
    constexpr int NPIECE = (2 * PATCH + 63) / 64;
    constexpr size_t LDS = (size_t)4 * NPIECE * 1024;           // two buffers x (hi, lo); the epilogue slabs (34 KiB) alias them
    static_assert(LDS >= 4 * 32 * 68 * 4, "epilogue slabs");
    auto fn = conv7_mfma_kernel<GEO, X1, NARROW, EMIT>;
    static LdsAttrOnce attr_set;
    DPIR_HIP(attr_set.set(reinterpret_cast<const void*>(fn), (int)LDS));
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), LDS, s, k);
    return Status{};
}

// Workgroups of the EMIT kernel the device holds at once: CUs x resident workgroups per CU (occupancy API, capped by the kernel's launch
// bound of two).  The fused hop's waiting set -- the workgroups of one (image, co-block) -- must fit with room to spare
// (conv7_emit_supported: at most HALF of this), instead of the constants 256 / 512 of an MI355X being assumed.
int conv7_emit_capacity() {
    static std::atomic<int> caps[16];      // per device ordinal (0 = not asked yet; stored + 1): engines on different devices share this process
    int dev = 0, cus = 0, occ = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::atomic<int>* slot = (dev >= 0 && dev < 16) ? &caps[dev] : nullptr;
    if (slot && slot->load() > 0) return slot->load() - 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    using G = Geo7<0>;
    constexpr int PATCH = G::TI * ((1 << G::LTH) + 2) * ((1 << G::LTW) + 2);
    constexpr size_t LDS = (size_t)4 * ((2 * PATCH + 63) / 64) * 1024;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv7_mfma_kernel<0, false, false, true>, 256, LDS) != hipSuccess || occ < 1) occ = 1;
    if (occ > 2) occ = 2;
    if (slot) slot->store(cus * occ + 1);
    return cus * occ;
}

// k as launch_conv6 fills it (geometry from H, W as conv6_geo); blocks = pixel tiles x co-blocks x ksplit
Status launch_conv7(hipStream_t s, const Conv6K& k, int blocks, bool x1) {
    if ((k.W & 3) || k.W < 8 || k.H < 8) return invalid("conv7: shape not tiled");
    const int geo = k.W >= 32 ? 0 : (k.W >= 16 ? 1 : 2);
    if (k.em.hi) {
        if (geo != 0 || k.ksplit != 1 || (k.Cout & 127) || (k.W & 31) || (k.H & 7) || !k.em.acc || !k.em.cnt || !k.em.range_ctr || (!x1 && !k.em.lo))
            return invalid("conv7: fused emission needs the 8 x 32 geometry, whole K, full 128-channel blocks and tiles inside the image");
        return x1 ? launch7<0, true, false, true>(s, k, blocks) : launch7<0, false, false, true>(s, k, blocks);
    }
    if (geo == 0 && k.Cout <= 32) return x1 ? launch7<0, true, true>(s, k, blocks) : launch7<0, false, true>(s, k, blocks);
    if (x1) {
        if (geo == 0) return launch7<0, true>(s, k, blocks);
        if (geo == 1) return launch7<1, true>(s, k, blocks);
        return launch7<2, true>(s, k, blocks);
    }
    if (geo == 0) return launch7<0, false>(s, k, blocks);
    if (geo == 1) return launch7<1, false>(s, k, blocks);
    return launch7<2, false>(s, k, blocks);
}

}  // namespace dpir
