// Device bodies of the wave-per-transform FFT prox at N = 256 and N = 512 (template parameter N; R = N / 64 complex values per lane) (utils/utils_sisr.py:9-19, 65-95), launched one pass per kernel by fft4.hip.  (They are
// bodies of a job index with a wait() hook because a single persistent launch of ticketed jobs was built on them too -- measured 2 x slower than the three
// launches, tools/dead_ends/prox_single_launch.)  See fft4.hip for the layout (column-major half spectrum, position order inside a column) and fft4_wave.h
// for the 256-point transform of one wave.
#pragma once
#include "common.h"
#include "elem.h"
#include "philox.h"
#include "fft4_wave.h"

namespace dpir {

struct NoWait4 { __device__ __forceinline__ void operator()() const {} };

constexpr int WAVES = 4, THREADS4 = 64 * WAVES;
// float2 per wave: the exchange tile of the transform (16 x 18 | 64 x 9); also N natural-order complex values or 2 x N staged floats
constexpr int wlds(int N) { return N == 256 ? 320 : 576; }
// row passes: a workgroup = RW waves = RW row pairs = 16 consecutive rows, so that the column-major spectrum is written / read in FULL 128-byte lines
// (16 rows x 8 bytes of one slot) through an LDS tile [slot][RW + 1] of 16-byte {A, B} entries (one wave = one entry per slot; + 1: bank spread)
constexpr int RW = 8, RTHREADS = 64 * RW, TST = RW + 1;

// Row order INSIDE a stored column: position pos(u) = 2 (u & 63) + ((u >> 6) & 1) + 128 (u >> 7), so that the four rows lane l of a column wave owns
// (u = l + 64 j) are positions {2l, 2l + 1} + 128 h, h < N / 128: a complex column is N / 128 fully contiguous 1 KB wave loads of 16 bytes per lane instead of N / 64
// 512-byte ones (half the requests; measured the same time per apply, profiles/r06 -- the pass is not bound by access width).  The row passes pair row r with row r + 64 (any two real rows
// can share a complex transform): their {A, B} entries are then adjacent positions, and the eight pairs of a workgroup fill one 128-byte line per slot.
__host__ __device__ __forceinline__ int pos4(int u) { return 2 * (u & 63) + ((u >> 6) & 1) + 128 * (u >> 7); }
// pair q (0..N/2-1) of a plane: rows rA = (q & 63) + 128 (q >> 6) and rA + 64
__device__ __forceinline__ int pair_row(int q) { return (q & 63) + 128 * (q >> 6); }

__device__ __forceinline__ void wsync() { wave_sync(); }

// ------------------------------------------------------------------------------------------------ rows forward
// grid: pairs / 8 workgroups x 512 threads; wave = one row pair.  NC = stored columns (slots) per plane; slot_col (sf > 1): slot -> column | mirror << 16, -1 padding.
// job = 8 row pairs (one per wave) = 16 rows of one plane: job index `wg` = plane * 16 + m.  TWREG: the per-lane twiddles are already in `w`.
template <int N, bool TWREG>
__device__ __forceinline__ void rows4_body(float2* sm4, size_t wg, const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out, int NC,
                                           const float2* tw, WaveTwN<N> w, RowsFuse fu, const int* slot_col) {
    constexpr int R = N / 64, Q = N / 256, WL = wlds(N);               // Q float4 per lane and row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t pair = wg * RW + wave;
    float2* lds = sm4 + wave * WL;
    float4* tile = reinterpret_cast<float4*>(sm4 + RW * WL);           // [NC][TST]
    if (sp) pm = sp->tau;
    const size_t plane = pair / (N / 2);
    const int r = pair_row((int)(pair - plane * (N / 2)));             // rows r and r + 64 of the plane
    const size_t ra = plane * N + r;
    // every global request of the wave first (rows, eps, then the per-lane twiddles): ONE memory round trip before the transform
    float4 qa[Q], qb[Q], ea[Q], eb[Q];
#pragma unroll
    for (int h = 0; h < Q; ++h) {
        qa[h] = *reinterpret_cast<const float4*>(x + ra * N + 256 * h + lane * 4);
        qb[h] = *reinterpret_cast<const float4*>(x + (ra + 64) * N + 256 * h + lane * 4);
        ea[h] = make_float4(0.f, 0.f, 0.f, 0.f); eb[h] = ea[h];
    }
    if (fu.eps6) {
        const size_t n = plane / 3, c = plane - n * 3;
        const float* ep = fu.eps6 + ((n * fu.out_ch + c) * N + r) * N + lane * 4;
#pragma unroll
        for (int h = 0; h < Q; ++h) { ea[h] = *reinterpret_cast<const float4*>(ep + 256 * h); eb[h] = *reinterpret_cast<const float4*>(ep + 64 * N + 256 * h); }
    }
    if (!TWREG) w = wave_tw_load<N>(tw, lane);
    if (fu.eps6) {
#pragma clang fp contract(off)
        const float c1 = sp->c1, c2 = sp->c2;
#pragma unroll
        for (int h = 0; h < Q; ++h) {
            qa[h].x = fminf(fmaxf(c1 * qa[h].x - c2 * ea[h].x, -1.0f), 1.0f); qa[h].y = fminf(fmaxf(c1 * qa[h].y - c2 * ea[h].y, -1.0f), 1.0f);
            qa[h].z = fminf(fmaxf(c1 * qa[h].z - c2 * ea[h].z, -1.0f), 1.0f); qa[h].w = fminf(fmaxf(c1 * qa[h].w - c2 * ea[h].w, -1.0f), 1.0f);
            qb[h].x = fminf(fmaxf(c1 * qb[h].x - c2 * eb[h].x, -1.0f), 1.0f); qb[h].y = fminf(fmaxf(c1 * qb[h].y - c2 * eb[h].y, -1.0f), 1.0f);
            qb[h].z = fminf(fmaxf(c1 * qb[h].z - c2 * eb[h].z, -1.0f), 1.0f); qb[h].w = fminf(fmaxf(c1 * qb[h].w - c2 * eb[h].w, -1.0f), 1.0f);
        }
    }
    float* st = reinterpret_cast<float*>(lds);                        // [2][N] floats
#pragma unroll
    for (int h = 0; h < Q; ++h) {
        *reinterpret_cast<float4*>(st + 256 * h + lane * 4) = qa[h];
        *reinterpret_cast<float4*>(st + N + 256 * h + lane * 4) = qb[h];
    }
    wsync();
    float2 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const float a = (st[lane + 64 * j] * pa + pb) * pm, b = (st[N + lane + 64 * j] * pa + pb) * pm;
        v[j] = make_float2(a, b);
    }
    wsync();
    wave_fft<false>(v, w, lds, lane);
    wsync();
#pragma unroll
    for (int j = 0; j < R; ++j) lds[lane + 64 * j] = v[j];
    wsync();
    // un-pack: A[k] = (Z[k] + conj(Z[N-k]))/2, B[k] = (Z[k] - conj(Z[N-k]))/(2i), k = 0..N/2, into this wave's entry of every slot's tile row
    if (!slot_col) {
#pragma unroll
        for (int j = 0; j < R / 2 + 1; ++j) {
            const int k = lane + 64 * j;
            if (k > N / 2) break;
            float2 zk = v[j], zn = lds[(N - k) & (N - 1)];
            zn.y = -zn.y;
            const float2 d = csub(zk, zn);
            tile[k * TST + wave] = make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y), 0.5f * d.y, -0.5f * d.x);
        }
    } else {
        for (int s = lane; s < NC; s += 64) {
            const int cm = slot_col[s];
            float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cm >= 0) {
                const int k = cm & 0xffff;
                float2 zk = lds[k], zn = lds[(N - k) & (N - 1)];
                zn.y = -zn.y;
                const float2 d = csub(zk, zn);
                ab = make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y), 0.5f * d.y, -0.5f * d.x);
            }
            tile[s * TST + wave] = ab;
        }
    }
    __syncthreads();
    // the workgroup's 16 rows of every slot are 16 consecutive positions: one full 128-byte line per slot, eight lanes per line
    const size_t plane0 = wg / (N / 2 / RW);
    const int m = (int)(wg - plane0 * (N / 2 / RW));
    float4* o4 = reinterpret_cast<float4*>(out + (plane0 * NC) * N + 16 * m);
    for (int i = threadIdx.x; i < NC * RW; i += RTHREADS) {
        const int s = i >> 3, c = i & 7;
        o4[(size_t)s * (N / 2) + c] = tile[s * TST + c];
    }
}

// ------------------------------------------------------------------------------------------------ rows inverse
// job index `wg` = plane * 16 + m; wait(): called by all threads after the requests that do not depend on the column pass, before those that do
template <int N, bool TWREG, class Wait>
__device__ __forceinline__ void irows4_body(float2* sm4, size_t wg, const float2* in, float* out, float scale, float oa, float ob, const float* blend_base, float g,
                                            int NC, const float2* tw, WaveTwN<N> w, RenoiseFuse rn, const int* col_slot, Wait wait) {
    constexpr int R = N / 64, Q = N / 256, WL = wlds(N);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t pair = wg * RW + wave;
    float2* lds = sm4 + wave * WL;
    float4* tile = reinterpret_cast<float4*>(sm4 + RW * WL);           // [N/2 + 1 columns][TST]
    const size_t plane = pair / (N / 2);
    const int r = pair_row((int)(pair - plane * (N / 2)));             // rows r and r + 64 of the plane
    const size_t ra = plane * N + r;
    // what the epilogue combines with the transform (x_t for the fused re-noise, or the blend base) does not depend on it: requested first
    const float* pre_src = rn.xt ? rn.xt : blend_base;
    const size_t ga = ra * N + lane * 4, gb = ga + 64 * N;
    float4 pre_a[Q], pre_b[Q];
#pragma unroll
    for (int h = 0; h < Q; ++h) {
        pre_a[h] = make_float4(0.f, 0.f, 0.f, 0.f); pre_b[h] = pre_a[h];
        if (pre_src) { pre_a[h] = *reinterpret_cast<const float4*>(pre_src + ga + 256 * h); pre_b[h] = *reinterpret_cast<const float4*>(pre_src + gb + 256 * h); }
    }
    wait();
    {   // the workgroup's 16 rows of columns 0..N/2: one full 128-byte line per column, eight lanes per line; all requests before the first LDS store
        const int m = (int)(wg - plane * (N / 2 / RW));
        const float4* i4 = reinterpret_cast<const float4*>(in + (plane * NC) * N + 16 * m);
        constexpr int NT = ((N / 2 + 1) * RW + RTHREADS - 1) / RTHREADS;
        float4 t[NT];
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int i = min(threadIdx.x + it * RTHREADS, (N / 2 + 1) * RW - 1);        // clamped, not branched: the surplus lanes re-read the last entry
            const int k = i >> 3, c = i & 7;
            const int s = col_slot ? col_slot[k] : k;
            t[it] = i4[(size_t)s * (N / 2) + c];
        }
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int i = min(threadIdx.x + it * RTHREADS, (N / 2 + 1) * RW - 1);
            tile[(i >> 3) * TST + (i & 7)] = t[it];
        }
    }
    if (!TWREG) w = wave_tw_load<N>(tw, lane);
    __syncthreads();
    float4 ab[R / 2 + 1];
#pragma unroll
    for (int j = 0; j < R / 2 + 1; ++j) {
        const int k = lane + 64 * j;
        ab[j] = k <= N / 2 ? tile[k * TST + wave] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Hermitian re-packing: Z[k] = A[k] + i B[k], Z[N-k] = conj(A[k]) + i conj(B[k])
#pragma unroll
    for (int j = 0; j < R / 2 + 1; ++j) {
        const int k = lane + 64 * j;
        if (k > N / 2) break;
        const float4 q = ab[j];
        lds[k] = make_float2(q.x - q.w, q.y + q.z);
        if (k > 0 && k < N / 2) lds[N - k] = make_float2(q.x + q.w, -q.y + q.z);
    }
    wsync();
    float2 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = lds[lane + 64 * j];
    wsync();
    wave_fft<true>(v, w, lds, lane);
    wsync();
    float* st = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        st[lane + 64 * j] = (v[j].x * scale) * oa + ob;
        st[N + lane + 64 * j] = (v[j].y * scale) * oa + ob;
    }
    wsync();
#pragma unroll
    for (int hh = 0; hh < 2 * Q; ++hh) {
        const int h = hh / Q, hq = hh % Q;                              // row (A | B) and float4 group inside the row
        float4 q = *reinterpret_cast<const float4*>(st + h * N + 256 * hq + lane * 4);
        const size_t gi = (h ? gb : ga) + 256 * hq;
        const float4 pre = h ? pre_b[hq] : pre_a[hq];
        if (blend_base) {
            const float4 b0 = rn.xt ? *reinterpret_cast<const float4*>(blend_base + gi) : pre;
            q.x = b0.x + g * (q.x - b0.x); q.y = b0.y + g * (q.y - b0.y); q.z = b0.z + g * (q.z - b0.z); q.w = b0.w + g * (q.w - b0.w);
        }
        if (rn.xt) {
#pragma clang fp contract(off)
            const StepDev sd = *rn.sp;
            const size_t per_image = (size_t)3 * N * N;
            const size_t n = gi / per_image, e = gi - n * per_image;
            float z1[4] = {0.f, 0.f, 0.f, 0.f}, z2[4];
            if (rn.n2) {                         // host-fed noise: this batch's tensors, step i
                const float4 t2 = *reinterpret_cast<const float4*>((rn.lp ? rn.lp->n2 : rn.n2) + (size_t)sd.i * rn.stride + gi);
                z2[0] = t2.x; z2[1] = t2.y; z2[2] = t2.z; z2[3] = t2.w;
                if (rn.with_n1) {
                    const float4 t1 = *reinterpret_cast<const float4*>((rn.lp ? rn.lp->n1 : rn.n1) + (size_t)sd.i * rn.stride + gi);
                    z1[0] = t1.x; z1[1] = t1.y; z1[2] = t1.z; z1[3] = t1.w;
                }
            } else {
                const uint64_t img = (uint64_t)(rn.lp->image_offset + (long long)n);
                philox_normal4(rn.lp->seed, 2 + 4 * (uint64_t)sd.i, img, e >> 2, z2);
                if (rn.with_n1) philox_normal4(rn.lp->seed, 1 + 4 * (uint64_t)sd.i, img, e >> 2, z1);
            }
            const float xv[4] = {pre.x, pre.y, pre.z, pre.w}, av[4] = {q.x, q.y, q.z, q.w};
            float rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = av[u];
                const float eps = (xv[u] - sd.sa_t * a) / sd.s1m_t;
                float inner = sd.q * eps;
                if (rn.with_n1) inner = inner + sd.es * z1[u];
                float vv = sd.sa_p * a + sd.k1 * inner;
                vv = vv + sd.k2 * z2[u];
                rv[u] = vv;
            }
            *reinterpret_cast<float4*>(rn.xt + gi) = make_float4(rv[0], rv[1], rv[2], rv[3]);
            continue;
        }
        *reinterpret_cast<float4*>(out + gi) = q;
    }
}

// ------------------------------------------------------------------------------------------------ columns
// wave = one stored column (slot) of one plane.  MODE 0: forward only; MODE 2 (sf = 1): forward -> FX = (FR - conj(FB) (FB FR)/(F2B + alpha)) / alpha,
// FR = FBFy + F(alpha x) -> inverse; MODE 3 (sf > 1): the same with FB FR and F2B averaged over the sf x sf aliases (utils_sisr.py:65-75 `splits` + mean).
// An ITEM = four consecutive slots of one plane = four waves (`wave` = 0..3 inside the item, `fold`: the item's [4][N / sf] LDS area for MODE 3);
// ceil(NC / 4) items per plane.  MODE 3 has two workgroup barriers: every wave of the workgroup must run an item (dead ones on slot 0, storing nothing).
// The solve's three spectra do not depend on the row pass: they are requested BEFORE wait(), the column after it.
// (A variant in which a workgroup WALKED several items with register prefetch of the next one measured slower, 29 vs 26 us per batch-16 apply; and
// 16-byte lane loads -- the position order -- are not faster than 8-byte ones here: the pass is bound by its load -> transform -> store chain with
// every wave of the launch in the same phase, profiles/r06.)
template <int MODE, int SF, int N, class Wait>
__device__ __forceinline__ void cols4_item_body(float2* lds, float2* fold, int plane, int item, int wave, float2* buf, const SolveArgs& a, int NC,
                                                const WaveTwN<N>& w, Wait wait) {
    constexpr int R = N / 64;
    const int lane = threadIdx.x & 63;
    const int s = item * WAVES + wave;
    const bool live = s < NC;                                          // dead waves re-read slot 0 (loads stay unconditional: a `live ? load : 0`
    const size_t col = ((size_t)plane * NC + (live ? s : 0)) * N;      // select makes hipcc branch around every dword) and store nothing
    const size_t icol = ((size_t)(plane / 3) * NC + (live ? s : 0)) * N;
    // lane l owns positions 2l, 2l + 1 (+ 128 h: registers 2h, 2h + 1): N / 128 contiguous 1 KB wave loads per complex column
    float2 v[R], fy[R], fb[R]; float f2[R];
    if (MODE >= 2) {
#pragma unroll
        for (int h = 0; h < R / 2; ++h) {
            const float4 y = reinterpret_cast<const float4*>(a.FBFy + col)[64 * h + lane];
            const float4 f = reinterpret_cast<const float4*>(a.FB + icol)[64 * h + lane];
            fy[2 * h] = make_float2(y.x, y.y); fy[2 * h + 1] = make_float2(y.z, y.w);
            fb[2 * h] = make_float2(f.x, f.y); fb[2 * h + 1] = make_float2(f.z, f.w);
            if (MODE == 2) {
                const float2 t = reinterpret_cast<const float2*>(a.F2B + icol)[64 * h + lane];
                f2[2 * h] = t.x; f2[2 * h + 1] = t.y;
            }
        }
    }
    wait();
#pragma unroll
    for (int h = 0; h < R / 2; ++h) {
        const float4 q = reinterpret_cast<const float4*>(buf + col)[64 * h + lane];
        v[2 * h] = make_float2(q.x, q.y); v[2 * h + 1] = make_float2(q.z, q.w);
    }
    wave_fft<false>(v, w, lds, lane);
    if (MODE == 2) {
        const float alpha = a.sp ? a.sp->tau : a.alpha;
        // the two quotients of the cancelling term keep the exact division of the reference's expression; the final (uniform) division by
        // alpha acts on the difference AFTER the cancellation, where one more rounding is not amplified: a multiplication by 1 / alpha
        const float inv_alpha = 1.0f / alpha;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float2 fr = cadd(fy[j], v[j]);
            const float2 x1 = cmul2(fb[j], fr);
            const float den = f2[j] + alpha;
            const float2 q = make_float2(x1.x / den, x1.y / den);
            const float2 tq = cmulc2(q, fb[j]);                          // conj(FB) * q
            v[j] = make_float2((fr.x - tq.x) * inv_alpha, (fr.y - tq.y) * inv_alpha);
        }
        wsync();
        wave_fft<true>(v, w, lds, lane);
    }
    if (MODE == 3) {
        // slot s = sf q + b: alias b of fold group q.  Row aliases u + a Hs (Hs = N / sf, a multiple of 64) are registers of ONE lane.
        const float alpha = a.sp ? a.sp->tau : a.alpha;
        const float inv_alpha = 1.0f / alpha;
        constexpr int Hs = N / SF, KH = R / SF;                         // KH folded values per lane
        const int QW = N / SF / 2 + 1;
        const int n_img = plane / 3;
        float2 sacc[KH];
#pragma unroll
        for (int i = 0; i < KH; ++i) sacc[i] = make_float2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            v[j] = cadd(fy[j], v[j]);
            sacc[j % KH] = cadd(sacc[j % KH], cmul2(fb[j], v[j]));
        }
#pragma unroll
        for (int i = 0; i < KH; ++i) fold[wave * Hs + lane + 64 * i] = sacc[i];
        __syncthreads();
        // R[p] of this slot's fold group: sum over the group's sf slots (mirrored aliases: conj of the mirrored row), / (sf^2 (invW + alpha))
        const int cmine = live ? a.slot_col[s] : -1;
        const bool mir = cmine >= 0 && (cmine >> 16);
        const int q = s / SF, w0 = (wave / SF) * SF;                    // first wave (slot) of my fold group inside the item
        const float inv_n = 1.0f / (float)(SF * SF);
        float2 Rr[KH];
#pragma unroll
        for (int i = 0; i < KH; ++i) {
            const int p0 = lane + 64 * i;                               // the row (mod Hs) this lane needs R for ...
            const int p = mir ? (Hs - p0) % Hs : p0;                    // ... which for a mirrored slot is R[(Hs - p) % Hs] conjugated
            const int pm = (Hs - p) % Hs;
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int b = 0; b < SF; ++b) {
                const int sb = q * SF + b;
                const int cm = sb < NC ? a.slot_col[sb] : -1;
                if (cm < 0) continue;
                if (cm >> 16) { const float2 z = fold[(w0 + b) * Hs + pm]; acc.x += z.x; acc.y -= z.y; }
                else acc = cadd(acc, fold[(w0 + b) * Hs + p]);
            }
            float2 rr = make_float2(0.f, 0.f);
            if (live && q < QW) {
                const float den = a.invW[((size_t)n_img * Hs + p) * QW + q] + alpha;
                rr = make_float2(acc.x * inv_n / den, acc.y * inv_n / den);
            }
            if (mir) rr.y = -rr.y;
            Rr[i] = rr;
        }
        __syncthreads();                                                // `fold` is free again
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float2 tq = cmulc2(Rr[j % KH], fb[j]);                // conj(FB) * R~
            v[j] = make_float2((v[j].x - tq.x) * inv_alpha, (v[j].y - tq.y) * inv_alpha);
        }
        wsync();
        wave_fft<true>(v, w, lds, lane);
    }
    if (live) {
        float4* base = reinterpret_cast<float4*>(buf + col);
#pragma unroll
        for (int h = 0; h < R / 2; ++h) base[64 * h + lane] = make_float4(v[2 * h].x, v[2 * h].y, v[2 * h + 1].x, v[2 * h + 1].y);
    }
    wsync();
}

}  // namespace dpir
