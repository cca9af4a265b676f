// Device bodies of the half-spectrum register-FFT prox (utils/utils_sisr.py:9-19, 65-95), used by fft2.hip (one kernel per pass); written as bodies of a job index so that
// a persistent single launch can run them as ticketed jobs (tried and measured slower: tools/dead_ends/prox_single_launch).  A body is written for
// a block of THREADS threads and a job index `bid` (what blockIdx.x is in the three-launch kernels); its arithmetic does not depend on
// THREADS, so both paths produce the same bits.
//   PERSIST = false: the body stages the twiddle table itself (requested first, stored once its own loads are in flight);
//   PERSIST = true : the table is already in LDS (loaded once per workgroup), and `wait()` is called right before the first load that
//                    depends on another job's output -- everything that does NOT depend on it is requested before the wait.
#pragma once
#include "common.h"
#include "elem.h"
#include "philox.h"
#include "fft_regs.h"

namespace dpir {

// Two-pass N = RT * RJ transform for one "slot": RT cooperating threads (t = 0..RT-1) with RJ values each (RJ a multiple of RT;
// 16 x 16 at N = 256, 8 x 8 at N = 64, 16 x 32 at N = 512).
//   pass 1 in : thread t holds x[RT j + t], j = 0..RJ-1            (stride-RT elements, offset t): one RJ-point register FFT
//   pass 2    : RJ / RT register FFTs of RT points per thread (k1 = t + RT s)
//   out       : thread t holds X[t + RT j], j = 0..RJ-1            (same distribution -> the inverse can start from it)
// xch: this slot's LDS exchange area of RJ*(RT+1) float2; twN: table of W_N^m (cos, -sin), m < N, in LDS.
template <int RT, int RJ, bool INV>
__device__ __forceinline__ void fft_two_pass(float2 (&v)[RJ], int t, float2* xch, const float2* twN) {
    constexpr int NS = RJ / RT;
    static_assert(RJ % RT == 0, "RJ must be a multiple of RT");
    RegFFT<RJ, INV>::run(v);                                  // over j -> Y[k1] for n2 = t
#pragma unroll
    for (int k1 = 0; k1 < RJ; ++k1) {
        float2 tw = twN[(t * k1) & (RT * RJ - 1)];
        float2 y = INV ? cmulc2(v[k1], tw) : cmul2(v[k1], tw);
        xch[k1 * (RT + 1) + t] = y;
    }
    __syncthreads();
    float2 u[NS][RT];
#pragma unroll
    for (int sft = 0; sft < NS; ++sft)
#pragma unroll
        for (int n2 = 0; n2 < RT; ++n2) u[sft][n2] = xch[(t + RT * sft) * (RT + 1) + n2];     // thread reads Y[k1 = t + RT s][n2]
    __syncthreads();
#pragma unroll
    for (int sft = 0; sft < NS; ++sft) {
        RegFFT<RT, INV>::run(u[sft]);                          // over n2 -> X[k1 + RJ k2]
#pragma unroll
        for (int k2 = 0; k2 < RT; ++k2) v[k2 * NS + sft] = u[sft][k2];                          // index t + RT (NS k2 + s)
    }
}

struct NoWait { __device__ __forceinline__ void operator()() const {} };

// ------------------------------------------------------------------------------------------------ rows forward
// One slot = one PAIR of real rows; a job = SLOTS = THREADS / R consecutive pairs starting at pair bid * SLOTS.
// Fused loop prologue (dpir_run_loop): when `eps6` is given, the row loaded is not x but the denoiser's clamped x0 prediction
// x0 = clamp(c1*x - c2*eps, -1, 1) (gaussian_diffusion.py:297,328-333), evaluated while staging -- x0 is never materialised.
template <int R, int RJ, int THREADS, bool PERSIST>
__device__ __forceinline__ void rfft_rows_body(float2* sm2, size_t bid, const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out,
                                               int WP, size_t total_rows, const float2* tw, RowsFuse fu, const int* slot_col) {
    constexpr int N = R * RJ, SLOTS = THREADS / R;
    float2* twN = sm2;                                  // [N]
    float2* xch = sm2 + N;                              // [SLOTS][RJ*(R+1)]
    // [SLOTS][N+4] natural-order Z of each slot (also the load staging).  ALIASED with the exchange area: staging is dead once the
    // two-pass layout has been gathered into registers, the exchange area is dead when fft_two_pass returns (each hand-over is a
    // __syncthreads) -- 19 -> 10.7 KiB per 64-thread workgroup at N = 256
    float2* zbuf = xch;
    if (sp) pm = sp->tau;
    // the twiddle table is requested FIRST but stored to LDS only after the row loads below are in flight too: a load -> ds_write pair
    // in front of them would be a whole memory round trip before the first row request leaves the CU
    constexpr int NTW = PERSIST ? 1 : (N + THREADS - 1) / THREADS;
    float2 twr[NTW];
    if (!PERSIST) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; twr[j] = i < N ? tw[i] : make_float2(0.f, 0.f); }
    }
    const int slot = threadIdx.x / R, t = threadIdx.x % R;
    const size_t pair = bid * SLOTS + slot;
    const size_t ra = 2 * pair, rb = 2 * pair + 1;
    const bool va = ra < total_rows, vb = rb < total_rows;
    // coalesced float4 loads of the block's 2*SLOTS rows into LDS, then the strided gather of the two-pass layout
    float* stage = reinterpret_cast<float*>(zbuf);          // [2*SLOTS][N + 4] floats == SLOTS*(N+4) float2
    {
        const size_t row0 = bid * SLOTS * 2;
        constexpr int V4 = N / 4;
        // ALL loads of the workgroup's rows are issued before the first one is consumed (a load -> LDS-store loop body is a chain of
        // dependent memory round trips: 8 per thread at N = 256)
        constexpr int NL = (2 * SLOTS * V4 + THREADS - 1) / THREADS;
        float4 q[NL], e4[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = threadIdx.x + j * THREADS;
            const int r = i / V4, c4 = i - r * V4;
            const size_t row = row0 + r;
            q[j] = make_float4(0.f, 0.f, 0.f, 0.f); e4[j] = q[j];
            if (i < 2 * SLOTS * V4 && row < total_rows) {
                q[j] = *reinterpret_cast<const float4*>(x + row * N + c4 * 4);
                if (fu.eps6) {
                    const size_t plane = row / N, n = plane / 3, c = plane - n * 3;
                    e4[j] = *reinterpret_cast<const float4*>(fu.eps6 + ((n * fu.out_ch + c) * N + (row - plane * N)) * N + c4 * 4);
                }
            }
        }
        if (!PERSIST) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; if (i < N) twN[i] = twr[j]; }
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = threadIdx.x + j * THREADS;
            const int r = i / V4, c4 = i - r * V4;
            if (i >= 2 * SLOTS * V4) continue;
            float4 qq = q[j];
            if (fu.eps6 && row0 + r < total_rows) {
#pragma clang fp contract(off)
                const float c1 = sp->c1, c2 = sp->c2;
                qq.x = fminf(fmaxf(c1 * qq.x - c2 * e4[j].x, -1.0f), 1.0f); qq.y = fminf(fmaxf(c1 * qq.y - c2 * e4[j].y, -1.0f), 1.0f);
                qq.z = fminf(fmaxf(c1 * qq.z - c2 * e4[j].z, -1.0f), 1.0f); qq.w = fminf(fmaxf(c1 * qq.w - c2 * e4[j].w, -1.0f), 1.0f);
            }
            *reinterpret_cast<float4*>(stage + r * (N + 4) + c4 * 4) = qq;
        }
    }
    __syncthreads();
    float2 v[RJ];
#pragma unroll
    for (int n1 = 0; n1 < RJ; ++n1) {
        int n = R * n1 + t;
        float a = (stage[(2 * slot) * (N + 4) + n] * pa + pb) * pm;
        float b = (stage[(2 * slot + 1) * (N + 4) + n] * pa + pb) * pm;
        v[n1] = make_float2(va ? a : 0.f, vb ? b : 0.f);
    }
    __syncthreads();
    fft_two_pass<R, RJ, false>(v, t, xch + slot * RJ * (R + 1), twN);
    float2* z = zbuf + slot * (N + 4);
#pragma unroll
    for (int k2 = 0; k2 < RJ; ++k2) z[t + R * k2] = v[k2];
    __syncthreads();
    // un-pack: A[k] = (Z[k] + conj(Z[N-k]))/2, B[k] = (Z[k] - conj(Z[N-k]))/(2i), k = 0..N/2; zero the padding columns
    for (int ks = t; ks < WP; ks += R) {
        float2 A = make_float2(0.f, 0.f), Bv = make_float2(0.f, 0.f);
        // stored slot ks holds spectrum column k: identity for sf = 1, the alias-grouped permutation for sf > 1 (-1: padding)
        int k = ks <= N / 2 ? ks : -1;
        if (slot_col) { const int cm = slot_col[ks]; k = cm < 0 ? -1 : (cm & 0xffff); }
        if (k >= 0) {
            float2 zk = z[k], zn = z[(N - k) & (N - 1)];
            zn.y = -zn.y;
            A = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y));
            float2 d = csub(zk, zn);
            Bv = make_float2(0.5f * d.y, -0.5f * d.x);
        }
        if (va) out[ra * WP + ks] = A;
        if (vb) out[rb * WP + ks] = Bv;
    }
}

// ------------------------------------------------------------------------------------------------ rows inverse
// Fused loop epilogue (dpir_run_loop): when `xt` is given, the value produced is x0' (the prox output in [-1,1]) and what is
// STORED is the re-noised iterate (main_ddpir.py:451-456)
//     eps = (x_t - sa_t x0') / s1m_t;   x = sa_p x0' + k1 (q eps + es n1) + k2 n2
// written over x_t; n1 / n2 are host-fed tensors or Philox draws (same (seed, image, stream, counter) as randn_kernel).
template <int R, int RJ, int THREADS, bool PERSIST, class Wait>
__device__ __forceinline__ void irfft_rows_body(float2* sm2, size_t bid, const float2* in, float* out, float scale, float oa, float ob,
                                                const float* blend_base, float g, int WP, size_t total_rows, const float2* tw, RenoiseFuse rn,
                                                const int* col_slot, Wait wait) {
    constexpr int N = R * RJ, SLOTS = THREADS / R;
    float2* twN = sm2;
    float2* xch = sm2 + N;
    float2* zbuf = xch;                                 // aliased with the exchange area (see rfft_rows_body)
    constexpr int NTW = PERSIST ? 1 : (N + THREADS - 1) / THREADS;
    float2 twr[NTW];                                    // requested first, stored after the spectrum loads are in flight (see rfft_rows_body)
    if (!PERSIST) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; twr[j] = i < N ? tw[i] : make_float2(0.f, 0.f); }
    }
    const int slot = threadIdx.x / R, t = threadIdx.x % R;
    const size_t pair = bid * SLOTS + slot;
    const size_t ra = 2 * pair, rb = 2 * pair + 1;
    const bool va = ra < total_rows, vb = rb < total_rows;
    float2* z = zbuf + slot * (N + 4);
    // what the epilogue combines with the transform's result (x_t for the fused re-noise, or the guidance-blend base) does not depend on
    // it (nor, in the persistent launch, on the column jobs): requested first, its latency hides under the wait and the transform
    const size_t row0 = bid * SLOTS * 2;
    constexpr int V4 = N / 4;
    constexpr int NS4 = (2 * SLOTS * V4 + THREADS - 1) / THREADS;
    const float* pre_src = rn.xt ? rn.xt : blend_base;
    float4 pre[NS4];
    if (PERSIST) {
#pragma unroll
        for (int j = 0; j < NS4; ++j) {
            const int i = threadIdx.x + j * THREADS;
            const int r = i / V4, c4 = i - r * V4;
            pre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pre_src && i < 2 * SLOTS * V4 && row0 + r < total_rows) pre[j] = *reinterpret_cast<const float4*>(pre_src + (row0 + r) * N + c4 * 4);
        }
        wait();
    }
    // Hermitian re-packing: Z[k] = A[k] + i B[k], Z[N-k] = conj(A[k]) + i conj(B[k])
    constexpr int NK = (N / 2 + 1 + R - 1) / R;               // all loads in flight before the first LDS write (see rfft_rows_body)
    float2 Av[NK], Bw[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const int k = t + j * R;
        Av[j] = make_float2(0.f, 0.f); Bw[j] = Av[j];
        if (k <= N / 2) {
            const int ks = col_slot ? col_slot[k] : k;      // where column k is stored (sf > 1: permuted)
            if (va) Av[j] = in[ra * WP + ks];
            if (vb) Bw[j] = in[rb * WP + ks];
        }
    }
    if (!PERSIST) {
#pragma unroll
        for (int j = 0; j < NS4; ++j) {
            const int i = threadIdx.x + j * THREADS;
            const int r = i / V4, c4 = i - r * V4;
            pre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pre_src && i < 2 * SLOTS * V4 && row0 + r < total_rows) pre[j] = *reinterpret_cast<const float4*>(pre_src + (row0 + r) * N + c4 * 4);
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; if (i < N) twN[i] = twr[j]; }
    }
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const int k = t + j * R;
        if (k > N / 2) continue;
        const float2 A = Av[j], Bv = Bw[j];
        z[k] = make_float2(A.x - Bv.y, A.y + Bv.x);
        if (k > 0 && k < N / 2) z[N - k] = make_float2(A.x + Bv.y, -A.y + Bv.x);
    }
    __syncthreads();
    float2 v[RJ];
#pragma unroll
    for (int n1 = 0; n1 < RJ; ++n1) v[n1] = z[R * n1 + t];
    __syncthreads();
    fft_two_pass<R, RJ, true>(v, t, xch + slot * RJ * (R + 1), twN);
    // stage the block's 2*SLOTS real rows in LDS, then float4 row-contiguous stores (a lane-strided direct store writes
    // 64-byte fragments of 8 different rows per instruction)
    float* stage = reinterpret_cast<float*>(zbuf);          // [2*SLOTS][N + 4] floats (the z area is dead now)
#pragma unroll
    for (int k2 = 0; k2 < RJ; ++k2) {
        int n = t + R * k2;
        stage[(2 * slot) * (N + 4) + n] = (v[k2].x * scale) * oa + ob;
        stage[(2 * slot + 1) * (N + 4) + n] = (v[k2].y * scale) * oa + ob;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NS4; ++j) {
        const int i = threadIdx.x + j * THREADS;
        if (i >= 2 * SLOTS * V4) continue;
        int r = i / V4, c4 = i - r * V4;
        size_t row = row0 + r;
        if (row >= total_rows) continue;
        float4 q = *reinterpret_cast<const float4*>(stage + r * (N + 4) + c4 * 4);
        size_t gi = row * N + c4 * 4;
        if (blend_base) {
            float4 b0 = rn.xt ? *reinterpret_cast<const float4*>(blend_base + gi) : pre[j];
            q.x = b0.x + g * (q.x - b0.x); q.y = b0.y + g * (q.y - b0.y); q.z = b0.z + g * (q.z - b0.z); q.w = b0.w + g * (q.w - b0.w);
        }
        if (rn.xt) {
#pragma clang fp contract(off)
            const StepDev st = *rn.sp;
            const size_t per_image = (size_t)3 * N * N;
            const size_t n = gi / per_image, e = gi - n * per_image;
            float z1[4] = {0.f, 0.f, 0.f, 0.f}, z2[4];
            if (rn.n2) {                         // host-fed noise: this batch's tensors, step i
                const float4 t2 = *reinterpret_cast<const float4*>((rn.lp ? rn.lp->n2 : rn.n2) + (size_t)st.i * rn.stride + gi);
                z2[0] = t2.x; z2[1] = t2.y; z2[2] = t2.z; z2[3] = t2.w;
                if (rn.with_n1) {
                    const float4 t1 = *reinterpret_cast<const float4*>((rn.lp ? rn.lp->n1 : rn.n1) + (size_t)st.i * rn.stride + gi);
                    z1[0] = t1.x; z1[1] = t1.y; z1[2] = t1.z; z1[3] = t1.w;
                }
            } else {
                const uint64_t img = (uint64_t)(rn.lp->image_offset + (long long)n);
                philox_normal4(rn.lp->seed, 2 + 4 * (uint64_t)st.i, img, e >> 2, z2);
                if (rn.with_n1) philox_normal4(rn.lp->seed, 1 + 4 * (uint64_t)st.i, img, e >> 2, z1);
            }
            const float4 xo = pre[j];
            const float xv[4] = {xo.x, xo.y, xo.z, xo.w}, av[4] = {q.x, q.y, q.z, q.w};
            float rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = av[u];
                const float eps = (xv[u] - st.sa_t * a) / st.s1m_t;
                float inner = st.q * eps;
                if (rn.with_n1) inner = inner + st.es * z1[u];
                float v = st.sa_p * a + st.k1 * inner;
                v = v + st.k2 * z2[u];
                rv[u] = v;
            }
            *reinterpret_cast<float4*>(rn.xt + gi) = make_float4(rv[0], rv[1], rv[2], rv[3]);
            continue;
        }
        *reinterpret_cast<float4*>(out + gi) = q;
    }
}

// ------------------------------------------------------------------------------------------------ columns
// A strip of CS = THREADS/R columns of one plane; thread = (column c, t).  MODE 0: forward only; MODE 2: forward ->
// solve (sf = 1: FX = (FR - conj(FB) * (FB*FR)/(F2B+alpha)) / alpha, FR = FBFy + F(alpha x)) -> inverse; MODE 3: the same for sf > 1.
// PERSIST: FBFy (what the solve adds to the transformed data; it does not depend on the row jobs) is requested before the wait.
template <int R, int RJ, int MODE, int THREADS, int SF, bool PERSIST, class Wait, bool PREFETCH = true>
__device__ __forceinline__ void cfft_cols_body(float2* sm2, int plane, int strip, float2* buf, const SolveArgs& a, int WP, const float2* tw, Wait wait) {
    constexpr int N = R * RJ, CS = THREADS / R;
    float2* twN = sm2;
    float2* xch = sm2 + N;                              // [CS][RJ*(R+1)+1]  (+1: lanes of a wave walk the slots)
    constexpr int XST = RJ * (R + 1) + 1;
    constexpr int NTW = PERSIST ? 1 : (N + THREADS - 1) / THREADS;
    float2 twr[NTW];                                    // requested first, stored once the column loads are in flight (see rfft_rows_body)
    if (!PERSIST) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; twr[j] = i < N ? tw[i] : make_float2(0.f, 0.f); }
    }
    const int c = threadIdx.x % CS, t = threadIdx.x / CS;     // lanes walk the strip's columns: 128-byte row segments
    const int col = strip * CS + c;
    float2* base = buf + (size_t)plane * N * WP + col;
    constexpr bool PF = PERSIST && PREFETCH && (MODE == 2 || MODE == 3);
    float2 fy[PF ? RJ : 1];
    if (PF) {
        const float2* FBFy = a.FBFy + (size_t)plane * N * WP + col;
#pragma unroll
        for (int k2 = 0; k2 < RJ; ++k2) fy[k2] = FBFy[(size_t)(t + R * k2) * WP];
    }
    if (PERSIST) wait();
    float2 v[RJ];
#pragma unroll
    for (int n1 = 0; n1 < RJ; ++n1) v[n1] = base[(size_t)(R * n1 + t) * WP];
    // (Requesting the solve's three spectra here as well, so that they travel with the data, made the three-launch kernel 10 % SLOWER -- 38.9
    // vs 35.2 us per apply, three times in one call, profiles/r04/dead_end_prox_cols_operand_prefetch_ab.log: 64 loads in flight per thread
    // delay the 16 the transform is waiting for.  There they are loaded after the forward transform, as in round 3.)
    if (!PERSIST) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; if (i < N) twN[i] = twr[j]; }
        __syncthreads();
    }
    fft_two_pass<R, RJ, false>(v, t, xch + c * XST, twN);
    if (MODE == 2) {
        float alpha = a.sp ? a.sp->tau : a.alpha;
        const int n_img = plane / 3;
        const float2* FB = a.FB + (size_t)n_img * N * WP + col;
        const float* F2B = a.F2B + (size_t)n_img * N * WP + col;
        const float2* FBFy = a.FBFy + (size_t)plane * N * WP + col;
#pragma unroll
        for (int k2 = 0; k2 < RJ; ++k2) {
            size_t off = (size_t)(t + R * k2) * WP;
            float2 fr = cadd(PF ? fy[k2] : FBFy[off], v[k2]);
            float2 fb = FB[off];
            float2 x1 = cmul2(fb, fr);
            float den = F2B[off] + alpha;
            float2 q = make_float2(x1.x / den, x1.y / den);
            float2 tq = cmulc2(q, fb);                          // conj(FB) * q
            v[k2] = make_float2((fr.x - tq.x) / alpha, (fr.y - tq.y) / alpha);
        }
        fft_two_pass<R, RJ, true>(v, t, xch + c * XST, twN);
    }
    if (MODE == 3) {
        // sf > 1 (utils_sisr.py:65-75 with `splits`): FBR = mean over the sf x sf aliases of FB * FR, FX = (FR - conj(FB) R~) / alpha with
        // R = FBR / (invW + alpha) tiled back over the aliases.  Slot c of the strip = alias b = c % sf of fold group c / sf.
        float alpha = a.sp ? a.sp->tau : a.alpha;
        constexpr int sf = SF, Hs = N / SF, KH = RJ / SF, ngrp = CS / SF;
        const int s0 = strip * CS;                                        // first slot of the strip
        const int QW = N / sf / 2 + 1;                                    // fold groups per row: q <= Ws / 2
        float2* sfold = xch + CS * XST;                                   // [CS][Hs] row-folded FB * FR
        float2* Rl = sfold + CS * Hs;                                     // [ngrp][Hs]
        const int n_img = plane / 3;
        const float2* FB = a.FB + (size_t)n_img * N * WP + col;
        const float2* FBFy = a.FBFy + (size_t)plane * N * WP + col;
        // v <- FR = FBFy + F(alpha x); rows u + a Hs of one thread are k2 = k2p + KH a: fold them while FB * FR is formed
        float2 sacc[KH];
#pragma unroll
        for (int i = 0; i < KH; ++i) sacc[i] = make_float2(0.f, 0.f);
#pragma unroll
        for (int k2 = 0; k2 < RJ; ++k2) {
            const size_t off = (size_t)(t + R * k2) * WP;
            v[k2] = cadd(PF ? fy[k2] : FBFy[off], v[k2]);
            sacc[k2 % KH] = cadd(sacc[k2 % KH], cmul2(FB[off], v[k2]));
        }
#pragma unroll
        for (int i = 0; i < KH; ++i) sfold[c * Hs + t + R * i] = sacc[i];
        __syncthreads();
        const float inv_n = 1.0f / (float)(sf * sf);
        for (int item = threadIdx.x; item < ngrp * Hs; item += THREADS) {
            const int ql = item / Hs, p = item - ql * Hs, pm = (Hs - p) % Hs;
            const int q = s0 / sf + ql;
            float2 acc = make_float2(0.f, 0.f);
            for (int b = 0; b < sf; ++b) {
                const int cc = ql * sf + b;
                const int cm = a.slot_col[s0 + cc];
                if (cm < 0) continue;
                if (cm >> 16) { const float2 z = sfold[cc * Hs + pm]; acc.x += z.x; acc.y -= z.y; }       // mirrored alias: conj of the mirrored row
                else acc = cadd(acc, sfold[cc * Hs + p]);
            }
            float2 r = make_float2(0.f, 0.f);
            if (q < QW) {
                const float den = a.invW[((size_t)n_img * Hs + p) * QW + q] + alpha;
                r = make_float2(acc.x * inv_n / den, acc.y * inv_n / den);
            }
            Rl[ql * Hs + p] = r;
        }
        __syncthreads();
        const int cmine = a.slot_col[s0 + c];
        const bool mir = cmine >= 0 && (cmine >> 16);
        const int ql = c / sf;
#pragma unroll
        for (int k2 = 0; k2 < RJ; ++k2) {
            const int p = t + R * (k2 % KH);
            float2 rr = mir ? Rl[ql * Hs + (Hs - p) % Hs] : Rl[ql * Hs + p];
            if (mir) rr.y = -rr.y;
            const float2 tq = cmulc2(rr, FB[(size_t)(t + R * k2) * WP]);   // conj(FB) * R~   (FB re-read: an L2 hit, not 2 RJ live registers)
            v[k2] = make_float2((v[k2].x - tq.x) / alpha, (v[k2].y - tq.y) / alpha);
        }
        fft_two_pass<R, RJ, true>(v, t, xch + c * XST, twN);
    }
#pragma unroll
    for (int k2 = 0; k2 < RJ; ++k2) base[(size_t)(t + R * k2) * WP] = v[k2];
}

// LDS float2 elements the bodies use beyond the twiddle table [N]
template <int R, int RJ, int THREADS> constexpr size_t rows_lds_elems() {
    constexpr size_t xch = (size_t)(THREADS / R) * RJ * (R + 1), zb = (size_t)(THREADS / R) * (R * RJ + 4);
    return xch > zb ? xch : zb;
}
template <int R, int RJ, int THREADS, int SF> constexpr size_t cols_lds_elems() {
    constexpr size_t CS = THREADS / R, N = R * RJ;
    return CS * (RJ * (R + 1) + 1) + (SF > 1 ? CS * (N / SF) + (CS / SF) * (N / SF) : 0);
}

}  // namespace dpir
