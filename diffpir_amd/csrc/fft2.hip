// fft2: half-spectrum, register-resident FFT path of the data-fidelity prox at N = 64 / 256 for sf = 1 (deblurring, the headline
// configuration) and sf = 2 / 4 (super-resolution, round 3).  Replaces the same reference lines as fft.hip
// (utils/utils_sisr.py:9-19, 65-95); the radix-2 / full-c2c kernels of fft.hip remain the general path (other sizes).
//
// sf > 1: the closed form averages FB*FR over the sf x sf spectral aliases (u + a H/sf, v + b W/sf) -- `splits` + mean in the
// reference.  On HALF spectra an alias column beyond W/2 is the conjugate of the mirrored row of a stored column, so the sf stored
// columns {q + b Ws <= W/2} u {W - (q + b Ws)} meet in one fold (index algebra prototyped in oracle/half_spectrum_sr.py).  The
// spectra are therefore stored COLUMN-PERMUTED: slot sf*q + b holds alias b of fold group q (q <= Ws/2), so that a 16-slot column
// strip is 16/sf complete groups: 128-byte row segments exactly as for sf = 1, the row aliases u + a Hs stay inside the thread
// that holds rows t + R k2 of the column FFT (Hs = R * R/sf), and only the column fold and the row mirror go through LDS.
//
// Structure (per batch of P = 3B image planes, HBM-bound):
//   rfft_rows   : two REAL rows are packed into one complex transform (z = a + i b), N = Ra x Rb two-pass FFT held in
//                 registers (16 x 16 at N = 256: one LDS exchange between the passes), un-packed with the Hermitian
//                 identity into two half-spectrum rows (N/2+1 columns, padded to a multiple of the strip width);
//   cfft_cols   : a strip of columns per workgroup: forward FFT -> closed-form spectral solve -> inverse FFT without
//                 leaving registers (the output distribution of the forward pass-2 IS the input distribution of the
//                 inverse pass-1), two LDS exchanges in total;
//   irfft_rows  : Hermitian re-packing of two half-spectrum rows into one complex inverse transform, real / imag parts
//                 are the two output rows; epilogue x*2-1 and the guidance blend.
// Algorithmic HBM bytes per image per step at 256^2: 2.50 MB; the padded half-spectrum intermediate
// (3 x 256 x 144 x 8 B = 0.88 MB, written once and read once by each neighbour kernel) stays in L2 / Infinity Cache.
#include "common.h"
#include "elem.h"
#include "philox.h"
#include "fft_regs.h"
#include <vector>

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

// ------------------------------------------------------------------------------------------------ rows forward
// One slot = one PAIR of real rows.  grid: ceil(total_rows/2 / SLOTS); block 256 = SLOTS*R threads.
// Fused loop prologue (dpir_run_loop): when `eps6` is given, the row loaded is not x but the denoiser's clamped x0 prediction
// x0 = clamp(c1*x - c2*eps, -1, 1) (gaussian_diffusion.py:297,328-333), evaluated while staging -- x0 is never materialised.
struct RowsFuse { const float* eps6; int out_ch; };
template <int R, int RJ, int THREADS>
__global__ __launch_bounds__(THREADS) void rfft_rows_kernel(const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out,
                                                         int WP, size_t total_rows, const float2* tw, RowsFuse fu, const int* slot_col) {
    constexpr int N = R * RJ, SLOTS = THREADS / R;
    extern __shared__ __attribute__((aligned(16))) float2 sm2[];
    float2* twN = sm2;                                  // [N]
    float2* xch = sm2 + N;                              // [SLOTS][RJ*(R+1)]
    // [SLOTS][N+4] natural-order Z of each slot (also the load staging).  ALIASED with the exchange area: staging is dead once the
    // two-pass layout has been gathered into registers, the exchange area is dead when fft_two_pass returns (each hand-over is a
    // __syncthreads) -- 19 -> 10.7 KiB per workgroup at N = 256, 8 -> 14 resident workgroups per CU (pays at B >= 64)
    float2* zbuf = xch;
    if (sp) pm = sp->tau;
    // the twiddle table is requested FIRST but stored to LDS only after the row loads below are in flight too: a load -> ds_write pair
    // in front of them would be a whole memory round trip before the first row request leaves the CU
    constexpr int NTW = (N + THREADS - 1) / THREADS;
    float2 twr[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; twr[j] = i < N ? tw[i] : make_float2(0.f, 0.f); }
    const int slot = threadIdx.x / R, t = threadIdx.x % R;
    const size_t pair = (size_t)blockIdx.x * SLOTS + slot;
    const size_t ra = 2 * pair, rb = 2 * pair + 1;
    const bool va = ra < total_rows, vb = rb < total_rows;
    // coalesced float4 loads of the block's 2*SLOTS rows into LDS, then the strided gather of the two-pass layout
    float* stage = reinterpret_cast<float*>(zbuf);          // [2*SLOTS][N + 4] floats == SLOTS*(N+4) float2
    {
        const size_t row0 = (size_t)blockIdx.x * SLOTS * 2;
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
#pragma unroll
        for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; if (i < N) twN[i] = twr[j]; }
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
struct RenoiseFuse { float* xt; const StepDev* sp; const LoopDev* lp; const float* n1; const float* n2; size_t stride; int with_n1; };
template <int R, int RJ, int THREADS>
__global__ __launch_bounds__(THREADS) void irfft_rows_kernel(const float2* in, float* out, float scale, float oa, float ob,
                                                          const float* blend_base, float g, int WP, size_t total_rows, const float2* tw, RenoiseFuse rn,
                                                          const int* col_slot) {
    constexpr int N = R * RJ, SLOTS = THREADS / R;
    extern __shared__ __attribute__((aligned(16))) float2 sm2[];
    float2* twN = sm2;
    float2* xch = sm2 + N;
    float2* zbuf = xch;                                 // aliased with the exchange area (see rfft_rows_kernel)
    constexpr int NTW = (N + THREADS - 1) / THREADS;
    float2 twr[NTW];                                    // requested first, stored after the spectrum loads are in flight (see rfft_rows_kernel)
#pragma unroll
    for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; twr[j] = i < N ? tw[i] : make_float2(0.f, 0.f); }
    const int slot = threadIdx.x / R, t = threadIdx.x % R;
    const size_t pair = (size_t)blockIdx.x * SLOTS + slot;
    const size_t ra = 2 * pair, rb = 2 * pair + 1;
    const bool va = ra < total_rows, vb = rb < total_rows;
    float2* z = zbuf + slot * (N + 4);
    // Hermitian re-packing: Z[k] = A[k] + i B[k], Z[N-k] = conj(A[k]) + i conj(B[k])
    constexpr int NK = (N / 2 + 1 + R - 1) / R;               // all loads in flight before the first LDS write (see rfft_rows_kernel)
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
    // what the epilogue combines with the transform's result (x_t for the fused re-noise, or the guidance-blend base) does not depend on
    // it: requested here, so that its latency hides under the transform instead of following it
    const size_t row0 = (size_t)blockIdx.x * SLOTS * 2;
    constexpr int V4 = N / 4;
    constexpr int NS4 = (2 * SLOTS * V4 + THREADS - 1) / THREADS;
    const float* pre_src = rn.xt ? rn.xt : blend_base;
    float4 pre[NS4];
#pragma unroll
    for (int j = 0; j < NS4; ++j) {
        const int i = threadIdx.x + j * THREADS;
        const int r = i / V4, c4 = i - r * V4;
        pre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pre_src && i < 2 * SLOTS * V4 && row0 + r < total_rows) pre[j] = *reinterpret_cast<const float4*>(pre_src + (row0 + r) * N + c4 * 4);
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; if (i < N) twN[i] = twr[j]; }
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
// A strip of CS = 256/R columns per workgroup; thread = (column c, t).  MODE 0: forward only; MODE 2: forward ->
// solve (sf = 1: FX = (FR - conj(FB) * (FB*FR)/(F2B+alpha)) / alpha, FR = FBFy + F(alpha x)) -> inverse.
template <int R, int RJ, int MODE, int THREADS, int SF>
__global__ __launch_bounds__(THREADS) void cfft_cols_kernel(float2* buf, SolveArgs a, int WP, const float2* tw) {
    constexpr int N = R * RJ, CS = THREADS / R;
    extern __shared__ __attribute__((aligned(16))) float2 sm2[];
    float2* twN = sm2;
    float2* xch = sm2 + N;                              // [CS][RJ*(R+1)+1]  (+1: lanes of a wave walk the slots)
    constexpr int XST = RJ * (R + 1) + 1;
    constexpr int NTW = (N + THREADS - 1) / THREADS;
    float2 twr[NTW];                                    // requested first, stored once the column loads are in flight (see rfft_rows_kernel)
#pragma unroll
    for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; twr[j] = i < N ? tw[i] : make_float2(0.f, 0.f); }
    const int c = threadIdx.x % CS, t = threadIdx.x / CS;     // lanes walk the strip's columns: 128-byte row segments
    const int strips = WP / CS;
    const int plane = blockIdx.x / strips;
    const int col = (blockIdx.x - plane * strips) * CS + c;
    float2* base = buf + (size_t)plane * N * WP + col;
    float2 v[RJ];
#pragma unroll
    for (int n1 = 0; n1 < RJ; ++n1) v[n1] = base[(size_t)(R * n1 + t) * WP];
    // (Requesting the solve's three spectra here as well, so that they travel with the data, made this kernel 10 % SLOWER -- 38.9 vs 35.2 us
    // per apply, three times in one call, profiles/r04/dead_end_prox_cols_operand_prefetch_ab.log: 64 loads in flight per thread delay the
    // 16 the transform is waiting for.  They are loaded after the forward transform, as in round 3.)
#pragma unroll
    for (int j = 0; j < NTW; ++j) { const int i = threadIdx.x + j * THREADS; if (i < N) twN[i] = twr[j]; }
    __syncthreads();
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
            float2 fr = cadd(FBFy[off], v[k2]);
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
        const int s0 = (blockIdx.x - plane * strips) * CS;               // first slot of the strip
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
            v[k2] = cadd(FBFy[off], v[k2]);
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

// invW[n, p, q] = mean over the sf x sf aliases of F2B (utils_sisr.py:71 `invW = mean(splits(F2B))`), from the permuted half layout
__global__ void fold_f2b_kernel(const float* F2B, const int* slot_col, int N, int WP, int sf, float* invW, size_t total) {
    const int Hs = N / sf, QW = N / sf / 2 + 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % QW);
        const int p = (int)((i / QW) % Hs);
        const size_t n = i / ((size_t)QW * Hs);
        const float* pl = F2B + n * (size_t)N * WP;
        float acc = 0.f;
        for (int b = 0; b < sf; ++b) {
            const int slot = sf * q + b;
            const int cm = slot_col[slot];
            if (cm < 0) continue;
            const int base_row = (cm >> 16) ? (Hs - p) % Hs : p;          // |FB|^2 is real: the mirrored alias is just the mirrored row
            for (int a = 0; a < sf; ++a) acc += pl[(size_t)(base_row + a * Hs) * WP + slot];
        }
        invW[i] = acc / (float)(sf * sf);
    }
}
// zero-stuffed up-sampling of the measurement as a REAL image (utils_sisr.upsample, :44-52)
__global__ void upsample_real_kernel(const float* y, int sf, float* out, int h, int w, size_t total) {
    const int H = h * sf, W = w * sf;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % W), Y = (int)((i / W) % H);
        const size_t p = i / ((size_t)H * W);
        out[i] = (Y % sf == 0 && X % sf == 0) ? y[(p * h + Y / sf) * w + X / sf] : 0.f;
    }
}

// FBFy <- conj(FB) * F(y); F2B = |FB|^2 on the padded half-spectrum layout
__global__ void precalc_finish2_kernel(const float2* FB, float2* FBFy, float* F2B, size_t hw, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i / hw, r = i - p * hw;
        size_t b = p / 3;
        float2 fb = FB[b * hw + r];
        FBFy[i] = cmulc2(FBFy[i], fb);
        if (p % 3 == 0) { float m = hypotf(fb.x, fb.y); F2B[b * hw + r] = m * m; }
    }
}
// p2o embedding as a REAL image (natural order): out[y][x] = k[(y + kh/2) % H][(x + kw/2) % W] inside the PSF support
__global__ void psf_embed_real_kernel(const float* k, int kh, int kw, float* out, int H, int W, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t b = i / ((size_t)H * W);
        size_t r = i - b * (size_t)H * W;
        int y = (int)(r / W), x = (int)(r - (size_t)y * W);
        int ky = (y + kh / 2) % H, kx = (x + kw / 2) % W;
        out[i] = (ky < kh && kx < kw) ? k[(b * kh + ky) * kw + kx] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
bool fft2_supported(int H, int W, int sf) { return (sf == 1 || sf == 2 || sf == 4) && H == W && (H == 512 || H == 256 || H == 64); }
// slot -> (column | mirrored << 16) or -1 (padding), and column -> canonical slot, for the alias-grouped layout of sf > 1
void fft2_build_map(int N, int sf, std::vector<int>& slot_col, std::vector<int>& col_slot) {
    const int WP = fft2_padded_width(N), Ws = N / sf;
    slot_col.assign(WP, -1);
    col_slot.assign(N / 2 + 1, -1);
    for (int q = 0; q <= Ws / 2; ++q)
        for (int b = 0; b < sf; ++b) {
            const int c = q + b * Ws;
            const int col = c <= N / 2 ? c : N - c, mir = c <= N / 2 ? 0 : 1;
            slot_col[sf * q + b] = col | (mir << 16);
            if (col_slot[col] < 0 || (!mir && (slot_col[col_slot[col]] >> 16))) col_slot[col] = sf * q + b;      // prefer the direct copy
        }
}
int fft2_padded_width(int W) { const int cs = 16; return (W / 2 + 1 + cs - 1) / cs * cs; }   // multiple of the column strip

constexpr int ROW_THREADS = 64;    // small workgroups: at B = 16 the whole prox is ~40 MB, concurrency comes from block count
// 16 columns per strip for both sizes: a strip row is one full 128-byte line (8 columns = half lines cost ~2x the requests)
template <int R> struct ColCfg { static constexpr int THREADS = 16 * R; };
template <int R, int RJ>
static size_t rows_lds() {      // twiddles + max(exchange area, natural-order / staging area): the two are aliased
    constexpr size_t xch = (size_t)(ROW_THREADS / R) * RJ * (R + 1), zb = (size_t)(ROW_THREADS / R) * (R * RJ + 4);
    return (R * RJ + (xch > zb ? xch : zb)) * sizeof(float2);
}
template <int R, int RJ>
static size_t cols_lds() { return (size_t)(R * RJ + (ColCfg<R>::THREADS / R) * (RJ * (R + 1) + 1)) * sizeof(float2); }

template <int R, int RJ>
static Status rfft_rows_R(hipStream_t s, const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out, int P, int N,
                          const float2* tw, RowsFuse fu, const int* slot_col) {
    int WP = fft2_padded_width(N);
    size_t rows = (size_t)P * N, pairs = (rows + 1) / 2;
    constexpr int SLOTS = ROW_THREADS / R;
    auto fn = rfft_rows_kernel<R, RJ, ROW_THREADS>;
    static LdsAttrOnce attr;
    DPIR_HIP(attr.set(reinterpret_cast<const void*>(fn), 160 * 1024));
    const size_t lds = rows_lds<R, RJ>();
    hipLaunchKernelGGL(fn, dim3((unsigned)((pairs + SLOTS - 1) / SLOTS)), dim3(ROW_THREADS), lds, s, x, pa, pb, pm, sp, out, WP, rows, tw, fu, slot_col);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_rfft_rows(hipStream_t s, const float2* twN, const float* x, float pa, float pb, float pm, const StepDev* sp,
                        float2* out, int P, int N, const float* eps6, int out_ch, const int* slot_col) {
    if (eps6 && !sp) return invalid("rfft_rows: the fused x0 prologue reads its coefficients from the device step block");
    const RowsFuse fu{eps6, out_ch};
    if (N == 512) return rfft_rows_R<16, 32>(s, x, pa, pb, pm, sp, out, P, N, twN, fu, slot_col);
    return N == 256 ? rfft_rows_R<16, 16>(s, x, pa, pb, pm, sp, out, P, N, twN, fu, slot_col) : rfft_rows_R<8, 8>(s, x, pa, pb, pm, sp, out, P, N, twN, fu, slot_col);
}

template <int R, int RJ>
static Status irfft_rows_R(hipStream_t s, const float2* in, float* out, float scale, float oa, float ob, const float* blend, float g,
                           int P, int N, const float2* tw, RenoiseFuse rn, const int* col_slot) {
    int WP = fft2_padded_width(N);
    size_t rows = (size_t)P * N, pairs = (rows + 1) / 2;
    constexpr int SLOTS = ROW_THREADS / R;
    auto fn = irfft_rows_kernel<R, RJ, ROW_THREADS>;
    static LdsAttrOnce attr;
    DPIR_HIP(attr.set(reinterpret_cast<const void*>(fn), 160 * 1024));
    const size_t lds = rows_lds<R, RJ>();
    hipLaunchKernelGGL(fn, dim3((unsigned)((pairs + SLOTS - 1) / SLOTS)), dim3(ROW_THREADS), lds, s, in, out, scale, oa, ob, blend, g, WP,
                       rows, tw, rn, col_slot);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_irfft_rows(hipStream_t s, const float2* twN, const float2* in, float* out, float scale, float oa, float ob,
                         const float* blend, float g, int P, int N, const RenoiseArgs* ra, const int* col_slot) {
    RenoiseFuse rn{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (ra) rn = RenoiseFuse{ra->xt, ra->sp, ra->lp, ra->n1, ra->n2, ra->stride, ra->with_n1};
    if (N == 512) return irfft_rows_R<16, 32>(s, in, out, scale, oa, ob, blend, g, P, N, twN, rn, col_slot);
    return N == 256 ? irfft_rows_R<16, 16>(s, in, out, scale, oa, ob, blend, g, P, N, twN, rn, col_slot)
                    : irfft_rows_R<8, 8>(s, in, out, scale, oa, ob, blend, g, P, N, twN, rn, col_slot);
}

template <int R, int RJ, int MODE, int SF = 1>
static Status cfft_cols_RM(hipStream_t s, float2* buf, const SolveArgs& a, int P, int N, const float2* tw) {
    int WP = fft2_padded_width(N);
    constexpr int COL_THREADS = ColCfg<R>::THREADS;
    constexpr int CS = COL_THREADS / R;
    size_t extra = 0;
    if (MODE == 3) {
        if (a.sf != SF || SF < 2 || RJ % SF || CS % SF || !a.invW || !a.slot_col) return invalid("cfft_cols: bad sf > 1 arguments");
        extra = ((size_t)CS * (N / a.sf) + (size_t)(CS / a.sf) * (N / a.sf)) * sizeof(float2);
    }
    auto fn = cfft_cols_kernel<R, RJ, MODE, COL_THREADS, SF>;
    static LdsAttrOnce attr;
    DPIR_HIP(attr.set(reinterpret_cast<const void*>(fn), 160 * 1024));
    const size_t lds = cols_lds<R, RJ>() + extra;
    hipLaunchKernelGGL(fn, dim3((unsigned)(P * (WP / CS))), dim3(COL_THREADS), lds, s, buf, a, WP, tw);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_cfft_cols(hipStream_t s, const float2* twN, float2* buf, const SolveArgs& a, bool solve, int P, int N) {
    const int m = !solve ? 0 : (a.sf > 1 ? 3 : 2);
    if (m == 3 && a.sf != 2 && a.sf != 4) return invalid("cfft_cols: sf must be 2 or 4 on the half-spectrum path");
#define DPIR_COLS(RT, RJ_)                                                                                              \
    do {                                                                                                                \
        if (m == 3) return a.sf == 2 ? cfft_cols_RM<RT, RJ_, 3, 2>(s, buf, a, P, N, twN) : cfft_cols_RM<RT, RJ_, 3, 4>(s, buf, a, P, N, twN); \
        return m == 2 ? cfft_cols_RM<RT, RJ_, 2>(s, buf, a, P, N, twN) : cfft_cols_RM<RT, RJ_, 0>(s, buf, a, P, N, twN);    \
    } while (0)
    if (N == 512) DPIR_COLS(16, 32);
    if (N == 256) DPIR_COLS(16, 16);
    DPIR_COLS(8, 8);
#undef DPIR_COLS
}
Status launch_fold_f2b(hipStream_t s, const float* F2B, const int* slot_col, int N, int sf, float* invW, int B) {
    const size_t total = (size_t)B * (N / sf) * (N / sf / 2 + 1);
    hipLaunchKernelGGL(fold_f2b_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, F2B, slot_col, N, fft2_padded_width(N), sf, invW, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_upsample_real(hipStream_t s, const float* y, int sf, float* out, int P, int h, int w) {
    const size_t total = (size_t)P * h * sf * w * sf;
    hipLaunchKernelGGL(upsample_real_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, y, sf, out, h, w, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_precalc_finish2(hipStream_t s, const float2* FB, float2* FBFy, float* F2B, int B, size_t hw) {
    size_t total = (size_t)B * 3 * hw;
    hipLaunchKernelGGL(precalc_finish2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, FB, FBFy, F2B, hw, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_psf_embed_real(hipStream_t s, const float* k, int kh, int kw, float* out, int B, int H, int W) {
    if (kh > H || kw > W) return invalid("PSF larger than the image");
    size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(psf_embed_real_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k, kh, kw, out, H, W, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
