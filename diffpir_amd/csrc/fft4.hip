// fft4: the half-spectrum FFT prox at N = 256 and N = 512 (utils/utils_sisr.py:9-19, 65-95; the headline deblurring / SR configurations) with ONE WAVE PER
// TRANSFORM (fft4_wave.h) and a COLUMN-MAJOR half spectrum [plane][slot][position].  Same mathematics as fft2.hip (row-pair packing, half spectrum,
// closed-form solve between the column transforms), different mapping to the machine (bodies: fft4_body.h):
//   * rows    : a workgroup = 8 waves = 8 pairs of real rows (row r with r + 64): float4 loads, wave-private LDS re-distribution, 256-point FFT in the wave,
//               Hermitian un-packing into an LDS tile [slot][8 + 1] of 16-byte {A[k], B[k]} entries, then FULL 128-byte lines (16 positions of one slot) to memory;
//   * columns : a wave takes one column (2 KB contiguous; position order: two 1 KB wave loads): forward FFT -> closed-form solve on registers whose operands
//               (FBFy, FB, F2B, stored the same way by pre_calculate) are loaded with the column -> inverse FFT -> store.  No workgroup barrier (sf = 1), no strips,
//               no padding columns (129 stored columns instead of fft2's 144);
//   * rows inv: the workgroup loads its 16 positions of columns 0..128 as full lines into the tile, each wave re-packs its pair (Hermitian), inverse FFT, float4
//               epilogue (x*2-1, guidance blend, or the fused re-noise with Philox draws per float4 group exactly as fft2's).
// A batch-16 apply is 6 waves per SIMD in every pass (fft2: 1.5) at 40-57 VGPRs.  Measured 24.9 us per apply against fft2's 33.0.
// sf > 1 keeps fft2's alias-grouped SLOT order (slot sf q + b = alias b of fold group q): a workgroup's four column waves are four consecutive slots, i.e.
// one fold group at sf = 4 and two at sf = 2; the row aliases u + a N/sf of a column live in ONE lane (registers j, since N/sf is a multiple of 64).
#include "fft4_body.h"
#include <vector>
#include <algorithm>
#include <stdlib.h>

namespace dpir {

// the three passes as one launch each (named kernels: they are what the rocprofv3 summaries under profiles/ list) (bodies: fft4_body.h)
template <int N>
__global__ __launch_bounds__(RTHREADS) void rfft4_rows_kernel(const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out, int NC,
                                                             const float2* tw, RowsFuse fu, const int* slot_col) {
    extern __shared__ __attribute__((aligned(16))) float2 sm4[];
    rows4_body<N, false>(sm4, blockIdx.x, x, pa, pb, pm, sp, out, NC, tw, WaveTwN<N>{}, fu, slot_col);
}
template <int N>
__global__ __launch_bounds__(RTHREADS) void irfft4_rows_kernel(const float2* in, float* out, float scale, float oa, float ob, const float* blend_base, float g,
                                                              int NC, const float2* tw, RenoiseFuse rn, const int* col_slot) {
    extern __shared__ __attribute__((aligned(16))) float2 sm4[];
    irows4_body<N, false>(sm4, blockIdx.x, in, out, scale, oa, ob, blend_base, g, NC, tw, WaveTwN<N>{}, rn, col_slot, NoWait4{});
}
// grid: P * ceil(NC / 4) workgroups of four waves: one item each
template <int MODE, int SF, int N>
__global__ __launch_bounds__(THREADS4) void cfft4_cols_kernel(float2* buf, SolveArgs a, int NC, const float2* tw) {
    extern __shared__ __attribute__((aligned(16))) float2 sm4[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = (NC + WAVES - 1) / WAVES;
    int plane, item;
    if (MODE >= 2) {
        // The solve's FB / F2B (/ invW) belong to the IMAGE: the three colour planes of an (image, item) pair read the same 3-6 KB per wave.  Workgroup b runs on
        // XCD b % 8 (own L2), so the pair's three workgroups are numbered 8 apart -- same XCD, dispatched back to back: the second and third hit that L2 instead of
        // fetching the spectra again over the fabric (counter traffic of the batch-16 column pass: 57.2 -> see profiles).  Grid: 3 x (pairs rounded up to 8).
        const int r = blockIdx.x & 7, t = blockIdx.x >> 3;
        const int c = t % 3, pair = 8 * (t / 3) + r;
        const int n = pair / groups;
        if (n >= a.images) return;                                     // padding workgroups (whole workgroup: no barrier is left waiting)
        item = pair - n * groups;
        plane = 3 * n + c;
    } else {
        plane = blockIdx.x / groups;
        item = blockIdx.x - plane * groups;
    }
    const WaveTwN<N> w = wave_tw_load<N>(tw, lane);
    cols4_item_body<MODE, SF, N>(sm4 + wave * wlds(N), sm4 + WAVES * wlds(N), plane, item, wave, buf, a, NC, w, NoWait4{});
}

// invW[n, p, q] = mean over the sf x sf aliases of F2B (utils_sisr.py:71), column-major slots
__global__ void fold_f2b4_kernel(const float* F2B, const int* slot_col, int N, int NC, int sf, float* invW, size_t total) {
    const int Hs = N / sf, QW = N / sf / 2 + 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % QW);
        const int p = (int)((i / QW) % Hs);
        const size_t n = i / ((size_t)QW * Hs);
        const float* pl = F2B + n * (size_t)NC * N;
        float acc = 0.f;
        for (int b = 0; b < sf; ++b) {
            const int slot = sf * q + b;
            const int cm = slot < NC ? slot_col[slot] : -1;
            if (cm < 0) continue;
            const int base_row = (cm >> 16) ? (Hs - p) % Hs : p;          // |FB|^2 is real: the mirrored alias is just the mirrored row
            for (int a = 0; a < sf; ++a) acc += pl[(size_t)slot * N + pos4(base_row + a * Hs)];
        }
        invW[i] = acc / (float)(sf * sf);
    }
}

int fft4_row_pos(int u) { return pos4(u); }
bool fft4_supported(int H, int W, int sf) { return H == W && (H == 256 || H == 512) && (sf == 1 || sf == 2 || sf == 4); }
// stored columns (slots) per plane: W/2 + 1 for sf = 1, sf * (W/sf/2 + 1) alias-grouped slots otherwise
int fft4_columns(int W, int sf) { return sf == 1 ? W / 2 + 1 : sf * (W / sf / 2 + 1); }
// slot -> (column | mirrored << 16) or -1, column -> canonical slot: fft2's alias grouping without strip padding
void fft4_build_map(int N, int sf, std::vector<int>& slot_col, std::vector<int>& col_slot) {
    const int NC = fft4_columns(N, sf), Ws = N / sf;
    slot_col.assign(NC, -1);
    col_slot.assign(N / 2 + 1, -1);
    for (int q = 0; q <= Ws / 2; ++q)
        for (int b = 0; b < sf; ++b) {
            const int c = q + b * Ws;
            if (c >= N) continue;
            const int col = c <= N / 2 ? c : N - c, mir = c <= N / 2 ? 0 : 1;
            slot_col[sf * q + b] = col | (mir << 16);
            if (col_slot[col] < 0 || (!mir && (slot_col[col_slot[col]] >> 16))) col_slot[col] = sf * q + b;      // prefer the direct copy
        }
}

static size_t lds4(int N, bool fold, int sf) { return ((size_t)WAVES * wlds(N) + (fold ? (size_t)WAVES * (N / sf) : 0)) * sizeof(float2); }
static size_t lds4_rows(int N, int NC) { return (size_t)RW * wlds(N) * sizeof(float2) + (size_t)NC * TST * sizeof(float4); }

// dynamic LDS above 64 KB (the 512-point row passes: 8 exchange tiles + the [slot][9] line tile = 74 KB) has to be allowed per kernel, once
template <class K> static Status allow_lds(K kernel, size_t bytes) {
    static size_t allowed[64];                  // per instantiation (= per kernel) and device: raise the limit only when a launch needs more than was granted so far
    if (bytes <= 64 * 1024) return Status{};
    int dev = 0;
    DPIR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || bytes > allowed[dev]) {
        DPIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        if (dev >= 0 && dev < 64) allowed[dev] = bytes;
    }
    return Status{};
}

template <int N>
static Status rows_fwd(hipStream_t s, const float2* tw, const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out, int P, int NC,
                       const float* eps6, int out_ch, const int* slot_col) {
    const size_t pairs = (size_t)P * N / 2;             // a multiple of RW: no partial workgroup
    DPIR_TRY(allow_lds(rfft4_rows_kernel<N>, lds4_rows(N, NC)));
    hipLaunchKernelGGL(rfft4_rows_kernel<N>, dim3((unsigned)(pairs / RW)), dim3(RTHREADS), lds4_rows(N, NC), s, x, pa, pb, pm, sp, out, NC, tw,
                       RowsFuse{eps6, out_ch}, slot_col);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_rfft4_rows(hipStream_t s, const float2* tw, int N, const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out, int P, int NC,
                         const float* eps6, int out_ch, const int* slot_col) {
    if (eps6 && !sp) return invalid("rfft4_rows: the fused x0 prologue reads its coefficients from the device step block");
    if (N == 256) return rows_fwd<256>(s, tw, x, pa, pb, pm, sp, out, P, NC, eps6, out_ch, slot_col);
    if (N == 512) return rows_fwd<512>(s, tw, x, pa, pb, pm, sp, out, P, NC, eps6, out_ch, slot_col);
    return invalid("rfft4_rows: N must be 256 or 512");
}
template <int N>
static Status rows_inv(hipStream_t s, const float2* tw, const float2* in, float* out, float scale, float oa, float ob, const float* blend, float g, int P,
                       int NC, RenoiseFuse rn, const int* col_slot) {
    const size_t pairs = (size_t)P * N / 2;
    DPIR_TRY(allow_lds(irfft4_rows_kernel<N>, lds4_rows(N, N / 2 + 1)));
    hipLaunchKernelGGL(irfft4_rows_kernel<N>, dim3((unsigned)(pairs / RW)), dim3(RTHREADS), lds4_rows(N, N / 2 + 1), s, in, out, scale, oa, ob, blend, g, NC,
                       tw, rn, col_slot);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_irfft4_rows(hipStream_t s, const float2* tw, int N, const float2* in, float* out, float scale, float oa, float ob, const float* blend, float g,
                          int P, int NC, const RenoiseArgs* ra, const int* col_slot) {
    RenoiseFuse rn{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (ra) rn = RenoiseFuse{ra->xt, ra->sp, ra->lp, ra->n1, ra->n2, ra->stride, ra->with_n1};
    if (N == 256) return rows_inv<256>(s, tw, in, out, scale, oa, ob, blend, g, P, NC, rn, col_slot);
    if (N == 512) return rows_inv<512>(s, tw, in, out, scale, oa, ob, blend, g, P, NC, rn, col_slot);
    return invalid("irfft4_rows: N must be 256 or 512");
}
template <int N>
static Status cols(hipStream_t s, const float2* tw, float2* buf, const SolveArgs& a_in, bool solve, int P, int NC) {
    const int groups = (NC + WAVES - 1) / WAVES;
    unsigned grid = (unsigned)(P * groups);
    SolveArgs a = a_in;
    if (solve) {
        if (P % 3) return invalid("cfft4_cols: the solve runs on the three colour planes of every image");
        a.images = P / 3;
        grid = 3u * (unsigned)(((size_t)(P / 3) * groups + 7) / 8 * 8);
    }
    if (!solve) hipLaunchKernelGGL((cfft4_cols_kernel<0, 1, N>), dim3(grid), dim3(THREADS4), lds4(N, false, 1), s, buf, a, NC, tw);
    else if (a.sf == 1) hipLaunchKernelGGL((cfft4_cols_kernel<2, 1, N>), dim3(grid), dim3(THREADS4), lds4(N, false, 1), s, buf, a, NC, tw);
    else {
        if ((a.sf != 2 && a.sf != 4) || !a.invW || !a.slot_col || NC % a.sf) return invalid("cfft4_cols: bad sf > 1 arguments");
        if (a.sf == 2) hipLaunchKernelGGL((cfft4_cols_kernel<3, 2, N>), dim3(grid), dim3(THREADS4), lds4(N, true, 2), s, buf, a, NC, tw);
        else hipLaunchKernelGGL((cfft4_cols_kernel<3, 4, N>), dim3(grid), dim3(THREADS4), lds4(N, true, 4), s, buf, a, NC, tw);
    }
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_cfft4_cols(hipStream_t s, const float2* tw, int N, float2* buf, const SolveArgs& a, bool solve, int P, int NC) {
    if (N == 256) return cols<256>(s, tw, buf, a, solve, P, NC);
    if (N == 512) return cols<512>(s, tw, buf, a, solve, P, NC);
    return invalid("cfft4_cols: N must be 256 or 512");
}
Status launch_fold_f2b4(hipStream_t s, const float* F2B, const int* slot_col, int N, int NC, int sf, float* invW, int B) {
    const size_t total = (size_t)B * (N / sf) * (N / sf / 2 + 1);
    hipLaunchKernelGGL(fold_f2b4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, F2B, slot_col, N, NC, sf, invW, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
