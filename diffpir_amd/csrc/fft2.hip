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
#include "fft2_body.h"
#include <vector>

namespace dpir {

// The three passes as one kernel each; the bodies live in fft2_body.h (shared with the persistent single launch, fft3.hip).
// rows forward: grid ceil(total_rows/2 / SLOTS), block THREADS = SLOTS*R.
template <int R, int RJ, int THREADS>
__global__ __launch_bounds__(THREADS) void rfft_rows_kernel(const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out,
                                                         int WP, size_t total_rows, const float2* tw, RowsFuse fu, const int* slot_col) {
    extern __shared__ __attribute__((aligned(16))) float2 sm2[];
    rfft_rows_body<R, RJ, THREADS, false>(sm2, blockIdx.x, x, pa, pb, pm, sp, out, WP, total_rows, tw, fu, slot_col);
}
template <int R, int RJ, int THREADS>
__global__ __launch_bounds__(THREADS) void irfft_rows_kernel(const float2* in, float* out, float scale, float oa, float ob,
                                                          const float* blend_base, float g, int WP, size_t total_rows, const float2* tw, RenoiseFuse rn,
                                                          const int* col_slot) {
    extern __shared__ __attribute__((aligned(16))) float2 sm2[];
    irfft_rows_body<R, RJ, THREADS, false>(sm2, blockIdx.x, in, out, scale, oa, ob, blend_base, g, WP, total_rows, tw, rn, col_slot, NoWait{});
}
// columns: one workgroup per (plane, strip of CS = THREADS/R columns)
template <int R, int RJ, int MODE, int THREADS, int SF>
__global__ __launch_bounds__(THREADS) void cfft_cols_kernel(float2* buf, SolveArgs a, int WP, const float2* tw) {
    extern __shared__ __attribute__((aligned(16))) float2 sm2[];
    constexpr int CS = THREADS / R;
    const int strips = WP / CS;
    const int plane = blockIdx.x / strips;
    cfft_cols_body<R, RJ, MODE, THREADS, SF, false>(sm2, plane, blockIdx.x - plane * strips, buf, a, WP, tw, NoWait{});
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
