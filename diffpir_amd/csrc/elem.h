// Declarations for elem.hip / fft.hip launchers.
#pragma once
#include "common.h"

namespace dpir {

struct RenoiseCoef { float sa_t, s1m_t, sa_p, k1, q, es, k2; };

// Device-resident copy of the CURRENT step's scalars.  Inside dpir_run_loop every kernel that needs a per-step
// scalar reads it from here (fixed address), so that one captured hipGraph serves all steps: the host only
// copies steps_dev[i] -> cur (16 words, stream-ordered D2D) before each replay.
struct StepDev {
    int t, last, i, pad;
    float c1, c2, tau, sa_t, s1m_t, sa_p, k1, q, es, k2;
};

// Per-batch values of dpir_run_loop that kernels read on the device (fixed address): with them out of the kernel
// arguments, ONE captured step graph serves every batch of the same shape (new y / mask / noise pointers, seed and
// image offset only rewrite this 64-byte block).
struct LoopDev {
    const float* y; const uint8_t* mask;
    const float* n1; const float* n2; const float* nrp;     // host-fed noise tensors (parity mode) or null
    unsigned long long seed; long long image_offset;
};

Status launch_xstart(hipStream_t s, const float* x, const float* out6, int out_ch, float c1, float c2, float* x0, int B, int HW, const StepDev* sp = nullptr);
Status launch_prox_mask(hipStream_t s, float* x0, const float* y, const uint8_t* mask, float tau, float g, size_t total, const StepDev* sp = nullptr,
                        const LoopDev* lp = nullptr);
Status launch_renoise(hipStream_t s, float* x, const float* x0, const RenoiseCoef& c, const float* n1, const float* n2, size_t total,
                      const StepDev* sp = nullptr, size_t noise_step_stride = 0, const LoopDev* lp = nullptr);
// repaint conditioning before the denoiser call; sp != null: coefficients and the host-noise step offset come from the device step
Status launch_repaint_mix(hipStream_t s, float* x, const float* y, const uint8_t* mask, const float* n, float sa, float s1m, size_t total,
                          const StepDev* sp = nullptr, size_t noise_step_stride = 0, const LoopDev* lp = nullptr);
Status launch_init_x(hipStream_t s, const float* src, const uint8_t* mask, const float* noise, float sa, float s1m, float* x, size_t total);
Status launch_finalize(hipStream_t s, const float* x, float* of, uint8_t* ou, int B, int HW);
// out[i] = x[i] op rhs, rhs = y[i] (y_numel == total) | y[0] (y_numel == 1) | scalar (y == null); op: 0 add, 1 sub, 2 mul, 3 div, 4 rhs - x, 5 rhs / x
Status launch_ewise(hipStream_t s, int op, const float* x, const float* y, size_t y_numel, float scalar, float* out, size_t total);
Status launch_affine(hipStream_t s, const float* x, float a, float b, float* out, size_t total);
Status launch_band_resample(hipStream_t s, const float* in, const float* w, const int* idx, int taps, int P, int L_in,
                            int L_out, int inner, float pa, float pb, float* out);
Status launch_ibp_update(hipStream_t s, float* x0, const float* y, const float* d, float gamma, float rho, int sf, int P, int H, int W,
                         const StepDev* sp = nullptr, const LoopDev* lp = nullptr);
Status launch_bicubic_up(hipStream_t s, const float* in, float* out, int P, int h, int w, int sf);
Status launch_randn(hipStream_t s, float* out, uint64_t seed, uint64_t stream_id, int64_t image_offset, int B, size_t per_image,
                    const StepDev* sp = nullptr, const LoopDev* lp = nullptr);   // sp: stream_id += 4 * sp->i; lp: seed / image_offset from the device block
// degrade.hip: degradation synthesis on the uint8 ground truth and per-image metric sums
Status launch_blur_wrap_u8(hipStream_t s, const uint8_t* gt, const float* k, int kh, int kw, int B, int H, int W, float* out);
Status launch_u8_to_single(hipStream_t s, const uint8_t* gt, const uint8_t* mask, int B, int HW, float* out);
Status launch_degrade_finish(hipStream_t s, float* y, const float* noise, double sigma2, const uint8_t* mask, const uint8_t* gt, int HW, size_t total);
Status launch_metrics(hipStream_t s, const float* x0, const uint8_t* gt, int B, int HW, double2* out);
void resizer_band(int in_len, int out_len, double scale, std::vector<float>& w_out, std::vector<int>& idx_out, int& taps_out);

// fft.hip ------------------------------------------------------------------------------------
// Twiddle table for size N: tw[k] = exp(-2 pi i k / N), k < N/2 (device pointer, float2)
struct FftPlan {
    int N = 0, logN = 0;
    float2* tw = nullptr;
};
// 2-D complex FFT building blocks operate on [P, H, W] complex64 planes (full c2c)
// rows: in-place FFT along W of every row; `real_in` (optional) supplies real input x*pa+pb instead of `buf`
Status launch_fft_rows(hipStream_t s, const FftPlan& pw, float2* buf, const float* real_in, float pa, float pb,
                       int P, int H, int W, bool inverse);
Status launch_fft_rows_real3(hipStream_t s, const FftPlan& pw, float2* buf, const float* real_in, float pa, float pb, float pm,
                             int P, int H, int W, const StepDev* sp = nullptr);   // sp: pm = sp->tau
// columns forward only (used by pre_calculate)
Status launch_fft_cols(hipStream_t s, const FftPlan& ph, float2* buf, int P, int H, int W, bool inverse);
// fused column pass of data_solution: col-FFT -> closed-form spectral solve -> inverse col-FFT
//   FR = FBFy + F(alpha*x);  FX = (FR - conj(FB) * tile(mean_alias(FB*FR)/(mean_alias(F2B)+alpha))) / alpha
struct SolveArgs {
    const float2* FB;    // [B,1,H,W]
    const float* F2B;    // [B,1,H,W]
    const float2* FBFy;  // [B,3,H,W]
    float alpha; int sf;
    const StepDev* sp;   // non-null: alpha = sp->tau
    // fft2.hip, sf > 1: mean of F2B over the aliases [B][H/sf][W/sf/2+1] and the slot map of the permuted half-spectrum layout
    const float* invW = nullptr; const int* slot_col = nullptr;
    int images = 0;      // fft4.hip: set by the launcher (planes / 3)
};
Status launch_fft_cols_solve(hipStream_t s, const FftPlan& ph, float2* buf, const SolveArgs& a, int B, int H, int W);
// inverse rows with real output: out = Re(ifft_row)*oa + ob, optionally blended: out = base + g*(val - base)
Status launch_ifft_rows_real(hipStream_t s, const FftPlan& pw, const float2* buf, float* out, float scale, float oa, float ob,
                             const float* blend_base, float g, int P, int H, int W);
// pointwise spectrum helpers for pre_calculate
Status launch_psf_embed(hipStream_t s, const float* k, int kh, int kw, float2* otf, int B, int H, int W);
Status launch_upsample_embed(hipStream_t s, const float* y, int sf, float2* out, int P, int h, int w);
Status launch_precalc_finish(hipStream_t s, const float2* FB, float2* FBFy_inout, float* F2B, int B, int H, int W);

// fft2.hip: half-spectrum register FFT path (sf = 1, N = 64 / 256).  twN = W_N^m table (N entries, device)
bool fft2_supported(int H, int W, int sf);
int fft2_padded_width(int W);
// eps6 != null (loop only): the row source is x0 = clamp(c1 x - c2 eps) computed on the fly from x and the UNet output
Status launch_rfft_rows(hipStream_t s, const float2* twN, const float* x, float pa, float pb, float pm, const StepDev* sp,
                        float2* out, int P, int N, const float* eps6 = nullptr, int out_ch = 0, const int* slot_col = nullptr);
// fused re-noise epilogue of the inverse row pass (loop only): x_t <- renoise(x_t, x0'), noise host-fed (n2 [, n1] + step stride,
// pointers re-read from lp when given) or Philox (n2 == null; seed / image offset from lp)
struct RenoiseArgs { float* xt; const StepDev* sp; const LoopDev* lp; const float* n1; const float* n2; size_t stride; int with_n1; };
Status launch_irfft_rows(hipStream_t s, const float2* twN, const float2* in, float* out, float scale, float oa, float ob,
                         const float* blend, float g, int P, int N, const RenoiseArgs* ra = nullptr, const int* col_slot = nullptr);
// sf > 1 on the half-spectrum path: alias-grouped column permutation (host tables), alias mean of F2B, zero-stuffed real up-sampling
void fft2_build_map(int N, int sf, std::vector<int>& slot_col, std::vector<int>& col_slot);
Status launch_fold_f2b(hipStream_t s, const float* F2B, const int* slot_col, int N, int sf, float* invW, int B);
Status launch_upsample_real(hipStream_t s, const float* y, int sf, float* out, int P, int h, int w);
Status launch_cfft_cols(hipStream_t s, const float2* twN, float2* buf, const SolveArgs& a, bool solve, int P, int N);
Status launch_precalc_finish2(hipStream_t s, const float2* FB, float2* FBFy, float* F2B, int B, size_t hw);
// fft4.hip: one wave per N-point transform, column-major half spectrum [plane][slot][row] (NC slots per plane); N x N = 256 x 256 or 512 x 512, sf 1 / 2 / 4
bool fft4_supported(int H, int W, int sf);
int fft4_columns(int W, int sf);
int fft4_row_pos(int u);      // position of row u inside a stored column
void fft4_build_map(int N, int sf, std::vector<int>& slot_col, std::vector<int>& col_slot);
Status launch_rfft4_rows(hipStream_t s, const float2* tw, int N, const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out, int P, int NC,
                         const float* eps6, int out_ch, const int* slot_col);
Status launch_irfft4_rows(hipStream_t s, const float2* tw, int N, const float2* in, float* out, float scale, float oa, float ob, const float* blend, float g,
                          int P, int NC, const RenoiseArgs* ra, const int* col_slot);
Status launch_cfft4_cols(hipStream_t s, const float2* tw, int N, float2* buf, const SolveArgs& a, bool solve, int P, int NC);
Status launch_fold_f2b4(hipStream_t s, const float* F2B, const int* slot_col, int N, int NC, int sf, float* invW, int B);
// the arguments of the three half-spectrum passes (rows forward -> columns with the solve -> rows inverse), whichever kernels run them
struct RowsFuse { const float* eps6; int out_ch; };      // eps -> x0 prologue of the row pass (loop only)
struct RenoiseFuse { float* xt; const StepDev* sp; const LoopDev* lp; const float* n1; const float* n2; size_t stride; int with_n1; };
struct ProxPassArgs {
    const float* x; float pa, pb, pm; const StepDev* sp; RowsFuse fu; const int* slot_col;                       // rows forward
    SolveArgs solve;                                                                                             // columns
    float* out; float scale, oa, ob; const float* blend_base; float g; RenoiseFuse rn; const int* col_slot;     // rows inverse
    float2* hbuf; const float2* tw;
};

Status launch_psf_embed_real(hipStream_t s, const float* k, int kh, int kw, float* out, int B, int H, int W);

}  // namespace dpir
