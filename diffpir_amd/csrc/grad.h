// Declarations for grad.hip / unet_bwd.hip: the input-gradient path (DPS modes).
#pragma once
#include "common.h"
#include "elem.h"

namespace dpir {

// GroupNorm (+FiLM) + SiLU backward for one normalisation site.  x: the (virtual concat) source at Hs x Ws; prm / stats: the
// forward's per-(image, channel) {mean, a, b, act} table and per-(image, group) {mean, rstd}; dA: gradient w.r.t. the activated,
// resampled convolution input [B, C, Ho, Wo] (mode as in act.hip); sums: scratch [B * 32 * gn_bwd_parts()] double2.
// Output: gradient w.r.t. x, written (acc = 0) or accumulated into ga (channels < ca) and gb (the rest).
struct GnBwdArgs {
    CatSrc x; const float4* prm = nullptr; const float2* stats = nullptr;
    const float* dA = nullptr; int mode = 0; int Hs = 0, Ws = 0;
    double2* sums = nullptr;
    float* ga = nullptr; float* gb = nullptr; int acc_a = 0, acc_b = 0;
    // optional: a tensor shaped like x.a (mode 0, no concat) added into ga BEFORE dx -- the identity skip's gradient (unet.py:256), which used to be its own
    // accum_adj pass over ga: ga = ((ga_old | 0) + extra) + dx, the same operations in the same order
    const float* extra = nullptr;
};
int gn_bwd_parts(int C, int Hs, int Ws);
Status launch_gn_bwd(hipStream_t s, const GnBwdArgs& a, int B);
// scal[0] = s (power of two, max|x| * s in [512, 1024)), scal[1] = 1 / s, prm[0 .. n_prm) = {0, s, 0, 0}; part: scratch of 512 floats
Status launch_grad_scale(hipStream_t s, const float* x, size_t total, float* part, float* scal, float4* prm, int n_prm);
Status launch_accum_adj(hipStream_t s, const float* src, int Cs, int c0, float* dst, int Cd, int mode, int B, int Hs, int Ws, bool acc);
Status launch_attention_bwd(hipStream_t s, const float* qkv, const float* dAtt, float* dqkv, float* P, float* dP, int B, int C, int T);

// p_sample coefficients of the current timestep (float64 tables of GaussianDiffusion cast to float32 as _extract_into_tensor does)
// ddim = 1: ddim_sample(eta = 0) (gaussian_diffusion.py:537-585): x_prev = x0 * sa_prev + s1m_prev * (c1 x - x0) / c2, the noise draw is consumed
// but multiplied by sigma = 0
struct PSampleCoef { float c1, c2, pc1, pc2, min_log, max_log, nonzero; int ddim = 0; float sa_prev = 0.f, s1m_prev = 0.f; };
// utils_model.model_fn 'epsilon' / 'score' (utils_model.py:247-255): (x - sa x0) / s1m  [* -1 / s1m]
Status launch_eps_from_xstart(hipStream_t s, const float* x, const float* x0, float sa, float s1m, int score, float* out, size_t total);
// out[0] = sqrt(sum(part[0..n)))   |   ssq[0] = sum(part[0..n))   |   out[0] = sqrt(ssq[0])    (the batch-wide residual norm, folded in a fixed order)
Status launch_norm_fold_ssq(hipStream_t s, const double* part, int n, double* ssq);
Status launch_norm_sqrt(hipStream_t s, const double* ssq, float* norm_out);
Status launch_psample(hipStream_t s, const float* x, const float* out6, int out_ch, const float* noise, const PSampleCoef& cf, float* x0,
                      float* xprev, unsigned char* inside, int B, int HW);
// norm_out == nullptr: only the partial sums are produced (the caller folds / all-reduces them: launch_norm_fold_ssq, launch_norm_sqrt)
Status launch_diff_norm(hipStream_t s, const float* y, float ma, float mb, const float* down, float* diff, size_t total, double* part, int nparts,
                        float* norm_out, float sa = 1.f, float s1m = 0.f, const float* noise = nullptr, const LoopDev* lp = nullptr);
Status launch_grad_step(hipStream_t s, const float* src, const float* gup, const float* norm, float lam, float rho, float tail, float* dst, size_t total,
                        const StepDev* sp = nullptr);
// out = -gup / norm  (d || m - A(x) || / d x for a linear operator A, gup = A^T (m - A(x)))
Status launch_neg_scale_by_norm(hipStream_t s, const float* gup, const float* norm, float* out, size_t total);
Status launch_band_resample_T(hipStream_t s, const float* gout, const float* w, const int* idx, int taps, int P, int L_in, int L_out, int inner,
                              float scale, float* gin);
Status launch_dps_seed(hipStream_t s, const float* gup, const float* norm, const unsigned char* inside, float c1, float c2, int out_ch, float* dout6,
                       float* direct, int B, int HW);
Status launch_dps_update(hipStream_t s, const float* xprev, const float* direct, const float* dx_net, float step_scale, float* x, float* grad_out,
                         size_t total);

}  // namespace dpir
