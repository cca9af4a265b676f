/*
 * diffpir_engine.h -- C ABI of the MI355X-native DiffPIR sampling engine (libdiffpir_hip.so).
 *
 * The reference (yuanzhi-zhu/DiffPIR) has no FFI: its "plugin surface" is three Python call
 * signatures used by main_ddpir.py's restoration loop.  Each entry point
 * below names the reference interface it replaces (paths relative to the reference root).
 * The Python shims in diffpir_amd/ (utils_model.model_fn, utils_sisr.pre_calculate /
 * data_solution, ...) mirror those signatures and call straight into these symbols via ctypes;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (DPIR_OK) or a negative dpir_status; nothing throws or aborts
 *     across the ABI; dpir_last_error() gives the message for the last failure on that engine.
 *   - "dev" pointers are raw device addresses (engine-allocated with dpir_malloc, or any other
 *     HIP allocation such as torch.Tensor.data_ptr()); "host" pointers are plain host memory.
 *   - tensors are fp32, NCHW, contiguous; images are in [-1,1] inside the loop and y in [0,1],
 *     exactly as in main_ddpir.py.  Masks are uint8 {0,1} (bit-exact integer semantics).
 *   - all work is enqueued on ONE engine-owned HIP stream; calls are asynchronous unless noted
 *     (dpir_sync, D2H copies).  One engine per device; an engine is not thread-safe, independent
 *     engines are.
 */
#ifndef DIFFPIR_ENGINE_H
#define DIFFPIR_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPIR_ABI_VERSION 2   /* 2 (round 5): dpir_dps_coef grew to 24 bytes and dpir_loop_desc gained ddim_sample in round 4 without a bump; bumped with every public struct layout change from here on */

typedef enum dpir_status {
    DPIR_OK = 0,
    DPIR_ERR_INVALID = -1,   /* bad argument / shape / missing weight */
    DPIR_ERR_HIP = -2,       /* a HIP runtime call failed */
    DPIR_ERR_NOMEM = -3,
    DPIR_ERR_STATE = -4,     /* e.g. forward before load_unet */
    DPIR_ERR_UNSUPPORTED = -5,
    DPIR_ERR_RANGE = -6      /* f16x3 mode: an activation left the f16 operand range (|v| > 65000) and was clamped; the
                              * result is wrong.  Reported by dpir_sync / dpir_d2h after the offending work; reload the
                              * weights with dpir_set_precision(e, 0).  Mirrors the reference's use_fp16 caveat
                              * (guided_diffusion/unet.py:618-632, fp16_util.py:15-32) -- but loudly. */
} dpir_status;

typedef struct dpir_engine dpir_engine;   /* opaque */
typedef struct dpir_prox dpir_prox;       /* opaque: FB / F2B / FBFy spectra of one batch */

/* ---- lifecycle -------------------------------------------------------------------------- */
int dpir_version(void);
/* number of HIP devices visible to this process (0 when there is none; never fails).  The reference reads
 * torch.cuda.device_count() for its unused world_size (main_ddpir.py:135); the multi-GPU launcher and tests use this. */
int dpir_device_count(int* n_out);
/* "<gcnArchName> | <device name> | pci <bus id> | <CUs> CUs | ordinal <n>" of the engine's device into buf (NUL-terminated, truncated to cap) */
int dpir_device_info(dpir_engine* e, char* buf, size_t cap);
/* device = HIP ordinal.  Fails with DPIR_ERR_HIP when no gfx950 device is visible. */
int dpir_create(int device, dpir_engine** out);
void dpir_destroy(dpir_engine* e);
const char* dpir_last_error(const dpir_engine* e);   /* never NULL */
/* hipStreamSynchronize on the engine stream, then the f16 operand range guard (DPIR_ERR_RANGE, above).  Also the point where a time-out of
 * conv7's fused GroupNorm hop (an inter-workgroup wait; only seen when 3+ engines / processes share the GPU) is handled: the engine switches
 * the hop off for its lifetime and, when the ONE eager forward (dpir_unet_forward / dpir_model_fn_xstart / dpir_p_sample) outstanding since the
 * last synchronisation is still the LAST work on the stream, its outputs do not alias its inputs and no dpir_free happened since, re-issues it on the
 * unfused path and returns DPIR_OK with correct results.  Anything else -- several un-synchronised forwards, or other calls (a prox step, dpir_finalize,
 * a copy) already queued behind the forward, which have consumed its invalid output -- cannot be repaired from here and returns DPIR_ERR_HIP ONCE
 * (not sticky: repeat the calls).  dpir_d2h does the same before it copies; dpir_run_loop re-runs itself. */
int dpir_sync(dpir_engine* e);
/* the engine's hipStream_t, for callers that enqueue their own work (e.g. RCCL) behind it */
void* dpir_stream(dpir_engine* e);

/* ---- device memory (plumbing) ------------------------------------------------------------ */
int dpir_malloc(dpir_engine* e, size_t bytes, void** dev_out);
int dpir_free(dpir_engine* e, void* dev);
int dpir_h2d(dpir_engine* e, void* dev_dst, const void* host_src, size_t bytes);   /* async */
int dpir_d2h(dpir_engine* e, void* host_dst, const void* dev_src, size_t bytes);   /* syncs  */
int dpir_d2d(dpir_engine* e, void* dev_dst, const void* dev_src, size_t bytes);    /* async  */

/* ---- denoiser: guided-diffusion UNet ----------------------------------------------------- */
/* Replaces script_util.create_model(...) hyper-parameters (guided_diffusion/script_util.py:130-184
 * as resolved by utils/utils_model.py:353-387 and main_ddpir.py:219-230). */
typedef struct dpir_unet_desc {
    int32_t image_size;          /* 256 / 512: only selects the default channel_mult */
    int32_t in_channels;         /* 3 */
    int32_t model_channels;      /* 128 (FFHQ) / 256 (ImageNet) */
    int32_t out_channels;        /* 6 (learn_sigma) */
    int32_t num_res_blocks;
    int32_t num_head_channels;   /* 64 */
    int32_t n_channel_mult;      /* 0 -> default for image_size (script_util.py:147-160) */
    float channel_mult[8];
    int32_t n_attention_ds;      /* downsample rates with attention, e.g. {16} */
    int32_t attention_ds[8];
    int32_t num_classes;         /* 0 = unconditional, else label_emb rows (class_cond) */
} dpir_unet_desc;

/* One entry of the reference state-dict (main_ddpir.py:234 torch.load -> load_state_dict). */
typedef struct dpir_tensor {
    const char* name;            /* reference key, e.g. "input_blocks.3.0.in_layers.2.weight" */
    const float* data;           /* HOST pointer, fp32, contiguous, reference layout (OIHW ...) */
    int32_t ndim;
    int64_t shape[4];
} dpir_tensor;

/* Replaces model.load_state_dict(...).to(device) (main_ddpir.py:231-240).  Copies and repacks the
 * weights into the engine's arena (caller keeps ownership of the host arrays).  Fails with
 * DPIR_ERR_INVALID naming the first missing / mis-shaped key. */
int dpir_load_unet(dpir_engine* e, const dpir_unet_desc* desc, const dpir_tensor* weights, int n_weights);

/* Arithmetic of the convolution / attention GEMMs, to be chosen BEFORE dpir_load_unet (weights are packed for it):
 *   0  exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the default;
 *   1  operand-split f16 MFMA: x = hi + lo in f16, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulation --
 *      22-bit products, measured error <= the fp32 MFMA chain's, 5.3x its rate;
 *   2  f16x1: f16 operands (weights and activations rounded once to f16), ONE MFMA per product, fp32 accumulation, fp32
 *      GroupNorm / softmax / residual stream -- the reference's own reduced-precision recipe (guided_diffusion/fp16_util.py:15-32,
 *      unet.py:618-632 `use_fp16`).  NOT within the 1e-3 dB parity contract: its measured quality delta is reported by bench.py. */
int dpir_set_precision(dpir_engine* e, int mode);

/* Replaces UNetModel.forward (guided_diffusion/unet.py:634-663): x_dev [B,3,H,W], t_host [B] int64
 * timesteps (host), y_host [B] int64 labels or NULL -> out_dev [B,out_channels,H,W].  Asynchronous on the engine stream.  When all B timesteps
 * are equal and there are no labels (what utils_model.model_fn always passes: `[t_step] * x.shape[0]`) the timestep is written on the device by a
 * kernel (no host copy, no synchronisation) and the time embedding + FiLM projection are evaluated for one row and shared; otherwise t / y are
 * uploaded (one stream synchronisation). */
int dpir_unet_forward(dpir_engine* e, const float* x_dev, const int64_t* t_host, const int64_t* y_host,
                      float* out_dev, int B, int H, int W);

/* Replaces utils_model.model_fn(..., model_out_type='pred_xstart') (utils/utils_model.py:207-258)
 * = SpacedDiffusion.p_sample -> p_mean_variance (guided_diffusion/gaussian_diffusion.py:232-333):
 * x0 = clamp(c1*x - c2*eps, -1, 1), eps = UNet(x,t)[:, :3];  c1 = float32(sqrt(1/acp64[t])),
 * c2 = float32(sqrt(1/acp64[t]-1)) computed by the caller from the float64 schedule. */
int dpir_model_fn_xstart(dpir_engine* e, const float* x_dev, int t, float c1, float c2,
                         const int64_t* y_host, float* x0_dev, int B, int H, int W);

/* Optional per-layer tap for parity tests: copies the named layer's most recent output (the
 * reference module path, e.g. "input_blocks.3.0") to host.  numel_out receives the element count. */
int dpir_unet_read_tap(dpir_engine* e, const char* layer, float* host_dst, size_t cap_floats, size_t* numel_out);

/* ---- data-fidelity operators ------------------------------------------------------------- */
/* Replaces sr.pre_calculate(y, k, sf) (utils/utils_sisr.py:78-95): y_dev [B,3,H/sf,W/sf] in [0,1],
 * k_dev [B,1,kh,kw]; builds FB, F2B, FBFy (full c2c spectra of size HxW) in an engine-owned object. */
int dpir_prox_fft_precalc(dpir_engine* e, const float* y_dev, const float* k_dev, int kh, int kw,
                          int sf, int B, int H, int W, dpir_prox** out);
void dpir_prox_free(dpir_engine* e, dpir_prox* p);
/* Copy spectra to host for cross-checks: which = 0 FB (complex64 [B,1,H,W]), 1 F2B (f32 [B,1,H,W]),
 * 2 FBFy (complex64 [B,3,H,W]). */
int dpir_prox_read(dpir_engine* e, const dpir_prox* p, int which, void* host_dst, size_t cap_bytes);
/* Replaces sr.data_solution(x, FB, FBC, F2B, FBFy, alpha, sf) (utils/utils_sisr.py:65-75):
 * x_dev [B,3,H,W] in [0,1] -> out_dev (may alias x_dev). */
int dpir_data_solution(dpir_engine* e, const dpir_prox* p, const float* x_dev, float alpha, float* out_dev);
/* Replaces main_ddpir.py:395-400: x0 <- x0 + g*(2*data_solution(x0/2+.5, tau) - 1 - x0), in place. */
int dpir_prox_fft_apply(dpir_engine* e, const dpir_prox* p, float* x0_dev, float tau, float guidance);
/* Measurement helper (bench.py `roofline_prox`): dpir_prox_fft_apply n times back to back between two HIP events on the engine stream -- eager launches
 * (use_graph 0) or ONE captured graph of the n applies (use_graph 1: how dpir_run_loop replays the step; no host launch cost, no event record between
 * applies) -> device microseconds per apply, launch boundaries included.  x0 is overwritten n + 1 (+ n) times. */
int dpir_prox_fft_apply_timed(dpir_engine* e, const dpir_prox* p, float* x0_dev, float tau, float guidance, int n, int use_graph, float* us_per_apply);
/* Which kernels run the half-spectrum data_solution (utils/utils_sisr.py:65-75).  mode 1 (default; env DPIR_PROX_MODE=0 selects 0 at dpir_create), 256 x 256
 * and 512 x 512: one wave per 256- / 512-point transform (64 lanes x 4 / 8 points, permlane-swap / DPP register transposes, a wave-private LDS tile, no workgroup
 * barrier inside a transform) on a COLUMN-major half spectrum, csrc/fft4.hip.  mode 0, and every other size: the two-pass register kernels of csrc/fft2.hip (a thread holds
 * 16 points) on the row-major padded spectrum.  Same mathematics, results within a few 1e-6 of each other.  A dpir_prox keeps the layout of the mode it
 * was created in (dpir_prox_read returns natural order either way).  Drops the captured step graphs. */
int dpir_set_prox_launch(dpir_engine* e, int mode);
/* Replaces main_ddpir.py:392-394: x0_p = (m*(2y-1)+tau*x0)/(m+tau); x0 += g*(x0_p-x0).  mask u8 [B,3,H,W]. */
int dpir_prox_mask(dpir_engine* e, float* x0_dev, const float* y_dev, const uint8_t* mask_dev,
                   float tau, float guidance, int B, int H, int W);
/* Replaces Resizer(in_shape, 1/sf).forward (utils/utils_resizer.py:55-74, cubic, antialiasing):
 * x_dev [B,3,H,W] -> out_dev [B,3,H/sf,W/sf]. */
int dpir_resize_down(dpir_engine* e, const float* x_dev, float* out_dev, int sf, int B, int H, int W);
/* Replaces main_ddpir.py:401-406 (sr_mode 'cubic', iterative back-projection), in place on x0:
 * in_iter times { x0 <- 2*(z + gamma*up_nearest(y - down(z))/(1+rho)) - 1, z = x0/2+.5 }. */
int dpir_prox_ibp(dpir_engine* e, float* x0_dev, const float* y_dev, float rho, float gamma,
                  int in_iter, int sf, int B, int H, int W);
/* Replaces F.interpolate(y, size=(h*sf,w*sf), mode='bicubic', align_corners=False) (main_ddpir.py:295). */
int dpir_bicubic_up(dpir_engine* e, const float* y_dev, float* out_dev, int sf, int B, int h, int w);

/* ---- loop arithmetic --------------------------------------------------------------------- */
/* Per-step scalars (SURVEY.md 8 a-S), all computed on the host exactly as the reference does. */
typedef struct dpir_step {
    int32_t t;            /* t_i: UNet timestep */
    int32_t last;         /* 1 on the final step: no prox / re-noise (main_ddpir.py:384,448) */
    float c1, c2;         /* eps -> x0 (gaussian_diffusion.py:328-333), float64 tables cast to f32 */
    float tau;            /* rhos[t_i] (main_ddpir.py:389) */
    float sa_t, s1m_t;    /* sqrt_alphas_cumprod[t_i], sqrt_1m_alphas_cumprod[t_i]  (float32 tables) */
    float sa_p;           /* sqrt_alphas_cumprod[t_im1] */
    float k1;             /* float32(np.sqrt(1-zeta)) */
    float q;              /* sqrt(s1m_p^2 - eta_sigma^2), float32 arithmetic as in the reference */
    float es;             /* eta_sigma = eta * s1m_p / s1m_t * sqrt(betas[t_i])  (0 when eta == 0) */
    float k2;             /* np.sqrt(zeta) * s1m_p */
} dpir_step;

/* Replaces main_ddpir.py:451-456 (re-noise to t_{i-1}); x_dev updated in place.
 * x = sa_p*x0 + k1*(q*eps + es*n1) + k2*n2, eps = (x - sa_t*x0)/s1m_t.  n1_dev may be NULL when es == 0. */
int dpir_renoise(dpir_engine* e, float* x_dev, const float* x0_dev, const dpir_step* s,
                 const float* n1_dev, const float* n2_dev, int B, int H, int W);
/* Replaces main_ddpir.py:470,482 + utils_image.tensor2uint_batch (utils/utils_image.py:238-242):
 * x_dev [B,3,H,W] in [-1,1] -> out_f32 [B,3,H,W] = x/2+.5 (optional) and out_u8 [B,H,W,3] (optional). */
/* main_ddpir.py:355-358: x = (sa_t*(2y-1) + s1m_t*n)*mask + (1-mask)*x  (coefficients from the step: sa_t, s1m_t) */
int dpir_repaint_mix(dpir_engine* e, float* x_dev, const float* y_dev, const uint8_t* mask_dev, const dpir_step* s,
                     const float* n_dev, int B, int H, int W);
int dpir_finalize(dpir_engine* e, const float* x_dev, float* out_f32_dev, uint8_t* out_u8_dev, int B, int H, int W);
/* N(0,1) on device (Philox4x32-10 + Box-Muller) keyed by (seed, image index offset, stream id):
 * the perf-mode replacement of torch.randn_like; parity mode feeds host noise. */
int dpir_randn(dpir_engine* e, float* out_dev, uint64_t seed, uint64_t stream_id, int64_t image_offset,
               int B, int C, int H, int W);

/* The loop body's own tensor arithmetic, for a loop that stays in the host language: main_ddpir.py:437 `x = xt - norm_grad * 1.`,
 * :440 `sa_t * (2*y-1) + s1m_t * randn_like(y)`, :444 `xt - norm_grad * lambda_ * norm / rhos[t_i] * 0.35`.  One float32 operation per
 * element, rounded once, like a torch elementwise op: out[i] = x[i] op rhs with rhs = y_dev[i] (y_numel == numel), y_dev[0]
 * (y_numel == 1: a 0-dim tensor such as `norm`) or `scalar` (y_dev NULL).  op: 0 add, 1 sub, 2 mul, 3 div, 4 rhs - x, 5 rhs / x.
 * out_dev may alias x_dev. */
int dpir_ewise(dpir_engine* e, int op, const float* x_dev, const float* y_dev, size_t y_numel, float scalar, float* out_dev, size_t numel);

/* ---- whole restoration loop (main_ddpir.py:291-470 for one batch) ------------------------- */
typedef enum dpir_task { DPIR_TASK_DEBLUR = 0, DPIR_TASK_SR_BLUR = 1, DPIR_TASK_INPAINT = 2, DPIR_TASK_SR_CUBIC = 3 } dpir_task;

typedef struct dpir_loop_desc {
    int32_t task;                 /* dpir_task */
    int32_t B, H, W;              /* H,W = restored (high-res) size */
    int32_t sf;                   /* 1 for deblur / inpaint */
    int32_t kh, kw;               /* PSF size (deblur / sr-blur) */
    int32_t in_iter;              /* sr-cubic */
    float gamma;                  /* sr-cubic */
    float guidance;               /* guidance_scale */
    float sa_start, s1m_start;    /* forward-noise coefficients at t_start (main_ddpir.py:315) */
    const float* y_dev;           /* [B,3,H/sf,W/sf] in [0,1] */
    const float* k_dev;           /* [B,1,kh,kw] or NULL */
    const uint8_t* mask_dev;      /* [B,3,H,W] or NULL */
    const int64_t* labels_host;   /* [B] or NULL (class-conditional UNet) */
    /* host-fed noise (parity mode): init [B,3,H,W]; n1/n2 [(n_steps-1),B,3,H,W]; any may be NULL ->
     * device Philox keyed by (seed, image_offset + b, draw index) */
    const float* noise_init_dev;
    const float* noise_n1_dev;
    const float* noise_n2_dev;
    uint64_t seed;
    int64_t image_offset;         /* global index of image 0 of this shard (multi-GPU invariance) */
    int32_t use_graph;            /* 1: capture one step as a hipGraph and replay it n_steps times */
    int32_t skip_dead_final_eval; /* 1: skip the last UNet call whose output is discarded (Q2) */
    /* generate_mode (main_ddpir.py:349-358, 384): 0 DiffPIR (prox every step), 1 repaint (inpainting only: the known region
     * is re-drawn at the current noise level before every denoiser call, no prox), 2 vanilla (inpainting only: no
     * conditioning inside the loop).  Re-noising is applied in all three (main_ddpir.py:448). */
    int32_t generate_mode;
    /* 1: sub_1_analytic = false -- the first-order data step of main_ddpir.py:420-430 instead of the closed-form prox:
     * x0 <- x0 - d||(2y-1) - Resizer(x0)|| / dx0 * ||.|| / rho (super-resolution tasks, DiffPIR mode; no network backward) */
    int32_t first_order;
    const float* noise_rp_dev;    /* repaint, host-fed noise: [n_steps,B,3,H,W] in step order; NULL -> device Philox (draw 3) */
    int32_t ddim_sample;          /* dpir_run_dps_loop: 1 -> x_prev from ddim_sample(eta=0) instead of p_sample (config.ddim_sample,
                                   * utils_model.py:219-240).  dpir_run_loop ignores it: 'pred_xstart' is the same tensor either way. */
} dpir_loop_desc;

/* Runs init -> n_steps x ([repaint mix ->] UNet -> [prox ->] re-noise) -> finalize.  Outputs (either may be NULL):
 * out_f32_dev [B,3,H,W] in [0,1] un-clamped (x_0 of main_ddpir.py:470), out_u8_dev [B,H,W,3].
 * Synchronisation: in the f32 mode, in gradient mode and once the fused hop is off the call returns as soon as the last step is enqueued
 * (asynchronous on the engine stream).  In the f16 modes with the fused hop on it BLOCKS until the loop has finished: it reads the guard word back
 * (one 8-byte D2H + hipStreamSynchronize) so that a hop time-out can re-run the whole loop on the unfused path before the caller sees a result --
 * callers that queue the next batch's uploads behind the loop lose that overlap; DPIR_FUSE_H1=0 trades ~2 % of the step for an asynchronous return. */
int dpir_run_loop(dpir_engine* e, const dpir_loop_desc* d, const dpir_step* steps_host, int n_steps,
                  float* out_f32_dev, uint8_t* out_u8_dev);

/* ---- gradient-based sampling (generate_mode 'DPS_y0') ---------------------- */
/* Replaces what torch.autograd does for the reference's DPS branch (main_ddpir.py:370-373, 434-438 with
 * utils_model.grad_and_value, utils/utils_model.py:390-394).  Gradient mode must be switched on BEFORE dpir_load_unet: it builds
 * the dgrad weight packs and makes every forward record what the backward needs (the fused elementwise prologues are off). */
int dpir_enable_grad(dpir_engine* e, int on);
/* Vector-Jacobian product of UNetModel.forward w.r.t. its input: runs the forward (out_dev [B,out_channels,H,W], may be NULL) and
 * returns dx_dev [B,3,H,W] = J(x)^T gout_dev, gout_dev [B,out_channels,H,W].  The unit the parity tests compare with
 * torch.autograd.grad on the reference network; per-layer gradients are readable as taps named "grad:<layer>". */
int dpir_unet_vjp(dpir_engine* e, const float* x_dev, const int64_t* t_host, const int64_t* y_host, const float* gout_dev,
                  float* out_dev, float* dx_dev, int B, int H, int W);
/* Per-step p_sample coefficients (gaussian_diffusion.py:153-167, 268-276: float64 tables cast to float32 by
 * _extract_into_tensor): posterior_mean_coef1/2[t], posterior_log_variance_clipped[t], log(betas[t]). */
typedef struct dpir_dps_coef {
    float pc1, pc2, min_log, max_log;
    float sa_prev, s1m_prev;      /* ddim_sample(eta=0) only: sqrt(alphas_cumprod_prev[t]), sqrt(1 - alphas_cumprod_prev[t]) in float32
                                   * (gaussian_diffusion.py:568-580) */
} dpir_dps_coef;
/* generate_mode 'DPS_y0' for the super-resolution tasks (the only DPS variant the reference can run as shipped: its deblurring
 * operator raises at main_ddpir.py:302 and the inpainting branch never defines xt).  Per step:
 *   xt, x0 = p_sample(x)                       (model_fn 'pred_x_prev_and_start', utils_model.py:207-258)
 *   norm   = || (2y - 1) - Resizer(x0) ||_2    over the whole batch          (grad_and_value)
 *   x      = xt - step_scale * d norm / d x    (main_ddpir.py:437, step_scale = 1), no re-noising (:448)
 * variant 1 = 'DPS_yt' (main_ddpir.py:439-445): y_t = sa_t (2y-1) + s1m_t n;  norm = || y_t - Resizer(xt) ||_2;
 *   x = xt - d norm / d xt * lambda * norm / rho_t * 0.35   -- differentiated w.r.t. xt itself, no network backward (gradient mode not needed).
 * d / steps_host as in dpir_run_loop (task DPIR_TASK_SR_BLUR or _SR_CUBIC; k / mask / n1 / n2 unused); coefs_host [n_steps];
 * noise_ps_dev: host-fed p_sample noise [n_steps, B,3,H,W] in step order or NULL -> device Philox (stream 4*(step+1));
 * noise_yt_dev (variant 1): host-fed y_t noise [n_steps, B,3,H/sf,W/sf] or NULL -> Philox (stream 4*(step+1)+1). */
int dpir_run_dps_loop(dpir_engine* e, const dpir_loop_desc* d, const dpir_step* steps_host, const dpir_dps_coef* coefs_host, int n_steps,
                      int variant, float lambda_, const float* noise_ps_dev, const float* noise_yt_dev, float step_scale,
                      float* out_f32_dev, uint8_t* out_u8_dev);

/* The two plugs the reference's DPS / first-order branches call, for a loop body that stays in the host language (round 4):
 *
 * dpir_p_sample = utils_model.model_fn(x, ..., model_out_type='pred_x_prev_and_start' | 'pred_x_prev') (utils/utils_model.py:207-246,
 *   called at main_ddpir.py:370-373): one denoiser call, then GaussianDiffusion.p_sample with the learned-range variance
 *   (gaussian_diffusion.py:232-326, 395-439) or, with c->ddim, ddim_sample(eta = 0) (:537-585).  noise_dev [B,3,H,W] is the
 *   randn_like draw of the sampler (required; ddim consumes it with sigma = 0).  Outputs xt ("sample") and x0 ("pred_xstart").
 *   In gradient mode the call leaves the forward's tape and the clamp mask on the engine for dpir_grad_and_value.
 *   Needs a learn_sigma model (out_channels == 6), else DPIR_ERR_UNSUPPORTED. */
typedef struct dpir_psample_coef {
    float c1, c2;                 /* sqrt_recip_alphas_cumprod[t], sqrt_recipm1_alphas_cumprod[t] */
    float pc1, pc2, min_log, max_log;   /* as dpir_dps_coef */
    int32_t ddim;
    float sa_prev, s1m_prev;      /* as dpir_dps_coef */
} dpir_psample_coef;
int dpir_p_sample(dpir_engine* e, const float* x_dev, int t, const dpir_psample_coef* c, const float* noise_dev, const int64_t* y_host,
                  float* xt_out_dev, float* x0_out_dev, int B, int H, int W);
/* model_out_type 'epsilon' / 'score' (utils/utils_model.py:247-255): out = (x - sqrt_ac x0) / sqrt_1m_ac  [score: * -1 / sqrt_1m_ac]. */
int dpir_eps_from_xstart(dpir_engine* e, const float* x_dev, const float* x0_dev, float sqrt_ac, float sqrt_1m_ac, int score,
                         float* out_dev, size_t numel);
/* dpir_grad_and_value = utils_model.grad_and_value(operator=Resizer(1/sf), x, x_hat, measurement) (utils/utils_model.py:390-394):
 *   difference = measurement - Resizer(x_hat);  norm = ||difference||_2 over the WHOLE batch (all ranks of the communicator when one
 *   with more than one rank is attached);  norm_grad = d norm / d x.
 * through_network = 1: x is the input of the LAST dpir_p_sample on this engine and x_hat_dev its pred_xstart output (checked) -- the
 *   gradient runs through the clamp and the denoiser (main_ddpir.py:436, generate_mode DPS_y0); gradient mode required.
 * through_network = 0: x is x_hat itself (main_ddpir.py:425 first-order data step, :443 DPS_yt): norm_grad = -Resizer^T(difference) / norm.
 * measurement_dev [B,3,H/sf,W/sf] in the operator's range (the reference passes 2y-1 or y_t).  norm_grad_out_dev [B,3,H,W];
 * norm_out_dev: one device float (may be NULL).  Asynchronous. */
int dpir_grad_and_value(dpir_engine* e, int through_network, const float* x_hat_dev, const float* measurement_dev, int sf,
                        float* norm_grad_out_dev, float* norm_out_dev, int B, int H, int W);

/* ---- multi-GPU: the one collective of the path ------------------------------ */
/* One process and one engine per GPU; images are block-partitioned over ranks, no exchange inside the loop (the reference is
 * single-GPU: main_ddpir.py:135 sets world_size and never uses it).  After a batch, ONE all-gather of the uint8 results over
 * RCCL / xGMI, enqueued on the engine stream behind the loop.  librccl.so is bound at run time (dlopen).
 *   dpir_comm_unique_id : rank 0 creates the 128-byte ncclUniqueId; the host side ships it to the other ranks;
 *   dpir_comm_init      : ncclCommInitRank on this engine's device;
 *   dpir_allgather_results(send [bytes_per_rank], recv [world * bytes_per_rank]) : ncclAllGather(ncclUint8);
 *   dpir_comm_allreduce_max / dpir_comm_barrier : the two rendezvous primitives a multi-GPU driver needs around its timed region
 *                         (MAX over ranks of one host double; a barrier is the same exchange with the value ignored) -- with
 *                         them the N-GPU launch needs no other communication library;
 *   dpir_comm_destroy   : also done by dpir_destroy. */
int dpir_comm_unique_id(void* id128_out);
/* ncclGetVersion() of the bound librccl.so (diagnostics of the multi-GPU bench line); DPIR_ERR_UNSUPPORTED when librccl cannot be loaded */
int dpir_comm_version(int* version_out);
int dpir_comm_init(dpir_engine* e, int world, int rank, const void* id128);
int dpir_allgather_results(dpir_engine* e, const void* send_dev, void* recv_dev, size_t bytes_per_rank);
int dpir_comm_allreduce_max(dpir_engine* e, double* value_inout);
int dpir_comm_barrier(dpir_engine* e);
int dpir_comm_destroy(dpir_engine* e);

/* ---- degradation synthesis and metrics: the steps either side of the loop ------------------ */
/* Replaces CustomDataset.__getitem__'s arithmetic (main_ddpir.py:84-114) on the device.  gt_u8_dev: ground truth, uint8
 * [B,H,W,3] (what util.imread_uint returns).  deblur: scipy.ndimage.convolve(img_H, k, mode='wrap') on the uint8 image (float64
 * accumulation, result cast back to uint8) / 255; sr (both sr_modes): utils_image.imresize_np(img_H / 255, 1 / sf) (== the Resizer
 * weights); inpaint: img_H * mask / 255.  Then img_L*2-1 + N(0, (2*noise_level_img)^2), /2 + 0.5 in float64 (:112-114) and, for
 * inpainting, * mask (:311-313).  noise_dev: [B,3,h,w] standard-normal floats (host-fed) or NULL -> device Philox keyed by
 * (seed, image_offset + b).  y_out_dev: [B,3,H/sf,W/sf] float32 in [0,1]. */
typedef struct dpir_degrade_desc {
    int32_t task;                 /* dpir_task */
    int32_t B, H, W, sf;
    int32_t kh, kw;               /* deblur PSF size */
    float noise_level_img;        /* already / 255 (main_ddpir.py:138) */
    uint64_t seed;
    int64_t image_offset;
} dpir_degrade_desc;
int dpir_degrade(dpir_engine* e, const dpir_degrade_desc* d, const uint8_t* gt_u8_dev, const float* k_dev, const uint8_t* mask_dev,
                 const float* noise_dev, float* y_out_dev);
/* Replaces main_ddpir.py:482-517: per image, PSNR of x_0*2-1 against img_H/255*2-1 (utils_image.calculate_psnr_batch terms,
 * max_pixel 2, eps 1e-10) and the same on the Y channel of rgb2ycbcr_batch(only_y) (utils_image.py:470-490; its two zero
 * channels are part of the mean, as in the reference).  x0_dev [B,3,H,W] in [0,1]; outputs: HOST arrays of B floats (syncs). */
int dpir_metrics(dpir_engine* e, const float* x0_dev, const uint8_t* gt_u8_dev, int B, int H, int W, float* psnr_host, float* psnr_y_host);

/* ---- instrumentation --------------------------------------------------------------------- */
/* Kernel-time accounting with HIP events on the engine stream.  class ids: 0 conv3x3, 1 conv1x1,
 * 2 groupnorm-stats, 3 attention, 4 fft-prox, 5 elementwise/other, 6 whole unet forward.
 * dpir_prof_enable(1) makes every launch of a class bracketed by events (slow; bench only). */
#define DPIR_PROF_CLASSES 8
int dpir_prof_enable(dpir_engine* e, int on);
int dpir_prof_reset(dpir_engine* e);
/* ms_out / count_out: arrays of DPIR_PROF_CLASSES */
int dpir_prof_read(dpir_engine* e, double* ms_out, int64_t* count_out);
/* FLOPs (2*MAC, conv+linear+attention) of one UNet forward for one image at HxW; 0 if no model */
double dpir_unet_flops(dpir_engine* e, int H, int W);
/* number of captured step graphs currently cached by dpir_run_loop (one per shape / task / mode, shared by all batches) */
int dpir_graph_cache_size(dpir_engine* e);
/* the same split by profiling class (0 conv3x3, 1 conv1x1 incl. qkv/proj_out, 3 attention matmuls, 5 linears; -1 total) */
double dpir_unet_flops_class(dpir_engine* e, int H, int W, int cls);

#ifdef __cplusplus
}
#endif
#endif /* DIFFPIR_ENGINE_H */
