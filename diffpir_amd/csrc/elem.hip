// Elementwise / resampling / RNG kernels of the DiffPIR loop (HBM-bound, float4-vectorised where the
// layout allows).  Compiled with -ffp-contract=off so that the arithmetic order written here -- which
// mirrors the reference expression by expression -- is what the hardware executes.
//
// Replaces: eps->x0 clamp (guided_diffusion/gaussian_diffusion.py:297,328-333), masked prox
// (main_ddpir.py:392-394), re-noise (main_ddpir.py:451-456), init (main_ddpir.py:293-315), output
// (main_ddpir.py:470 + utils/utils_image.py:238-242), Resizer (utils/utils_resizer.py:55-74),
// IBP prox (main_ddpir.py:401-406), torch bicubic interpolate (main_ddpir.py:295), torch.randn_like.
#include "common.h"
#include "elem.h"
#include "philox.h"

namespace dpir {

static inline dim3 grid1d(size_t n, int per_block = 256) {
    size_t b = (n + per_block - 1) / per_block;
    if (b > 65535u * 16u) b = 65535u * 16u;
    return dim3((unsigned)b);
}
#define GRID_STRIDE(i, n) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

// ---------------------------------------------------------------- eps -> clamped x0
__global__ void xstart_kernel(const float* x, const float* out6, int out_ch, float c1, float c2, float* x0, size_t chw, size_t total,
                              const StepDev* sp) {
    if (sp) { c1 = sp->c1; c2 = sp->c2; }
    GRID_STRIDE(i, total) {
        size_t n = i / chw, r = i - n * chw;
        float eps = out6[n * (chw / 3) * out_ch + r];
        float v = c1 * x[i] - c2 * eps;
        x0[i] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
}
Status launch_xstart(hipStream_t s, const float* x, const float* out6, int out_ch, float c1, float c2, float* x0, int B, int HW, const StepDev* sp) {
    size_t total = (size_t)B * 3 * HW;
    hipLaunchKernelGGL(xstart_kernel, grid1d(total), dim3(256), 0, s, x, out6, out_ch, c1, c2, x0, (size_t)3 * HW, total, sp);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- masked prox
__global__ void prox_mask_kernel(float* x0, const float* y, const uint8_t* mask, float tau, float g, size_t total, const StepDev* sp,
                                 const LoopDev* lp) {
    if (sp) tau = sp->tau;
    if (lp) { y = lp->y; mask = lp->mask; }
    GRID_STRIDE(i, total) {
        float m = (float)mask[i];
        float v = x0[i];
        float num = m * (2.0f * y[i] - 1.0f) + tau * v;
        float xp = num / (m + tau);
        x0[i] = v + g * (xp - v);
    }
}
Status launch_prox_mask(hipStream_t s, float* x0, const float* y, const uint8_t* mask, float tau, float g, size_t total, const StepDev* sp,
                        const LoopDev* lp) {
    hipLaunchKernelGGL(prox_mask_kernel, grid1d(total), dim3(256), 0, s, x0, y, mask, tau, g, total, sp, lp);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- repaint conditioning (main_ddpir.py:355-358)
// x = (sqrt_ac[t] * (2y - 1) + sqrt_1m_ac[t] * n) * mask + (1 - mask) * x      (generate_mode == 'repaint', inpainting)
__global__ void repaint_mix_kernel(float* x, const float* y, const uint8_t* mask, const float* n, float sa, float s1m, size_t total,
                                   const StepDev* sp, size_t stride, const LoopDev* lp) {
    if (lp) { y = lp->y; mask = lp->mask; if (stride) n = lp->nrp; }
    if (sp) { sa = sp->sa_t; s1m = sp->s1m_t; n += (size_t)sp->i * stride; }
    GRID_STRIDE(i, total) {
        float m = (float)mask[i];
        float known = sa * (2.0f * y[i] - 1.0f) + s1m * n[i];
        x[i] = known * m + (1.0f - m) * x[i];
    }
}
Status launch_repaint_mix(hipStream_t s, float* x, const float* y, const uint8_t* mask, const float* n, float sa, float s1m, size_t total,
                          const StepDev* sp, size_t noise_step_stride, const LoopDev* lp) {
    hipLaunchKernelGGL(repaint_mix_kernel, grid1d(total), dim3(256), 0, s, x, y, mask, n, sa, s1m, total, sp, noise_step_stride, lp);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- re-noise
__global__ void renoise_kernel(float* x, const float* x0, RenoiseCoef c, const float* n1, const float* n2, size_t total,
                               const StepDev* sp, size_t stride, const LoopDev* lp) {
    if (lp && stride) { if (n1) n1 = lp->n1; n2 = lp->n2; }     // host-fed noise: this batch's tensors
    if (sp) {
        c.sa_t = sp->sa_t; c.s1m_t = sp->s1m_t; c.sa_p = sp->sa_p; c.k1 = sp->k1; c.q = sp->q; c.es = sp->es; c.k2 = sp->k2;
        if (n1) n1 += (size_t)sp->i * stride;
        n2 += (size_t)sp->i * stride;
    }
    GRID_STRIDE(i, total) {
        float a = x0[i];
        float eps = (x[i] - c.sa_t * a) / c.s1m_t;
        float inner = c.q * eps;
        if (n1) inner = inner + c.es * n1[i];
        float v = c.sa_p * a + c.k1 * inner;
        v = v + c.k2 * n2[i];
        x[i] = v;
    }
}
Status launch_renoise(hipStream_t s, float* x, const float* x0, const RenoiseCoef& c, const float* n1, const float* n2, size_t total,
                      const StepDev* sp, size_t noise_step_stride, const LoopDev* lp) {
    hipLaunchKernelGGL(renoise_kernel, grid1d(total), dim3(256), 0, s, x, x0, c, n1, n2, total, sp, noise_step_stride, lp);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- init: x = sa*(2*src-1) + s1m*noise
__global__ void init_x_kernel(const float* src, const uint8_t* mask, const float* noise, float sa, float s1m, float* x, size_t total) {
    GRID_STRIDE(i, total) {
        float v = src[i];
        if (mask) v = v * (float)mask[i];
        x[i] = sa * (2.0f * v - 1.0f) + s1m * noise[i];
    }
}
Status launch_init_x(hipStream_t s, const float* src, const uint8_t* mask, const float* noise, float sa, float s1m, float* x, size_t total) {
    hipLaunchKernelGGL(init_x_kernel, grid1d(total), dim3(256), 0, s, src, mask, noise, sa, s1m, x, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- finalize: x/2+.5 (f32 NCHW) and u8 NHWC
__global__ void finalize_kernel(const float* x, float* of, uint8_t* ou, int HW, size_t total) {
    GRID_STRIDE(i, total) {
        float v = x[i] / 2.0f + 0.5f;
        if (of) of[i] = v;
        if (ou) {
            size_t n = i / ((size_t)3 * HW);
            size_t r = i - n * 3 * HW;
            size_t c = r / HW, p = r - c * HW;
            float q = fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f;
            ou[(n * HW + p) * 3 + c] = (uint8_t)rintf(q);
        }
    }
}
// One float32 operation per element, rounded once, as a torch elementwise op does (this file is built with -ffp-contract=off).
__global__ void ewise_kernel(int op, const float* x, const float* y, int y_bcast, float scalar, float* out, size_t total) {
    const float ys = y && y_bcast ? y[0] : scalar;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float a = x[i], b = (y && !y_bcast) ? y[i] : ys;
        float r;
        switch (op) {
            case 0: r = a + b; break;
            case 1: r = a - b; break;
            case 2: r = a * b; break;
            case 3: r = a / b; break;
            case 4: r = b - a; break;
            default: r = b / a; break;
        }
        out[i] = r;
    }
}
Status launch_ewise(hipStream_t s, int op, const float* x, const float* y, size_t y_numel, float scalar, float* out, size_t total) {
    if (op < 0 || op > 5) return Status{DPIR_ERR_INVALID, "ewise: unknown op"};
    if (y && y_numel != 1 && y_numel != total) return Status{DPIR_ERR_INVALID, "ewise: the second operand must have the same number of elements or one"};
    if (!total) return Status{};
    size_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(ewise_kernel, dim3((unsigned)blocks), dim3(256), 0, s, op, x, y, (int)(y && y_numel == 1), scalar, out, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

Status launch_finalize(hipStream_t s, const float* x, float* of, uint8_t* ou, int B, int HW) {
    size_t total = (size_t)B * 3 * HW;
    hipLaunchKernelGGL(finalize_kernel, grid1d(total), dim3(256), 0, s, x, of, ou, HW, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- affine map (x*a + b), used for x/2+.5 etc.
__global__ void affine_kernel(const float* x, float a, float b, float* out, size_t total) {
    GRID_STRIDE(i, total) out[i] = x[i] * a + b;
}
Status launch_affine(hipStream_t s, const float* x, float a, float b, float* out, size_t total) {
    hipLaunchKernelGGL(affine_kernel, grid1d(total), dim3(256), 0, s, x, a, b, out, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- Resizer: banded gather along one axis
// in [P, L_in, inner] -> out [P, L_out, inner]; out[p,o,i] = sum_t w[t*L_out+o] * f(in[p, idx[t*L_out+o], i])
// with f(v) = v*pa + pb (fuses z = x0/2+.5 into the first pass)
__global__ void band_resample_kernel(const float* in, const float* w, const int* idx, int taps, int L_in, int L_out,
                                     int inner, float pa, float pb, float* out, size_t total) {
    GRID_STRIDE(i, total) {
        size_t p = i / ((size_t)L_out * inner);
        size_t r = i - p * (size_t)L_out * inner;
        int o = (int)(r / inner), ii = (int)(r - (size_t)o * inner);
        const float* base = in + p * (size_t)L_in * inner + ii;
        float acc = 0.f;
        for (int t = 0; t < taps; ++t) {
            float v = base[(size_t)idx[t * L_out + o] * inner] * pa + pb;
            acc = acc + v * w[t * L_out + o];
        }
        out[i] = acc;
    }
}
Status launch_band_resample(hipStream_t s, const float* in, const float* w, const int* idx, int taps, int P, int L_in,
                            int L_out, int inner, float pa, float pb, float* out) {
    size_t total = (size_t)P * L_out * inner;
    hipLaunchKernelGGL(band_resample_kernel, grid1d(total), dim3(256), 0, s, in, w, idx, taps, L_in, L_out, inner, pa, pb, out, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// IBP update: x0 <- 2*( z + gamma*(y - d)[up nearest]/(1+rho) ) - 1, z = x0/2+.5   (main_ddpir.py:404-406)
__global__ void ibp_update_kernel(float* x0, const float* y, const float* d, float gamma, float rho, int sf, int H, int W, size_t total,
                                  const StepDev* sp, const LoopDev* lp) {
    if (sp) rho = sp->tau;
    if (lp) y = lp->y;
    GRID_STRIDE(i, total) {
        size_t plane = i / ((size_t)H * W);
        size_t r = i - plane * (size_t)H * W;
        int yy = (int)(r / W), xx = (int)(r - (size_t)yy * W);
        int h = H / sf, w = W / sf;
        size_t li = plane * (size_t)h * w + (size_t)(yy / sf) * w + xx / sf;
        float z = x0[i] / 2.0f + 0.5f;
        float diff = y[li] - d[li];
        z = z + gamma * diff / (1.0f + rho);
        x0[i] = z * 2.0f - 1.0f;
    }
}
Status launch_ibp_update(hipStream_t s, float* x0, const float* y, const float* d, float gamma, float rho, int sf, int P, int H, int W,
                         const StepDev* sp, const LoopDev* lp) {
    size_t total = (size_t)P * H * W;
    hipLaunchKernelGGL(ibp_update_kernel, grid1d(total), dim3(256), 0, s, x0, y, d, gamma, rho, sf, H, W, total, sp, lp);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- torch bicubic (A=-0.75, align_corners=False)
__device__ __forceinline__ void cubic_coeffs(float t, float* c) {
    const float A = -0.75f;
    float x = t + 1.0f;
    c[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
    x = t;
    c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 1.0f - t;
    c[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 2.0f - t;
    c[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}
__global__ void bicubic_up_kernel(const float* in, float* out, int h, int w, int H, int W, float sy, float sx, size_t total) {
    GRID_STRIDE(i, total) {
        size_t plane = i / ((size_t)H * W);
        size_t r = i - plane * (size_t)H * W;
        int oy = (int)(r / W), ox = (int)(r - (size_t)oy * W);
        float fy = sy * ((float)oy + 0.5f) - 0.5f;
        float fx = sx * ((float)ox + 0.5f) - 0.5f;
        int iy = (int)floorf(fy), ix = (int)floorf(fx);
        float cy[4], cx[4];
        cubic_coeffs(fy - (float)iy, cy);
        cubic_coeffs(fx - (float)ix, cx);
        const float* p = in + plane * (size_t)h * w;
        float acc = 0.f;
        for (int a = 0; a < 4; ++a) {
            int yy = min(max(iy - 1 + a, 0), h - 1);
            float row = 0.f;
            for (int b = 0; b < 4; ++b) {
                int xx = min(max(ix - 1 + b, 0), w - 1);
                row = row + p[(size_t)yy * w + xx] * cx[b];
            }
            acc = acc + row * cy[a];
        }
        out[i] = acc;
    }
}
Status launch_bicubic_up(hipStream_t s, const float* in, float* out, int P, int h, int w, int sf) {
    int H = h * sf, W = w * sf;
    size_t total = (size_t)P * H * W;
    hipLaunchKernelGGL(bicubic_up_kernel, grid1d(total), dim3(256), 0, s, in, out, h, w, H, W, (float)h / (float)H, (float)w / (float)W, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- Philox4x32-10 + Box-Muller (philox.h)
// one thread produces 4 normals for elements [4j, 4j+4) of image (image_offset + n); counter = (j, image, stream)
__global__ void randn_kernel(float* out, uint64_t seed, uint64_t stream_id, int64_t image_offset, size_t per_image, size_t total4,
                             const StepDev* sp, const LoopDev* lp) {
    // draw kinds per step: 1 = eta term, 2 = zeta term, 3 = repaint mix -> a stride of 4 keeps every (kind, step) pair distinct
    if (sp) stream_id += 4 * (uint64_t)sp->i;
    if (lp) { seed = lp->seed; image_offset = lp->image_offset; }
    GRID_STRIDE(i, total4) {
        size_t q = (per_image + 3) / 4;
        size_t n = i / q, j = i - n * q;
        float z[4];
        philox_normal4(seed, stream_id, (uint64_t)(image_offset + (int64_t)n), j, z);
        float* o = out + n * per_image + j * 4;
        for (int e = 0; e < 4; ++e)
            if (j * 4 + e < per_image) o[e] = z[e];
    }
}
Status launch_randn(hipStream_t s, float* out, uint64_t seed, uint64_t stream_id, int64_t image_offset, int B, size_t per_image,
                    const StepDev* sp, const LoopDev* lp) {
    size_t total4 = (size_t)B * ((per_image + 3) / 4);
    hipLaunchKernelGGL(randn_kernel, grid1d(total4), dim3(256), 0, s, out, seed, stream_id, image_offset, per_image, total4, sp, lp);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- host: Resizer band tables (utils_resizer.py:104-167)
static double cubic_k(double x) {
    double a = fabs(x), a2 = a * a, a3 = a2 * a;
    if (a <= 1) return 1.5 * a3 - 2.5 * a2 + 1;
    if (a <= 2) return -0.5 * a3 + 2.5 * a2 - 4 * a + 2;
    return 0.0;
}
void resizer_band(int in_len, int out_len, double scale, std::vector<float>& w_out, std::vector<int>& idx_out, int& taps_out) {
    bool aa = scale < 1;
    double kw = aa ? 4.0 / scale : 4.0;
    int ekw = (int)ceil(kw) + 2;
    std::vector<double> w((size_t)out_len * ekw);
    std::vector<int> fov((size_t)out_len * ekw);
    for (int o = 0; o < out_len; ++o) {
        double oc = o + 1;
        double shifted = oc - (out_len - in_len * scale) / 2;
        double match = shifted / scale + 0.5 * (1 - 1 / scale);
        double left = floor(match - kw / 2);
        double sum = 0;
        for (int t = 0; t < ekw; ++t) {
            int f = (int)(int16_t)(left + t - 1);
            double arg = match - f - 1;
            double v = aa ? scale * cubic_k(scale * arg) : cubic_k(arg);
            w[(size_t)o * ekw + t] = v;
            fov[(size_t)o * ekw + t] = f;
            sum += v;
        }
        if (sum == 0) sum = 1;
        for (int t = 0; t < ekw; ++t) w[(size_t)o * ekw + t] /= sum;
        for (int t = 0; t < ekw; ++t) {   // mirror boundary: index into [0..n-1, n-1..0]
            int m = 2 * in_len;
            int f = fov[(size_t)o * ekw + t] % m;
            if (f < 0) f += m;
            fov[(size_t)o * ekw + t] = f < in_len ? f : (2 * in_len - 1 - f);
        }
    }
    // drop tap columns that are zero for every output position
    std::vector<int> keep;
    for (int t = 0; t < ekw; ++t) {
        bool any = false;
        for (int o = 0; o < out_len; ++o) if (w[(size_t)o * ekw + t] != 0) { any = true; break; }
        if (any) keep.push_back(t);
    }
    taps_out = (int)keep.size();
    w_out.resize((size_t)taps_out * out_len);
    idx_out.resize((size_t)taps_out * out_len);
    for (int k = 0; k < taps_out; ++k)
        for (int o = 0; o < out_len; ++o) {
            w_out[(size_t)k * out_len + o] = (float)w[(size_t)o * ekw + keep[k]];
            idx_out[(size_t)k * out_len + o] = fov[(size_t)o * ekw + keep[k]];
        }
}

}  // namespace dpir
