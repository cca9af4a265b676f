// FFT data-fidelity prox: closed-form Wiener / USRNet solve in the Fourier domain.
//
// Replaces utils/utils_sisr.py:22-41 (p2o), :44-52 (upsample), :78-95 (pre_calculate), :65-75
// (data_solution) and :9-19 (splits).  HBM-bound; every transform is done in LDS.
//
// Layout trick: forward transforms are radix-2 decimation-in-frequency (natural in -> bit-reversed
// out), inverse transforms decimation-in-time (bit-reversed in -> natural out), and ALL spectra (FB,
// F2B, FBFy and the working buffer) stay in the doubly bit-reversed layout.  No permutation pass ever
// runs, and the sf*sf spectral aliases that the reference gathers with chunk/stack/cat ("splits") --
// frequencies (u + i*H/sf, v + j*W/sf) -- become one CONTIGUOUS sf x sf tile of the stored array,
// because the top log2(sf) frequency bits are the bottom log2(sf) storage bits.  The whole solve is
// therefore local to the 16-column strip a workgroup already holds in LDS between its forward and
// inverse column transforms.
#include "common.h"
#include "elem.h"

namespace dpir {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)

// `batch` transforms of N = 1<<logN points living in LDS at d[b*stride + i]; all 256 threads call.
// FWD: DIF, natural -> bit-reversed.  INV: DIT, bit-reversed -> natural (unnormalised).
template <bool INV>
__device__ __forceinline__ void batched_fft_lds(float2* d, int batch, int stride, int logN, const float2* tw) {
    const int N = 1 << logN;
    const int nb = batch << (logN - 1);
    for (int st = 0; st < logN; ++st) {
        const int s = INV ? st : (logN - 1 - st);
        const int half = 1 << s;
        for (int j = threadIdx.x; j < nb; j += blockDim.x) {
            int b = j >> (logN - 1);
            int jj = j & ((N >> 1) - 1);
            int pos = jj & (half - 1);
            int grp = jj >> s;
            int i0 = b * stride + (grp << (s + 1)) + pos;
            int i1 = i0 + half;
            float2 w = tw[pos << (logN - 1 - s)];
            float2 a = d[i0], c = d[i1];
            if (INV) {
                c = cmulc(c, w);
                d[i0] = make_float2(a.x + c.x, a.y + c.y);
                d[i1] = make_float2(a.x - c.x, a.y - c.y);
            } else {
                d[i0] = make_float2(a.x + c.x, a.y + c.y);
                d[i1] = cmul(make_float2(a.x - c.x, a.y - c.y), w);
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void load_twiddles(float2* lds_tw, const float2* tw, int N) {
    for (int i = threadIdx.x; i < (N >> 1); i += blockDim.x) lds_tw[i] = tw[i];
}

// ---------------------------------------------------------------- rows, forward (optionally real input)
// grid: total_rows / R blocks; block: 256 threads; R rows of W points
__global__ __launch_bounds__(256) void fft_rows_fwd_kernel(float2* buf, const float* real_in, float pa, float pb, float pm,
                                                            int W, int logW, int R, size_t total_rows, const float2* tw, const StepDev* sp) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    float2* lds_tw = sm;
    float2* d = sm + (W >> 1);
    if (sp) pm = sp->tau;
    load_twiddles(lds_tw, tw, W);
    size_t row0 = (size_t)blockIdx.x * R;
    for (int i = threadIdx.x; i < R * W; i += 256) {
        size_t row = row0 + (i >> logW);
        float2 v = make_float2(0.f, 0.f);
        if (row < total_rows) {
            size_t g = row * W + (i & (W - 1));
            if (real_in) v.x = (real_in[g] * pa + pb) * pm;
            else v = buf[g];
        }
        d[i] = v;
    }
    __syncthreads();
    batched_fft_lds<false>(d, R, W, logW, lds_tw);
    for (int i = threadIdx.x; i < R * W; i += 256) {
        size_t row = row0 + (i >> logW);
        if (row < total_rows) buf[row * W + (i & (W - 1))] = d[i];
    }
}

// ---------------------------------------------------------------- rows, inverse, real output
__global__ __launch_bounds__(256) void ifft_rows_real_kernel(const float2* buf, float* out, float scale, float oa, float ob,
                                                              const float* blend_base, float g, int W, int logW, int R,
                                                              size_t total_rows, const float2* tw) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    float2* lds_tw = sm;
    float2* d = sm + (W >> 1);
    load_twiddles(lds_tw, tw, W);
    size_t row0 = (size_t)blockIdx.x * R;
    for (int i = threadIdx.x; i < R * W; i += 256) {
        size_t row = row0 + (i >> logW);
        d[i] = row < total_rows ? buf[row * W + (i & (W - 1))] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    batched_fft_lds<true>(d, R, W, logW, lds_tw);
    for (int i = threadIdx.x; i < R * W; i += 256) {
        size_t row = row0 + (i >> logW);
        if (row < total_rows) {
            size_t gi = row * W + (i & (W - 1));
            float v = (d[i].x * scale) * oa + ob;
            if (blend_base) { float b0 = blend_base[gi]; v = b0 + g * (v - b0); }
            out[gi] = v;
        }
    }
}

// ---------------------------------------------------------------- columns: FFT -> (solve) -> inverse FFT
constexpr int CW = 16;   // columns per workgroup (128-byte row segments)

template <int MODE>   // 0: forward only, 1: inverse only, 2: forward + solve + inverse
__global__ __launch_bounds__(256) void fft_cols_kernel(float2* buf, SolveArgs a, int H, int W, int logH, const float2* tw) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    float2* lds_tw = sm;
    float2* d = sm + (H >> 1);
    const int HS = H + 1;                   // padded column stride
    load_twiddles(lds_tw, tw, H);
    const int strips = W / CW;
    const int plane = blockIdx.x / strips;
    const int m0 = (blockIdx.x - plane * strips) * CW;
    float2* base = buf + (size_t)plane * H * W + m0;
    for (int i = threadIdx.x; i < CW * H; i += 256) {
        int c = i & (CW - 1), r = i >> 4;
        d[c * HS + r] = base[(size_t)r * W + c];
    }
    __syncthreads();
    if (MODE != 1) batched_fft_lds<false>(d, CW, HS, logH, lds_tw);
    if (MODE == 2) {
        if (a.sp) a.alpha = a.sp->tau;
        // one work item per sf x sf alias tile (rows/cols are in bit-reversed storage order)
        const int sf = a.sf;
        const int n_img = plane / 3;
        const float2* FB = a.FB + (size_t)n_img * H * W + m0;
        const float* F2B = a.F2B + (size_t)n_img * H * W + m0;
        const float2* FBFy = a.FBFy + (size_t)plane * H * W + m0;
        const int tc = CW / sf, tr = H / sf;
        const float inv_n = 1.0f / (float)(sf * sf);
        for (int it = threadIdx.x; it < tc * tr; it += 256) {
            int cb = it % tc, rb = it / tc;
            float2 fbr = make_float2(0.f, 0.f);
            float invw = 0.f;
            for (int i = 0; i < sf; ++i)
                for (int j = 0; j < sf; ++j) {
                    int r = rb * sf + i, c = cb * sf + j;
                    size_t gi = (size_t)r * W + c;
                    float2 fr = d[c * HS + r];
                    float2 y = FBFy[gi];
                    fr.x += y.x; fr.y += y.y;
                    d[c * HS + r] = fr;
                    float2 x1 = cmul(FB[gi], fr);
                    fbr.x += x1.x; fbr.y += x1.y;
                    invw += F2B[gi];
                }
            fbr.x *= inv_n; fbr.y *= inv_n; invw *= inv_n;
            float den = invw + a.alpha;
            float2 q = make_float2(fbr.x / den, fbr.y / den);
            for (int i = 0; i < sf; ++i)
                for (int j = 0; j < sf; ++j) {
                    int r = rb * sf + i, c = cb * sf + j;
                    size_t gi = (size_t)r * W + c;
                    float2 fr = d[c * HS + r];
                    float2 t = cmulc(q, FB[gi]);        // conj(FB) * q
                    d[c * HS + r] = make_float2((fr.x - t.x) / a.alpha, (fr.y - t.y) / a.alpha);
                }
        }
        __syncthreads();
    }
    if (MODE != 0) batched_fft_lds<true>(d, CW, HS, logH, lds_tw);
    for (int i = threadIdx.x; i < CW * H; i += 256) {
        int c = i & (CW - 1), r = i >> 4;
        base[(size_t)r * W + c] = d[c * HS + r];
    }
}

static int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static Status check_dims(int H, int W) {
    if (!pow2(H) || !pow2(W) || H < 16 || W < 16 || H > 2048 || W > 2048)
        return Status{DPIR_ERR_UNSUPPORTED, "fft prox: H and W must be powers of two in [16, 2048]"};
    return Status{};
}

Status launch_fft_rows(hipStream_t s, const FftPlan& pw, float2* buf, const float* real_in, float pa, float pb,
                       int P, int H, int W, bool inverse) {
    if (inverse) return invalid("launch_fft_rows: inverse goes through launch_ifft_rows_real");
    DPIR_TRY(check_dims(H, W));
    if (pw.N != W) return invalid("fft rows: plan size mismatch");
    int R = W >= 1024 ? 1 : 1024 / W;
    size_t rows = (size_t)P * H;
    size_t lds = ((W >> 1) + (size_t)R * W) * sizeof(float2);
    hipLaunchKernelGGL(fft_rows_fwd_kernel, dim3((unsigned)((rows + R - 1) / R)), dim3(256), lds, s, buf, real_in, pa, pb, 1.0f,
                       W, pw.logN, R, rows, pw.tw, (const StepDev*)nullptr);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// real input with the reference's two-step prologue: v = (x*pa + pb) * pm
Status launch_fft_rows_real3(hipStream_t s, const FftPlan& pw, float2* buf, const float* real_in, float pa, float pb, float pm,
                             int P, int H, int W, const StepDev* sp) {
    DPIR_TRY(check_dims(H, W));
    if (pw.N != W) return invalid("fft rows: plan size mismatch");
    int R = W >= 1024 ? 1 : 1024 / W;
    size_t rows = (size_t)P * H;
    size_t lds = ((W >> 1) + (size_t)R * W) * sizeof(float2);
    hipLaunchKernelGGL(fft_rows_fwd_kernel, dim3((unsigned)((rows + R - 1) / R)), dim3(256), lds, s, buf, real_in, pa, pb, pm,
                       W, pw.logN, R, rows, pw.tw, sp);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

template <int MODE>
static Status launch_cols_mode(hipStream_t s, const FftPlan& ph, float2* buf, const SolveArgs& a, int P, int H, int W) {
    DPIR_TRY(check_dims(H, W));
    if (ph.N != H) return invalid("fft cols: plan size mismatch");
    size_t lds = ((H >> 1) + (size_t)CW * (H + 1)) * sizeof(float2);
    auto fn = fft_cols_kernel<MODE>;
    static LdsAttrOnce attr;
    DPIR_HIP(attr.set(reinterpret_cast<const void*>(fn), 160 * 1024));
    hipLaunchKernelGGL(fn, dim3((unsigned)(P * (W / CW))), dim3(256), lds, s, buf, a, H, W, ph.logN, ph.tw);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

Status launch_fft_cols(hipStream_t s, const FftPlan& ph, float2* buf, int P, int H, int W, bool inverse) {
    SolveArgs a{};
    a.sp = nullptr;
    return inverse ? launch_cols_mode<1>(s, ph, buf, a, P, H, W) : launch_cols_mode<0>(s, ph, buf, a, P, H, W);
}

Status launch_fft_cols_solve(hipStream_t s, const FftPlan& ph, float2* buf, const SolveArgs& a, int B, int H, int W) {
    if (a.sf < 1 || (CW % a.sf) || (H % a.sf)) return Status{DPIR_ERR_UNSUPPORTED, "fft prox: sf must be 1, 2, 4, 8 or 16"};
    return launch_cols_mode<2>(s, ph, buf, a, B * 3, H, W);
}

Status launch_ifft_rows_real(hipStream_t s, const FftPlan& pw, const float2* buf, float* out, float scale, float oa, float ob,
                             const float* blend_base, float g, int P, int H, int W) {
    DPIR_TRY(check_dims(H, W));
    if (pw.N != W) return invalid("ifft rows: plan size mismatch");
    int R = W >= 1024 ? 1 : 1024 / W;
    size_t rows = (size_t)P * H;
    size_t lds = ((W >> 1) + (size_t)R * W) * sizeof(float2);
    hipLaunchKernelGGL(ifft_rows_real_kernel, dim3((unsigned)((rows + R - 1) / R)), dim3(256), lds, s, buf, out, scale, oa, ob,
                       blend_base, g, W, pw.logN, R, rows, pw.tw);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// ---------------------------------------------------------------- pre_calculate helpers
// p2o (utils_sisr.py:22-41): zero-padded PSF circularly shifted by -(kh/2, kw/2), natural order, imag 0
__global__ void psf_embed_kernel(const float* k, int kh, int kw, float2* otf, int H, int W, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t b = i / ((size_t)H * W);
        size_t r = i - b * (size_t)H * W;
        int y = (int)(r / W), x = (int)(r - (size_t)y * W);
        int ky = (y + kh / 2) % H, kx = (x + kw / 2) % W;     // out[y] = in[(y + s) mod H]
        float v = (ky < kh && kx < kw) ? k[(b * kh + ky) * kw + kx] : 0.f;
        otf[i] = make_float2(v, 0.f);
    }
}
Status launch_psf_embed(hipStream_t s, const float* k, int kh, int kw, float2* otf, int B, int H, int W) {
    if (kh > H || kw > W) return invalid("PSF larger than the image");
    size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(psf_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k, kh, kw, otf, H, W, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
// upsample (utils_sisr.py:44-52): zero-stuffing, y at [::sf, ::sf]
__global__ void upsample_embed_kernel(const float* y, int sf, float2* out, int h, int w, size_t total) {
    int H = h * sf, W = w * sf;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i / ((size_t)H * W);
        size_t r = i - p * (size_t)H * W;
        int yy = (int)(r / W), xx = (int)(r - (size_t)yy * W);
        float v = (yy % sf == 0 && xx % sf == 0) ? y[(p * h + yy / sf) * w + xx / sf] : 0.f;
        out[i] = make_float2(v, 0.f);
    }
}
Status launch_upsample_embed(hipStream_t s, const float* y, int sf, float2* out, int P, int h, int w) {
    size_t total = (size_t)P * h * sf * w * sf;
    hipLaunchKernelGGL(upsample_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, y, sf, out, h, w, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
// FBFy <- conj(FB) * F(STy); F2B = |FB|^2      (utils_sisr.py:91-94)
__global__ void precalc_finish_kernel(const float2* FB, float2* FBFy, float* F2B, size_t hw, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i / hw, r = i - p * hw;
        size_t b = p / 3;
        float2 fb = FB[b * hw + r];
        FBFy[i] = cmulc(FBFy[i], fb);
        if (p % 3 == 0) { float m = hypotf(fb.x, fb.y); F2B[b * hw + r] = m * m; }
    }
}
Status launch_precalc_finish(hipStream_t s, const float2* FB, float2* FBFy_inout, float* F2B, int B, int H, int W) {
    size_t hw = (size_t)H * W, total = (size_t)B * 3 * hw;
    hipLaunchKernelGGL(precalc_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, FB, FBFy_inout, F2B, hw, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
