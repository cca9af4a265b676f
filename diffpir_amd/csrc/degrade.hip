// On-device degradation synthesis and metrics: the steps either side of the restoration loop.
//
// Replaces CustomDataset.__getitem__'s arithmetic (main_ddpir.py:84-114): wrap-around blur of the uint8 ground truth
// (scipy.ndimage.convolve(img_H, k[..., None], mode='wrap'), whose result is cast back to uint8), x1/sf bicubic down-sampling
// (utils_image.imresize_np == the Resizer weights to 2e-7, reuses band_resample), masking, AWGN in [-1,1] space
// (:112-114, evaluated in float64 like numpy does), and the metrics of main_ddpir.py:482-517 (utils_image.py:601-610 PSNR over
// max_pixel 2, :470-490 the Y channel of rgb2ycbcr_batch).
#include "common.h"
#include "elem.h"

namespace dpir {

// y[b,c,yy,xx] = uint8( sum_{i,j} k[b,i,j] * gt[b, (yy + kh/2 - i) mod H, (xx + kw/2 - j) mod W, c] ) / 255
// float64 accumulation in scipy's tap order (i outer, j inner), C cast (truncation) to uint8, then uint2single.
__global__ __launch_bounds__(256) void blur_wrap_u8_kernel(const uint8_t* gt, const float* k, int kh, int kw, int H, int W, float* out) {
    extern __shared__ float ksh[];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < kh * kw; i += 256) ksh[i] = k[(size_t)b * kh * kw + i];
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int yy = pix / W, xx = pix - yy * W;
    const uint8_t* img = gt + (size_t)b * H * W * 3;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int i = 0; i < kh; ++i) {
        int sy = (yy + kh / 2 - i) % H;
        if (sy < 0) sy += H;
        const uint8_t* row = img + (size_t)sy * W * 3;
        for (int j = 0; j < kw; ++j) {
            int sx = (xx + kw / 2 - j) % W;
            if (sx < 0) sx += W;
            const double w = (double)ksh[i * kw + j];
            const uint8_t* px = row + sx * 3;
            a0 += w * (double)px[0]; a1 += w * (double)px[1]; a2 += w * (double)px[2];
        }
    }
    const double acc[3] = {a0, a1, a2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int q = (int)acc[c];                                     // C cast of the float64 result, as ndimage does
        const uint8_t u = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
        out[((size_t)b * 3 + c) * H * W + pix] = (float)((double)u / 255.0);      // util.uint2single
    }
}

// uint8 NHWC -> float NCHW / 255 (util.uint2single), optionally times the inpainting mask (img_H * mask / 255, main_ddpir.py:108)
__global__ void u8_to_single_kernel(const uint8_t* gt, const uint8_t* mask, int HW, float* out, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / ((size_t)3 * HW), r = i - n * 3 * HW;
        const size_t c = r / HW, p = r - c * HW;
        double v = (double)gt[(n * HW + p) * 3 + c];
        if (mask) v *= (double)mask[i];
        out[i] = (float)(v / 255.0);
    }
}

// img_L = img_L*2-1; img_L += np.random.normal(0, 2 sigma) [float64, IN PLACE: img_L keeps its dtype]; img_L = img_L/2+0.5
// (main_ddpir.py:112-114).  img_L is float32 for deblur / sr (uint2single, imresize_np) and float64 for inpainting
// (uint8 * mask / 255.), where the product with the mask follows in float32 (main_ddpir.py:311-313).
// src64: inpainting's float64 img_L = gt * mask / 255 is rebuilt here from the uint8 ground truth (y then holds nothing yet).
__global__ void degrade_finish_kernel(float* y, const float* noise, double sigma2, const uint8_t* mask, const uint8_t* gt, int HW, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const double nz = noise ? (double)noise[i] * sigma2 : 0.0;
        float f;
        if (gt) {                                   // float64 path
            const size_t n = i / ((size_t)3 * HW), r = i - n * 3 * HW;
            const size_t c = r / HW, p = r - c * HW;
            double v = (double)gt[(n * HW + p) * 3 + c] * (double)mask[i] / 255.0;
            v = v * 2.0 - 1.0;
            v += nz;
            v = v / 2.0 + 0.5;
            f = (float)v * (float)mask[i];
        } else {                                    // float32 path, the noise added in float64 and rounded back
            float v = y[i] * 2.0f - 1.0f;
            v = (float)((double)v + nz);
            f = v / 2.0f + 0.5f;
        }
        y[i] = f;
    }
}

// per image: sum over (c, h, w) of (x0*2-1 - (gt/255*2-1))^2 and of the squared difference of the Y channels
__global__ __launch_bounds__(256) void metrics_kernel(const float* x0, const uint8_t* gt, int HW, double2* out) {
    const int n = blockIdx.x;
    double s = 0.0, sy = 0.0;
    for (int p = threadIdx.x; p < HW; p += 256) {
        float a[3], b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[c] = x0[((size_t)n * 3 + c) * HW + p] * 2.0f - 1.0f;
            b[c] = (float)gt[((size_t)n * HW + p) * 3 + c] / 255.0f * 2.0f - 1.0f;
            const float d = a[c] - b[c];
            s += (double)(d * d);
        }
        const float ya = (0.299f * a[0] + 0.587f * a[1]) + 0.114f * a[2];
        const float yb = (0.299f * b[0] + 0.587f * b[1]) + 0.114f * b[2];
        const float dy = ya - yb;
        sy += (double)(dy * dy);
    }
    __shared__ double red[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); sy += __shfl_xor(sy, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = sy; }
    __syncthreads();
    if (threadIdx.x == 0)
        out[n] = make_double2((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
}

Status launch_blur_wrap_u8(hipStream_t s, const uint8_t* gt, const float* k, int kh, int kw, int B, int H, int W, float* out) {
    if (kh > H || kw > W) return invalid("degrade: PSF larger than the image");
    hipLaunchKernelGGL(blur_wrap_u8_kernel, dim3((unsigned)((H * W + 255) / 256), (unsigned)B), dim3(256), (size_t)kh * kw * sizeof(float), s,
                       gt, k, kh, kw, H, W, out);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_u8_to_single(hipStream_t s, const uint8_t* gt, const uint8_t* mask, int B, int HW, float* out) {
    const size_t total = (size_t)B * 3 * HW;
    hipLaunchKernelGGL(u8_to_single_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gt, mask, HW, out, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_degrade_finish(hipStream_t s, float* y, const float* noise, double sigma2, const uint8_t* mask, const uint8_t* gt, int HW, size_t total) {
    hipLaunchKernelGGL(degrade_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, y, noise, sigma2, mask, gt, HW, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_metrics(hipStream_t s, const float* x0, const uint8_t* gt, int B, int HW, double2* out) {
    hipLaunchKernelGGL(metrics_kernel, dim3((unsigned)B), dim3(256), 0, s, x0, gt, HW, out);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
