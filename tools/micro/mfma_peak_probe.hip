// Sustained rate of v_mfma_f32_32x32x16_f16 with everything else removed: 512-thread workgroups (2 waves per SIMD), 4
// independent accumulators per wave, operands in registers.  Gives the power/clock-limited ceiling that conv4's MFMA phase
// can be compared with.   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/micro/mfma_peak_probe.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 2) void probe(float* out, int iters) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f - threadIdx.x * 0.002f); }
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 1.2345e33f) out[0] = s;
}

int main() {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 1024, 4096}) {
        const int iters = 2000;
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        double flops = (double)blocks * 8 /*waves*/ * iters * 32 * 32768.0;
        printf("blocks %5d: %.3f ms  %.1f TFLOP/s f16 dense (%.1f fp32-equivalent at 3 MFMAs per product)\n", blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 3);
    }
    return 0;
}
