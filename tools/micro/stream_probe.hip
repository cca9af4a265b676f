// Calibration for short memory-bound kernels (the FFT prox passes move 25-45 MB each at batch 16): what does a dependent chain of plain float4
// streaming kernels cost per kernel on this machine, as a function of bytes moved and grid size?  Captured as ONE hipGraph of 30 dependent
// launches (no host launch cost), timed with two events.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/build/stream_probe tools/micro/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// out[i] = a[i] + b[i] (nb = 1) or a[i] (nb = 0); grid-stride float4
__global__ __launch_bounds__(256) void stream_kernel(const float4* a, const float4* b, float4* out, size_t n4, int nb) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        if (nb) { const float4 w = b[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        out[i] = v;
    }
}
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 1024) p[0] = 1.f; }

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t maxb = (size_t)256 << 20;
    float4 *a, *b, *c; CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&c, maxb));
    CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb)); CK(hipMemset(c, 0, maxb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int chain = 30;
    auto run = [&](const char* what, size_t bytes_per_stream, int nb, int grid) -> int {
        const size_t n4 = bytes_per_stream / 16;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < chain; ++i) {
            if (bytes_per_stream == 0) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, (float*)nullptr);
            else hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, s, (i & 1) ? c : a, b, (i & 1) ? a : c, n4, nb);     // dependent: ping-pong
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double us = best * 1e3 / chain, moved = (double)bytes_per_stream * (2 + nb);
        printf("%-34s grid %5d: %7.2f us per kernel", what, grid, us);
        if (bytes_per_stream) printf("  %6.1f MB moved -> %5.2f TB/s", moved / 1e6, moved / us / 1e6);
        printf("\n");
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return 0;
    };
    for (int grid : {256, 1024, 2048}) if (run("empty kernel", 0, 0, grid)) return 1;
    for (size_t mb : {1, 4, 12, 25, 50, 100, 200})
        for (int grid : {512, 1024, 2048, 4096, 8192})
            if (run((std::to_string(mb) + " MiB in + same out (copy)").c_str(), mb << 20, 0, grid)) return 1;
    for (size_t mb : {12, 50}) for (int grid : {2048, 4096}) if (run((std::to_string(mb) + " MiB x2 in + out (add)").c_str(), mb << 20, 1, grid)) return 1;
    return 0;
}
