// Development-only probes (include/diffpir_debug.h); not on the product path.
// dpir_debug_victim: workgroups that park a known pattern in LDS and in registers, wait, and verify it -- used by
// tools/concurrent_check*.py to find out what a kernel of ANOTHER engine running on the same GPU can disturb.
#include "engine.h"
#include "../../include/diffpir_debug.h"

namespace dpir {
__global__ void victim_kernel(int lds_words, long long spin, unsigned long long* bad) {
    extern __shared__ unsigned vsm[];
    const unsigned key = blockIdx.x * 2654435761u + 12345u;
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) vsm[i] = key ^ (unsigned)(i * 40503u);
    unsigned r[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) r[k] = key + threadIdx.x * 97u + k * 7919u;
#pragma unroll
    for (int k = 0; k < 32; ++k) asm volatile("" : "+v"(r[k]));
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    unsigned long long nb = 0;
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) nb += vsm[i] != (key ^ (unsigned)(i * 40503u));
#pragma unroll
    for (int k = 0; k < 32; ++k) { asm volatile("" : "+v"(r[k])); nb += (r[k] != key + threadIdx.x * 97u + k * 7919u) ? (1ull << 32) : 0ull; }
    if (nb) atomicAdd(bad, nb);
}

// mode 0: v_add_f32 x2, 1: v_pk_add_f32, 2: v_pk_fma_f32, 3: v_pk_mul_f32 (by +-1) -- exact small-integer arithmetic, verified at the end
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ void victim_alu_kernel(int iters, int mode, unsigned long long* bad) {
    const float lx = (float)(threadIdx.x % 64 + 1), ly = (float)(2 * (threadIdx.x % 64) + 1);
    f2v acc = {1.f, 2.f};
    const f2v inc = {lx, ly}, ones = {1.f, 1.f}, flip = {-1.f, 1.f};
    float sx = 1.f, sy = 2.f;
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(sx) : "v"(lx)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(sy) : "v"(ly)); }
        else if (mode == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(inc));
        else if (mode == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(ones), "v"(inc));
        else { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc) : "v"(flip)); }
    }
    bool ok;
    if (mode == 0) ok = sx == 1.f + lx * iters && sy == 2.f + ly * iters;
    else if (mode == 3) ok = acc.x == ((iters & 1) ? -1.f : 1.f) && acc.y == 2.f;
    else ok = acc.x == 1.f + lx * iters && acc.y == 2.f + ly * iters;
    if (!ok) atomicAdd(bad, 1ull);
}
}  // namespace dpir

extern "C" int dpir_debug_victim(dpir_engine* e, int lds_bytes, int threads, int blocks, long long spin_ticks, int iters,
                                 unsigned long long* bad_out) {
    if (!e || !bad_out) return DPIR_ERR_INVALID;
    (void)hipSetDevice(e->device);
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 8) != hipSuccess) return DPIR_ERR_HIP;
    (void)hipMemsetAsync(d, 0, 8, e->stream);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dpir::victim_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(dpir::victim_kernel, dim3(blocks), dim3(threads), lds_bytes, e->stream, lds_bytes / 4, spin_ticks, d);
    int rc = hipMemcpyAsync(bad_out, d, 8, hipMemcpyDeviceToHost, e->stream) == hipSuccess && hipStreamSynchronize(e->stream) == hipSuccess ? DPIR_OK : DPIR_ERR_HIP;
    (void)hipFree(d);
    return rc;
}

extern "C" int dpir_debug_victim_alu(dpir_engine* e, int mode, int blocks, int iters_in_kernel, int launches, unsigned long long* bad_out) {
    if (!e || !bad_out) return DPIR_ERR_INVALID;
    (void)hipSetDevice(e->device);
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 8) != hipSuccess) return DPIR_ERR_HIP;
    (void)hipMemsetAsync(d, 0, 8, e->stream);
    for (int i = 0; i < launches; ++i)
        hipLaunchKernelGGL(dpir::victim_alu_kernel, dim3(blocks), dim3(64), 0, e->stream, iters_in_kernel, mode, d);
    int rc = hipMemcpyAsync(bad_out, d, 8, hipMemcpyDeviceToHost, e->stream) == hipSuccess && hipStreamSynchronize(e->stream) == hipSuccess ? DPIR_OK : DPIR_ERR_HIP;
    (void)hipFree(d);
    return rc;
}
