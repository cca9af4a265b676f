// VALU issue rate on gfx950 at 1, 2, 4 and 8 waves per SIMD: cycles per wave64 instruction per SIMD for the non-packed
// v_fma_f32 / v_add_f32 and the packed v_pk_fma_f32, with 8 independent chains per wave (throughput) and with ONE dependent
// chain per wave (issue-to-issue latency of a dependent VALU op).  Settles the 2-vs-4-cycle question of DESIGN.md section 3.4
// (MI355X_MICROARCH.md: "v_fma_f32 (wave64) 2 cyc (SIMD-32)").  Cycles come from s_memtime inside the wave, so the figure does
// not depend on the clock the box happens to run at; the wall-clock figure is printed next to it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/micro/valu_issue_probe.hip && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float float2v __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(1024) void probe(unsigned long long* cyc, float* sink, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    float2v pb = {b, b}, pc = {c, c};
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // 8 independent v_fma_f32 chains, 64 instructions per iteration
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (MODE == 1) {     // 8 independent v_add_f32 chains
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 2) {     // 8 independent v_pk_fma_f32 chains (2 fp32 lanes per instruction)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        } else if (MODE == 3) {     // ONE dependent v_fma_f32 chain, 64 instructions per iteration
            REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                              "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                              : "+v"(a0) : "v"(b), "v"(c));)
        } else {                    // ONE dependent v_pk_fma_f32 chain
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                              "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2"
                              : "+v"(p0) : "v"(pb), "v"(pc));)
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x;
    if (s == 1.2345e33f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned long long* dcyc, float* dsink) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 8}) {
        // one CU holds 4 SIMDs: a 256*w-thread workgroup puts w waves on each; 8 waves per SIMD = two 1024-thread workgroups per CU
        int threads = wps == 8 ? 1024 : 256 * wps, blocks = wps == 8 ? 512 : 256;
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, dcyc, dsink, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, dcyc, dsink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int nw = blocks * threads / 64;
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), dcyc, nw * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        double med = (double)h[nw / 2], instr = iters * 64.0;
        // s_memtime ticks at a constant 100 MHz on this part or at the shader clock: report both raw ticks and the wall-clock figure
        double us = ms * 1e3;
        printf("%-22s %d waves/SIMD: s_memtime ticks/instr/wave %.3f | wall %.1f us -> %.3f ns per wave-instruction per SIMD (= %.2f cyc at 2.4 GHz, %.2f at 1.7 GHz)\n",
               name, wps, med / instr, us, us * 1e3 / (instr * wps), us * 1e3 / (instr * wps) * 2.4, us * 1e3 / (instr * wps) * 1.7);
    }
}

int main() {
    unsigned long long* dcyc; float* dsink;
    hipMalloc(&dcyc, 512 * 16 * 8); hipMalloc(&dsink, 4);
    run<0>("v_fma_f32 x8 indep", dcyc, dsink);
    run<1>("v_add_f32 x8 indep", dcyc, dsink);
    run<2>("v_pk_fma_f32 x8 indep", dcyc, dsink);
    run<3>("v_fma_f32 dependent", dcyc, dsink);
    run<4>("v_pk_fma_f32 dependent", dcyc, dsink);
    return 0;
}
