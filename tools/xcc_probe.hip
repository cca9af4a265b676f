// Which XCD does workgroup b run on?  blockIdx.x, HW_REG_XCC_ID, HW_REG_HW_ID per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* o) {
    if (threadIdx.x == 0) {
        o[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;       // XCC_ID[3:0]
        o[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_ID
    }
}
int main() {
    const int G = 1024; unsigned* d; hipMalloc(&d, G * 8);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d); std::vector<unsigned> h(2 * G); hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    int hist[16] = {}, mism = 0;
    for (int b = 0; b < G; ++b) { hist[h[2 * b] & 15]++; if ((h[2 * b] & 7) != (unsigned)(b % 8)) ++mism; }
    printf("xcc histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\nblocks with xcc != b %% 8: %d of %d\n", mism, G);
    for (int b = 0; b < 24; ++b) printf("b %d xcc %u hwid %08x\n", b, h[2 * b], h[2 * b + 1]);
    return 0;
}
