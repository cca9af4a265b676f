// Probe for the operand-split f16 MFMA design (run on the GPU box):
//  1. operand mapping of v_mfma_f32_32x32x16_f16: D = A*B with A[i][k] at lane (i = l%32, g = l/32), element j <-> k = 8g+j,
//     B[k][j'] likewise -- checked against a host reference with ASYMMETRIC operands;
//  2. subnormal f16 inputs: flushed or kept?
//  3. accuracy of hi/lo splitting (3 MFMAs) vs fp32 MFMA vs fp64 reference on random data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void k_f16(const _Float16* a, const _Float16* b, float* out) {   // a: [32][16] row-major, b: [16][32]
    int l = threadIdx.x, i = l & 31, g = l >> 5;
    half8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = a[i * 16 + 8 * g + j]; bv[j] = b[(8 * g + j) * 32 + i]; }
    floatx16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * g; out[row * 32 + i] = c[r]; }
}
__global__ void k_split(const float* a, const float* b, float* out, float sa, float sb) {   // fp32 in, 3 MFMAs
    int l = threadIdx.x, i = l & 31, g = l >> 5;
    half8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
        float x = a[i * 16 + 8 * g + j] * sa; _Float16 h = (_Float16)x; ah[j] = h; al[j] = (_Float16)(x - (float)h);
        float y = b[(8 * g + j) * 32 + i] * sb; _Float16 q = (_Float16)y; bh[j] = q; bl[j] = (_Float16)(y - (float)q);
    }
    floatx16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    float inv = 1.0f / (sa * sb);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * g; out[row * 32 + i] = c[r] * inv; }
}
__global__ void k_f32(const float* a, const float* b, float* out) {   // exact fp32 MFMA, K = 16 via 8 steps of 2
    int l = threadIdx.x, i = l & 31, g = l >> 5;
    floatx16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    for (int kk = 0; kk < 8; ++kk) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * 16 + 2 * kk + g], b[(2 * kk + g) * 32 + i], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * g; out[row * 32 + i] = c[r]; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    std::vector<float> A(32 * 16), B(16 * 32), out(32 * 32);
    srand(1);
    for (auto& v : A) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : B) v = (rand() / (float)RAND_MAX) * 2 - 1;
    float *dA, *dB, *dO; _Float16 *hA, *hB;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dO, out.size() * 4));
    CK(hipMalloc(&hA, A.size() * 2)); CK(hipMalloc(&hB, B.size() * 2));
    // 1. mapping with f16-exact operands
    std::vector<_Float16> Ah(A.size()), Bh(B.size());
    for (size_t i = 0; i < A.size(); ++i) { Ah[i] = (_Float16)A[i]; Bh[i] = (_Float16)B[i]; }
    CK(hipMemcpy(hA, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(hB, Bh.data(), Bh.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_f16, dim3(1), dim3(64), 0, 0, hA, hB, dO);
    CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
    double me = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double r = 0; for (int k = 0; k < 16; ++k) r += (double)(float)Ah[i * 16 + k] * (double)(float)Bh[k * 32 + j]; me = fmax(me, fabs(r - out[i * 32 + j])); }
    printf("1. mapping check (f16-exact operands): max |D - A*B| = %.3e  (%s)\n", me, me < 1e-5 ? "OK" : "WRONG MAPPING");
    // 2. subnormals: a = 2^-20 (subnormal in f16), b = 2^10 -> product 2^-10 per term
    for (auto& v : Ah) v = (_Float16)9.5367431640625e-07f;
    for (auto& v : Bh) v = (_Float16)1024.0f;
    CK(hipMemcpy(hA, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(hB, Bh.data(), Bh.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_f16, dim3(1), dim3(64), 0, 0, hA, hB, dO);
    CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
    printf("2. subnormal f16 A input: D[0][0] = %.6e (expected %.6e if kept, 0 if flushed)\n", out[0], 16 * 9.5367431640625e-07 * 1024.0);
    // 3. accuracy
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    std::vector<double> ref(32 * 32);
    double nrm = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double r = 0; for (int k = 0; k < 16; ++k) r += (double)A[i * 16 + k] * (double)B[k * 32 + j]; ref[i * 32 + j] = r; nrm = fmax(nrm, fabs(r)); }
    for (float sc : {1.0f, 64.0f, 1024.0f}) {
        hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dO, sc, sc);
        CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
        double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(out[i] - ref[i]));
        printf("3. f16x3 split, operand scale %6.0f: max abs err %.3e (rel to max |D| %.3e)\n", sc, e, e / nrm);
    }
    hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
    double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(out[i] - ref[i]));
    printf("3. fp32 MFMA                       : max abs err %.3e (rel to max |D| %.3e)\n", e, e / nrm);
    // small-magnitude operands (|x| < 0.1): where the low halves go subnormal
    for (auto& v : A) v *= 0.05f;
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    nrm = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double r = 0; for (int k = 0; k < 16; ++k) r += (double)A[i * 16 + k] * (double)B[k * 32 + j]; ref[i * 32 + j] = r; nrm = fmax(nrm, fabs(r)); }
    for (float sc : {1.0f, 64.0f}) {
        hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dO, sc, 1.0f);
        CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
        double e2 = 0; for (int i = 0; i < 1024; ++i) e2 = fmax(e2, fabs(out[i] - ref[i]));
        printf("4. small A (|a|<0.05), A scale %4.0f: max abs err %.3e (rel %.3e)\n", sc, e2, e2 / nrm);
    }
    return 0;
}
