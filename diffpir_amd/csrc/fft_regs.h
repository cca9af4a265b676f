// Complex helpers and the in-register R-point FFTs shared by fft2.hip and the development probe (dbg_fft.inc).
#pragma once
#include <hip/hip_runtime.h>
namespace dpir {
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul2(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc2(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
template <bool INV> __device__ __forceinline__ float2 mul_mi(float2 a) {   // a * (-i) forward, a * (+i) inverse
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <bool INV> __device__ __forceinline__ void fft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    float2 s0 = cadd(a0, a2), s1 = csub(a0, a2), s2 = cadd(a1, a3), s3 = mul_mi<INV>(csub(a1, a3));
    a0 = cadd(s0, s2); a2 = csub(s0, s2); a1 = cadd(s1, s3); a3 = csub(s1, s3);
}
template <bool INV> __device__ __forceinline__ void fft2p(float2& a0, float2& a1) {
    float2 t = a0; a0 = cadd(t, a1); a1 = csub(t, a1);
}

// in-register R-point FFT, natural order in and out (R = 8 or 16)
template <int R, bool INV> struct RegFFT;
template <bool INV> struct RegFFT<16, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[16]) {
        const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
        // step 1: 4-point FFTs over n1 for each n2 (x[4 n1 + n2])
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) fft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
        // now v[4 k1 + n2] = t[k1][n2]; twiddle W16^(n2 k1) (conjugated for the inverse)
        const float2 w[10] = {{1.f, 0.f}, {c1, -s1}, {h, -h}, {s1, -c1}, {0.f, -1.f}, {-s1, -c1}, {-h, -h}, {-c1, -s1}, {-1.f, 0.f}, {-c1, s1}};
#pragma unroll
        for (int k1 = 1; k1 < 4; ++k1)
#pragma unroll
            for (int n2 = 1; n2 < 4; ++n2) {
                float2 tw = w[k1 * n2];
                v[4 * k1 + n2] = INV ? cmulc2(v[4 * k1 + n2], tw) : cmul2(v[4 * k1 + n2], tw);
            }
        // step 2: 4-point FFTs over n2 for each k1 -> X[k1 + 4 k2] at position 4 k1 + k2
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) fft4<INV>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
        // transpose 4x4 to natural order: out[k1 + 4 k2] <- v[4 k1 + k2]
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = a + 1; b < 4; ++b) { float2 t = v[4 * a + b]; v[4 * a + b] = v[4 * b + a]; v[4 * b + a] = t; }
    }
};
// 32 = 2 x 16 (decimation in time): X[k] = E[k] + W32^k O[k], X[k + 16] = E[k] - W32^k O[k]
template <bool INV> struct RegFFT<32, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[32]) {
        float2 e[16], o[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
        RegFFT<16, INV>::run(e);
        RegFFT<16, INV>::run(o);
        const float c[16] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                             0.38268343236508977f, 0.19509032201612825f, 0.f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f,
                             -0.70710678118654752f, -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323043f};
        const float sn[16] = {0.f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f, 0.83146961230254524f,
                              0.92387953251128674f, 0.98078528040323043f, 1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                              0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float2 w = make_float2(c[k], -sn[k]);
            const float2 t = INV ? cmulc2(o[k], w) : cmul2(o[k], w);
            v[k] = cadd(e[k], t);
            v[k + 16] = csub(e[k], t);
        }
    }
};
template <bool INV> struct RegFFT<8, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[8]) {
        const float h = 0.70710678118654752f;
        // 8 = 2 x 4: n = 4 n1 + n2 (n1 < 2, n2 < 4), k = k1 + 2 k2
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) fft2p<INV>(v[n2], v[4 + n2]);       // t[k1][n2] at v[4 k1 + n2]
        const float2 w[4] = {{1.f, 0.f}, {h, -h}, {0.f, -1.f}, {-h, -h}};    // W8^(n2) for k1 = 1
#pragma unroll
        for (int n2 = 1; n2 < 4; ++n2) v[4 + n2] = INV ? cmulc2(v[4 + n2], w[n2]) : cmul2(v[4 + n2], w[n2]);
        fft4<INV>(v[0], v[1], v[2], v[3]);                                  // X[0 + 2 k2] at v[k2]
        fft4<INV>(v[4], v[5], v[6], v[7]);                                  // X[1 + 2 k2] at v[4 + k2]
        float2 o[8];
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) { o[2 * k2] = v[k2]; o[2 * k2 + 1] = v[4 + k2]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = o[i];
    }
};
}  // namespace dpir
