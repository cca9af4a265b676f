// One 256-point complex FFT per WAVE: 64 lanes x 4 points, no workgroup barrier anywhere.
//
// Why: the two-pass register FFT of fft2_body.h gives a thread 16 points, so a whole batch-16 prox is 1.5 waves per SIMD -- a lone wave issues
// a VALU instruction every 5 cycles and nothing hides its memory and LDS round trips (profiles/r05/prox_pmc_sq*.txt: 49-70 % of the wave
// cycles parked).  With 4 points per lane the same transforms are 6 waves per SIMD.
//
// Index algebra (N = 256 = 16 x 16, n = 16 n1 + n2, k = k1 + 16 k2).  lane = 16 g + c, register j:
//   in   : v[j] = x[lane + 64 j]              i.e. n2 = c, n1 = g + 4 j          (a wave instruction touches 64 consecutive elements)
//   pass 1 (16-point DFT over n1 for fixed n2 = c), itself 4 x 4 with n1 = g + 4 j, k1 = a + 4 b:
//          radix 4 over j (registers) -> index a;  x W16^(g a);  4 x 4 transpose of (lane row g, register a) with v_permlane32_swap /
//          v_permlane16_swap (two instructions per register pair, no LDS);  radix 4 over g (registers) -> index b
//          => lane (a, c), register b holds Y[k1 = a + 4 b] of column n2 = c
//   twiddle W256^(n2 k1);  16 x 16 transpose (k1, n2) through a wave-private LDS tile [16][18] (write rows, read columns: both
//          conflict-free at stride 18; same wave writes and reads, LDS operations of a wave execute in order: no barrier)
//          => lane (g', c'), register j' holds element k1 = c', n2 = g' + 4 j'
//   pass 2 = pass 1 over n2  => lane (a', c'), register b' holds X[k1 + 16 k2], k2 = a' + 4 b', i.e.
//   out  : v[b] = X[lane + 64 b]              (the input distribution again: a forward transform can be followed by an inverse in place)
#pragma once
#include "fft_regs.h"

namespace dpir {

struct WaveTw { float2 t16[3]; float2 t256[4]; };     // per-lane constants: W16^(g a), a = 1..3;  W256^(c (g + 4 b)), b = 0..3

// The seven constants of lane l are the same for every wave: the host lays them out as a [7][64] table BEHIND the 256 entries of the W_256^m table
// (wave_tw_fill), so a wave reads them as seven fully contiguous 512-byte loads instead of seven 64-address gathers.
constexpr int WAVE_TW_OFFSET = 256, WAVE_TW_COUNT = 7 * 64;
__device__ __forceinline__ WaveTw wave_tw_load(const float2* tw, int lane) {
    const float2* t = tw + WAVE_TW_OFFSET + lane;
    WaveTw w;
#pragma unroll
    for (int a = 0; a < 3; ++a) w.t16[a] = t[64 * a];
#pragma unroll
    for (int b = 0; b < 4; ++b) w.t256[b] = t[64 * (3 + b)];
    return w;
}
// host: out[WAVE_TW_COUNT] from the table of W_256^m
inline void wave_tw_fill(const float2* w256, float2* out) {
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, c = lane & 15;
        for (int a = 1; a < 4; ++a) out[64 * (a - 1) + lane] = w256[(16 * g * a) & 255];
        for (int b = 0; b < 4; ++b) out[64 * (3 + b) + lane] = w256[(c * (g + 4 * b)) & 255];
    }
}

// first.upper half <-> second.lower half (lanes 32..63 / 0..31)
__device__ __forceinline__ void swap32(float& first, float& second) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(first), __float_as_uint(second), false, false);
    first = __uint_as_float(r[0]); second = __uint_as_float(r[1]);
}
// first.odd rows <-> second.even rows (row = 16 lanes)
__device__ __forceinline__ void swap16(float& first, float& second) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(first), __float_as_uint(second), false, false);
    first = __uint_as_float(r[0]); second = __uint_as_float(r[1]);
}
// (lane row g, register a) -> (lane row a, register g) for the four lanes c, c + 16, c + 32, c + 48
__device__ __forceinline__ void transpose4(float2 (&v)[4]) {
    swap32(v[0].x, v[2].x); swap32(v[0].y, v[2].y); swap32(v[1].x, v[3].x); swap32(v[1].y, v[3].y);
    swap16(v[0].x, v[1].x); swap16(v[0].y, v[1].y); swap16(v[2].x, v[3].x); swap16(v[2].y, v[3].y);
}

// 16-point DFT over m = g + 4 j (lane row g, register j) -> index a + 4 b at (lane row a, register b)
template <bool INV>
__device__ __forceinline__ void quad16(float2 (&v)[4], const float2 (&t16)[3]) {
    fft4<INV>(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int a = 1; a < 4; ++a) v[a] = INV ? cmulc2(v[a], t16[a - 1]) : cmul2(v[a], t16[a - 1]);
    transpose4(v);
    fft4<INV>(v[0], v[1], v[2], v[3]);
}

constexpr int WAVE_FFT_LDS = 16 * 18;       // float2 elements of the wave-private transpose tile

// lds: this wave's tile.  Unnormalised; INV conjugates every twiddle.
template <bool INV>
__device__ __forceinline__ void wave_fft256(float2 (&v)[4], const WaveTw& w, float2* lds, int lane) {
    const int g = lane >> 4, c = lane & 15;
    quad16<INV>(v, w.t16);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float2 y = INV ? cmulc2(v[b], w.t256[b]) : cmul2(v[b], w.t256[b]);
        lds[(g + 4 * b) * 18 + c] = y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = lds[c * 18 + g + 4 * j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    quad16<INV>(v, w.t16);
}

}  // namespace dpir
