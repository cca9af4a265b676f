// One 256-point (or 512-point: the second half of this file) complex FFT per WAVE: 64 lanes x 4 (8) points, no workgroup barrier anywhere.
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

// Per-lane constants.  N = 256: t[0..2] = W16^(g a), a = 1..3;  t[3..6] = W256^(c (g + 4 b)), b = 0..3  (g = lane >> 4, c = lane & 15).
//                     N = 512: t[0..6] = W64^(g8 a), a = 1..7;  t[7..14] = W512^(n2 (g8 + 8 b)), b = 0..7  (g8 = lane >> 3, n2 = lane & 7).
template <int N> struct WaveTwN { float2 t[N == 256 ? 7 : 15]; };
typedef WaveTwN<256> WaveTw;

// The constants of lane l are the same for every wave: the host lays them out as a [7 | 15][64] table BEHIND the N entries of the W_N^m table
// (wave_tw_fill), so a wave reads them as fully contiguous 512-byte loads instead of 64-address gathers.
constexpr int wave_tw_count(int N) { return (N == 256 ? 7 : 15) * 64; }
template <int N> __device__ __forceinline__ WaveTwN<N> wave_tw_load(const float2* tw, int lane) {
    const float2* t = tw + N + lane;
    WaveTwN<N> w;
#pragma unroll
    for (int a = 0; a < (N == 256 ? 7 : 15); ++a) w.t[a] = t[64 * a];
    return w;
}
// host: out[wave_tw_count(N)] from the table of W_N^m
inline void wave_tw_fill(int N, const float2* wN, float2* out) {
    for (int lane = 0; lane < 64; ++lane) {
        if (N == 256) {
            const int g = lane >> 4, c = lane & 15;
            for (int a = 1; a < 4; ++a) out[64 * (a - 1) + lane] = wN[(16 * g * a) & 255];
            for (int b = 0; b < 4; ++b) out[64 * (3 + b) + lane] = wN[(c * (g + 4 * b)) & 255];
        } else {
            const int g8 = lane >> 3, n2 = lane & 7;
            for (int a = 1; a < 8; ++a) out[64 * (a - 1) + lane] = wN[(8 * g8 * a) & 511];
            for (int b = 0; b < 8; ++b) out[64 * (7 + b) + lane] = wN[(n2 * (g8 + 8 * b)) & 511];
        }
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
// first.lanes with bit 3 set <-> second.lanes with bit 3 clear: DPP row_ror:8 reads lane ^ 8 inside a row of 16, the bank mask (a bank = 4 lanes of a
// row) selects which half of the destination takes it -- two v_mov_dpp per dword, no select
__device__ __forceinline__ void swap8(float& first, float& second) {
    const int f = (int)__float_as_uint(first), s = (int)__float_as_uint(second);
    const int nf = __builtin_amdgcn_update_dpp(f, s, 0x128, 0xF, 0xC, false);
    const int ns = __builtin_amdgcn_update_dpp(s, f, 0x128, 0xF, 0x3, false);
    first = __uint_as_float((unsigned)nf); second = __uint_as_float((unsigned)ns);
}
// (lane row g, register a) -> (lane row a, register g) for the four lanes c, c + 16, c + 32, c + 48
__device__ __forceinline__ void transpose4(float2 (&v)[4]) {
    swap32(v[0].x, v[2].x); swap32(v[0].y, v[2].y); swap32(v[1].x, v[3].x); swap32(v[1].y, v[3].y);
    swap16(v[0].x, v[1].x); swap16(v[0].y, v[1].y); swap16(v[2].x, v[3].x); swap16(v[2].y, v[3].y);
}
// (lane bits 5..3 = g8, register a) -> (lane bits 5..3 = a, register g8): one swap per (lane bit, register bit) pair
__device__ __forceinline__ void transpose8(float2 (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { swap32(v[r].x, v[r + 4].x); swap32(v[r].y, v[r + 4].y); }
#pragma unroll
    for (int r = 0; r < 8; ++r) if (!(r & 2)) { swap16(v[r].x, v[r + 2].x); swap16(v[r].y, v[r + 2].y); }
#pragma unroll
    for (int r = 0; r < 8; r += 2) { swap8(v[r].x, v[r + 1].x); swap8(v[r].y, v[r + 1].y); }
}

// 16-point DFT over m = g + 4 j (lane row g, register j) -> index a + 4 b at (lane row a, register b)
template <bool INV>
__device__ __forceinline__ void quad16(float2 (&v)[4], const float2* t16) {
    fft4<INV>(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int a = 1; a < 4; ++a) v[a] = INV ? cmulc2(v[a], t16[a - 1]) : cmul2(v[a], t16[a - 1]);
    transpose4(v);
    fft4<INV>(v[0], v[1], v[2], v[3]);
}

// float2 elements of the wave-private exchange tile: 16 x 18 (N = 256), 64 x 9 (N = 512)
constexpr int wave_fft_lds(int N) { return N == 256 ? 16 * 18 : 64 * 9; }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// lds: this wave's tile.  Unnormalised; INV conjugates every twiddle.  v[j] = x[lane + 64 j] in, X[lane + 64 j] out.
template <bool INV>
__device__ __forceinline__ void wave_fft(float2 (&v)[4], const WaveTwN<256>& w, float2* lds, int lane) {
    const int g = lane >> 4, c = lane & 15;
    quad16<INV>(v, w.t);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float2 y = INV ? cmulc2(v[b], w.t[3 + b]) : cmul2(v[b], w.t[3 + b]);
        lds[(g + 4 * b) * 18 + c] = y;
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = lds[c * 18 + g + 4 * j];
    wave_sync();
    quad16<INV>(v, w.t);
}

// N = 512 = 8 x 8 x 8, n = 8 n1 + n2 with n2 = lane & 7 and n1 = g8 + 8 j (g8 = lane >> 3), k = k1 + 64 k2:
//   64-point DFT over n1 = radix 8 over j (registers) -> a;  x W64^(g8 a);  8 x 8 transpose (lane bits 5..3, register);  radix 8 over g8 -> b:
//          lane (a, n2), register b holds Y[n2][k1 = a + 8 b]
//   x W512^(n2 k1);  exchange through the tile [k1][9] (written with stride-9 rows, read as lane k1's eight consecutive entries: both conflict-free bar three
//          two-way collisions of the write);  radix 8 over n2 -> k2: lane k1, register k2.
template <bool INV>
__device__ __forceinline__ void wave_fft(float2 (&v)[8], const WaveTwN<512>& w, float2* lds, int lane) {
    const int g8 = lane >> 3, n2 = lane & 7;
    RegFFT<8, INV>::run(v);
#pragma unroll
    for (int a = 1; a < 8; ++a) v[a] = INV ? cmulc2(v[a], w.t[a - 1]) : cmul2(v[a], w.t[a - 1]);
    transpose8(v);
    RegFFT<8, INV>::run(v);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const float2 y = INV ? cmulc2(v[b], w.t[7 + b]) : cmul2(v[b], w.t[7 + b]);
        lds[(g8 + 8 * b) * 9 + n2] = y;
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lds[lane * 9 + j];
    wave_sync();
    RegFFT<8, INV>::run(v);
}

}  // namespace dpir
