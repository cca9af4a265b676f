// conv8: GroupNorm affine + SiLU + f16 hi/lo split + 3x3 convolution to AT MOST 16 output channels in ONE kernel -- the network's
// output layer `out = conv3x3(SiLU(GroupNorm32(h)))`, 128 -> 6 channels at full resolution (guided_diffusion/unet.py:612-616,
// forward :662), round 5.
//
// Why a kernel of its own.  Until round 4 this layer ran as act_split (read h fp32, write the normalised / activated / split f16 planes:
// 537 MB in + 537 MB out at B = 16, ~200 us) followed by conv7's NARROW variant (~216 us; its 32x32x16 MFMA tile has 32 output-channel
// rows of which 6 are live, so the layer was matrix-pipe bound on zeros).  The layer's compulsory traffic is h once (537 MB) and 25 MB of
// output; here the fp32 rows go HBM -> registers -> (GroupNorm, SiLU, split) -> LDS as MFMA operands, never back to HBM, and the product
// runs on v_mfma_f32_16x16x32_f16 (16 output-channel rows, half the padding).  Arithmetic is the f16x3 scheme of conv6.hip: per product
// al*bh, ah*bl, ah*bh accumulated in fp32, weights pre-scaled by a power of two (f16x1: ah*bh only).
//
// Workgroup: 512 threads (8 waves), tile 8 rows x 32 columns of one image; K in steps of 32 input channels (4 groups of 8 = one half8 MFMA
// operand entry).  Per step: the 10 x 34 halo patch of 32 channels (fp32, buffer loads prefetched into registers one step ahead so that
// the HBM latency overlaps the previous step's MFMAs), transformed and written to LDS as [group][position] half8 hi / lo (43.5 KiB); the
// step's 18 weight fragments (9 taps x hi / lo, 18 KiB, L2 hits) travel the same way.  Wave w owns tile row w = 2 pixel tiles of 16 and
// issues 9 x 2 x 3 MFMAs per step, the fragments of tap t + 1 requested before the MFMAs of tap t.  ~66 KiB LDS and <= 128 VGPRs -> two
// workgroups per CU = FOUR waves per SIMD.  That occupancy is the point of the shape: the prologue is ~20 VALU instructions per element
// (GroupNorm affine, exp2 + rcp, range guard, two conversions) on 1.33 x the elements (halo), and a SIMD with one or two waves issues a
// VALU instruction only every 5 / 2.5 cycles (tools/micro/valu_issue_probe.hip) -- the 256-thread version (two waves per SIMD, 212 VGPRs)
// took 350-400 us per launch at B = 16 whatever was prefetched (profiles/r05/conv8_*_kernel_trace.txt).
#include "common.h"
#include "elem.h"
#include "lds_dma.h"
#include <vector>
#include <cmath>

namespace dpir {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct Conv8K {
    const float* x;            // [B][C][H][W]
    const float4* prm;         // [B][C] {mean, scale, shift, silu flag} (gn_prm_kernel)
    const half8* w;            // [C/32][9 taps][hi|lo][64 lanes] A fragments
    const float* bias; float* out;
    int B, C, Cout, H, W, tiles_x, tiles_y;
    float out_scale;
    unsigned long long* range_ctr;
};

constexpr int C8_TW = 32, C8_TH = 8, C8_LW = 34, C8_LH = 10, C8_PATCH = C8_LW * C8_LH, C8_KG = 4;
constexpr int C8_ITEMS = C8_KG * C8_PATCH;                 // (group, position) entries per K step
constexpr int C8_THREADS = 512;                            // 8 waves: one tile row (2 pixel tiles of 16) each
constexpr int C8_NIT = (C8_ITEMS + C8_THREADS - 1) / C8_THREADS;   // per thread
constexpr int C8_WFR = 18 * 64;                            // weight entries (half8) per K step
constexpr int C8_MAXC = 256;

__device__ __forceinline__ float silu8(float v) {          // act.hip's silu_a
    float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

template <bool X1>
__global__ __launch_bounds__(C8_THREADS, 2) void conv8_fused_kernel(Conv8K p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ half8 s_hi[C8_ITEMS];
    __shared__ half8 s_lo[X1 ? 1 : C8_ITEMS];
    __shared__ half8 s_w[C8_WFR];
    __shared__ float4 s_prm[C8_MAXC];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);       // XCD-contiguous tiles: halo re-reads hit that XCD's L2
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int n = bid / tiles_per_img;
    const int trem = bid - n * tiles_per_img;
    const int ty0 = (trem / p.tiles_x) * C8_TH, tx0 = (trem % p.tiles_x) * C8_TW;
    const int HW = p.H * p.W;
    const int ksteps = p.C >> 5;

    for (int c = tid; c < p.C; c += C8_THREADS) s_prm[c] = p.prm[(size_t)n * p.C + c];

    // this thread's (group, position) items: global offset of channel 0 of the group inside the image, or -1 outside the image (zero padding
    // of the ACTIVATED tensor: the padded value is 0, not SiLU(GroupNorm(0)))
    int goff[C8_NIT], sidx[C8_NIT], grp[C8_NIT];
#pragma unroll
    for (int it = 0; it < C8_NIT; ++it) {
        const int item = it * C8_THREADS + tid;
        const int g = item / C8_PATCH, pos = item - g * C8_PATCH;
        const int hy = pos / C8_LW, hx = pos - hy * C8_LW;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool ok = item < C8_ITEMS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        goff[it] = ok ? g * 8 * HW + gy * p.W + gx : -1;
        sidx[it] = item < C8_ITEMS ? item : -1;
        grp[it] = g < C8_KG ? g : 0;
    }
    const float* ximg = p.x + (size_t)n * p.C * HW;

    // BRANCH-FREE on purpose.  The first version guarded every load and every element's transform with `goff >= 0` / the SiLU flag: the compiler
    // turned each of the elements of a step (48 per thread in that 256-thread version; C8_NIT x 8 = 24 at 512 threads) into its own basic block with an `s_waitcnt lgkmcnt(0)` (the GroupNorm table read) in front
    // of a branch -- 400 us per launch at B = 16 (profiles/r05/conv8_first_version_kernel_trace.txt).  Here every lane loads (out-of-image
    // items read offset 0 of their plane, a valid address) and the padding zero is a select at the end.
    // Buffer loads: ONE descriptor for the image (SGPRs), one byte-offset VGPR per item, the channel term (ks * 32 + j) * HW * 4 in the
    // scalar offset.  With flat `global_load` every one of the loads of a step (48 per thread at 256 threads) kept its own 64-bit address (96 VGPRs): spills at the
    // 256-VGPR budget of two workgroups per CU.
    float v[C8_NIT][8];
    const __amdgpu_buffer_rsrc_t rx = rsrc_uniform(ximg, (unsigned)((size_t)p.C * HW * 4));
    unsigned boff[C8_NIT];
#pragma unroll
    for (int it = 0; it < C8_NIT; ++it) boff[it] = (unsigned)(goff[it] >= 0 ? goff[it] : grp[it] * 8 * HW) * 4u;
    auto prefetch = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < C8_NIT; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[it][j] = as_f32(__builtin_amdgcn_raw_buffer_load_b32(rx, boff[it], (unsigned)((ks * 32 + j) * HW) * 4u, 0));
    };
    // the step's 18 weight fragments travel the same way: requested one step ahead into registers, written to LDS at the top of the step.
    // (First version: load -> s_waitcnt vmcnt(0) -> ds_write, five times in a row at the top of EVERY step -- five exposed L2 round trips
    // per step, the largest single item of the 376 us that version took.)
    constexpr int NWQ = (C8_WFR + C8_THREADS - 1) / C8_THREADS;
    half8 wq[NWQ];
    auto wfetch = [&](int ks) __attribute__((always_inline)) {
        const half8* wsrc = p.w + (size_t)ks * C8_WFR;
#pragma unroll
        for (int q = 0; q < NWQ; ++q) {
            const int i = q * C8_THREADS + tid;
            wq[q] = wsrc[i < C8_WFR ? i : 0];
        }
    };
    wfetch(0);
    prefetch(0);

    constexpr int NT = 2;                    // pixel tiles per wave: row `wave`, columns 0-15 and 16-31
    floatx4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
    bool bad = false;
    // B fragment base of this lane: pixel = lane & 15 of a 16-pixel tile, K group = lane >> 4
    const int lb = (lane >> 4) * C8_PATCH + (lane & 15);

    for (int ks = 0; ks < ksteps; ++ks) {
        __syncthreads();                      // the previous step's fragment reads are done (first pass: s_prm is staged)
        // weights of this step -> LDS (requested during the previous step)
#pragma unroll
        for (int q = 0; q < NWQ; ++q) {
            const int i = q * C8_THREADS + tid;
            if (i < C8_WFR) s_w[i] = wq[q];
        }
        // activations: GroupNorm affine + SiLU + split (the layer always activates: launch_conv8 refuses a table without the SiLU flag)
#pragma unroll
        for (int it = 0; it < C8_NIT; ++it) {
            half8 h8, l8;
            const int c0 = ks * 32 + grp[it] * 8;
            const bool inside = goff[it] >= 0;
            float4 m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = s_prm[c0 + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = silu8((v[it][j] - m[j].x) * m[j].y + m[j].z);
                t = inside ? t : 0.f;
                bad |= !(fabsf(t) <= 65000.f);
                t = fminf(fmaxf(t, -65000.f), 65000.f);
                const _Float16 hh = (_Float16)t;
                h8[j] = hh;
                l8[j] = (_Float16)(t - (float)hh);
            }
            if (sidx[it] >= 0) {
                s_hi[sidx[it]] = h8;
                if (!X1) s_lo[sidx[it]] = l8;
            }
            // keep the scheduler from hoisting the NEXT items' table reads above this point: all items' 8 x float4 at once (six items = 192 registers in the 256-thread version, C8_NIT = 3 here) are 192
            // live VGPRs on top of the 48 prefetched values -> spills; one item's 32 cost ~60 cycles of exposed LDS latency per item
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ks + 1 < ksteps) { wfetch(ks + 1); prefetch(ks + 1); }     // in flight during the MFMAs below
        barrier_lds_only();                        // LDS writes of every wave landed; the prefetch is NOT waited for
        // MFMA phase: the fragments of tap t + 1 (2 A + 8 B, LDS) are requested BEFORE the 12 MFMAs of tap t are issued, in a two-set register
        // ring (the scheduling barriers pin that order: left alone, the compiler put every ds_read right in front of the MFMA that consumes
        // it -- `s_waitcnt lgkmcnt(1)` ten times per tap, i.e. an exposed LDS latency per MFMA group)
        half8 fah[2], fal[2], fbh[2][NT], fbl[2][NT];
        auto ldfr = [&](int tap, int st) __attribute__((always_inline)) {
            const int dy = tap / 3, dx = tap - dy * 3;
            fah[st] = s_w[(tap * 2 + 0) * 64 + lane];
            if (!X1) fal[st] = s_w[(tap * 2 + 1) * 64 + lane];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int idx = lb + (wave + dy) * C8_LW + t * 16 + dx;
                fbh[st][t] = s_hi[idx];
                if (!X1) fbl[st][t] = s_lo[idx];
            }
        };
        ldfr(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int st = tap & 1;
            if (tap + 1 < 9) ldfr(tap + 1, st ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (!X1) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fal[st], fbh[st][t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (!X1) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah[st], fbl[st][t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah[st], fbh[st][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // epilogue: D[row = 4 (lane >> 4) + i = output channel][col = lane & 15 = pixel]
    const int co0 = (lane >> 4) * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int gy = ty0 + wave, gx = tx0 + t * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + i;
            if (co < p.Cout) p.out[((size_t)n * p.Cout + co) * HW + (size_t)gy * p.W + gx] = acc[t][i] * p.out_scale + p.bias[co];
        }
    }
    const unsigned long long mbad = __ballot(bad);
    if (mbad != 0ull && p.range_ctr && lane == (int)__builtin_ctzll(mbad)) atomicAdd(p.range_ctr, (unsigned long long)__builtin_popcountll(mbad));
#endif
}

bool conv8_supported(int B, int C, int Cout, int H, int W) {
    return B > 0 && Cout >= 1 && Cout <= 16 && C % 32 == 0 && C >= 32 && C <= C8_MAXC && H % C8_TH == 0 && W % C8_TW == 0 &&
           (size_t)C * H * W < ((size_t)1 << 30);      // one image's bytes fit the 32-bit range of a buffer descriptor
}

Status launch_conv8(hipStream_t s, const Conv8Args& a) {
    if (!conv8_supported(a.B, a.C, a.Cout, a.H, a.W)) return invalid("conv8: shape not supported");
    if (!a.x || !a.prm || !a.w || !a.bias || !a.out) return invalid("conv8: null operand");
    if (!a.silu) return invalid("conv8: the fused prologue is GroupNorm + SiLU (the output layer); a table without activation takes the planes route");
    Conv8K k;
    k.x = a.x; k.prm = a.prm; k.w = reinterpret_cast<const half8*>(a.w); k.bias = a.bias; k.out = a.out;
    k.B = a.B; k.C = a.C; k.Cout = a.Cout; k.H = a.H; k.W = a.W;
    k.tiles_x = a.W / C8_TW; k.tiles_y = a.H / C8_TH;
    k.out_scale = 1.0f / a.w_scale;
    k.range_ctr = a.range_ctr;
    const int blocks = a.B * k.tiles_x * k.tiles_y;
    if (a.x1) hipLaunchKernelGGL(conv8_fused_kernel<true>, dim3(blocks), dim3(C8_THREADS), 0, s, k);
    else hipLaunchKernelGGL(conv8_fused_kernel<false>, dim3(blocks), dim3(C8_THREADS), 0, s, k);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// Host: OIHW fp32 (cout <= 16) -> [K step (32 ci)][tap][hi|lo][lane: m = lane & 15 (co), group = lane >> 4][8 ci] f16, i.e. the A
// fragments of v_mfma_f32_16x16x32_f16 in lane order; scaled by a power of two so that max|w| * scale is in [512, 1024) (conv6's rule).
float pack_weights_conv8(const float* w, int cout, int cin, std::vector<uint16_t>& out) {
    const int taps = 9, ksteps = cin / 32;
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin * taps; ++i) mx = fmaxf(mx, fabsf(w[i]));
    float scale = 1.0f;
    if (mx > 0.f) scale = exp2f(floorf(log2f(1024.0f / mx)));
    while (mx * scale >= 1024.0f) scale *= 0.5f;
    out.assign((size_t)ksteps * taps * 2 * 64 * 8, 0);
    for (int ks = 0; ks < ksteps; ++ks)
        for (int tap = 0; tap < taps; ++tap)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = lane & 15, ci = ks * 32 + (lane >> 4) * 8 + j;
                    const float v = co < cout ? w[((size_t)co * cin + ci) * taps + tap] * scale : 0.f;
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    const size_t base = ((size_t)ks * taps + tap) * 2;
                    __builtin_memcpy(&out[((base + 0) * 64 + lane) * 8 + j], &h, 2);
                    __builtin_memcpy(&out[((base + 1) * 64 + lane) * 8 + j], &l, 2);
                }
    return scale;
}

}  // namespace dpir
