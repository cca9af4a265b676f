// conv5: 1x1 convolution (pointwise GEMM  out[co, px] = W[co, ci] * X[ci, px]) on the f16 matrix pipe with operand
// splitting (arithmetic and accuracy: see conv6.hip).  Unlike the 3x3 case there is no 9-tap reuse to amortise a
// separate split pre-pass over, so the fp32 NCHW activations are DMA'd straight into LDS ([16 ch][256 px] per K chunk,
// one 1 KiB piece per channel row; the virtual concat is resolved per piece) and each lane splits its own B fragment
// (8 channels of one pixel) into f16 hi/lo on the VALU, which runs beside the f16 MFMA pipe.
// Tile: 128 output channels x 256 flattened pixels, 4 waves x (128 co x 64 px).  X is read from HBM once per 128 output channels.
//
// Pipeline (r3).  The K loop runs on a THREE-stage LDS ring (24 KiB per stage: 16 KiB activations + 8 KiB weights; <= 80 KiB per
// workgroup, two workgroups per CU) with the DMA of chunk c+2 issued while chunk c is multiplied, ordered by the wave's own counted
// `s_waitcnt vmcnt(N)` and a barrier that waits for LDS operations only (lds_dma.h).  The first version used
// `__builtin_amdgcn_global_load_lds` + `__syncthreads()`: the compiler tracks that builtin's LDS write and put `s_waitcnt vmcnt(0)`
// in front of the first ds_read after EVERY prefetch, i.e. the "double buffer" waited for the piece it had just requested and each
// 16-channel chunk paid a full HBM latency (profiles/r03: 705 us for the 256-channel 256x256 skip projection = 0.95 TB/s).  The
// buffer-descriptor form is not tracked; all ordering here is explicit.
#include "common.h"
#include "lds_dma.h"
#include <vector>
#include <stdlib.h>

namespace dpir {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct Conv5K {
    const float* sa; const float* sb; int ca, cb;
    unsigned bytes_a, bytes_b, bytes_w, bytes_prm;
    const float4* prm;
    const char* w16; const float* bias; float* out; const float* res;
    int B, Cout, HW;
    int n_chunks, n_co_blocks;
    int n_tiles, pair_xcd;            // 256-pixel tiles; co-blocks of a tile on one XCD (n_co_blocks > 1)
    long long total_px;
    float out_scale;
    const float* out_scale_dev;       // optional device scalar multiplied into out_scale (dgrad)
    unsigned long long* range_ctr;
    // EMIT: the kernel also produces the split operand planes of the ResBlock's first 3x3 convolution from the SAME input rows
    // (GroupNorm affine + SiLU per eprm, f16 hi/lo, blocked [n][C8][HW][8] as act.hip writes them): the concat input of an
    // output-path ResBlock is read from HBM once for the 1x1 skip projection and for in_layers (unet.py:236-256).
    const float4* eprm; _Float16* ehi; _Float16* elo; int eC8;
};


// GroupNorm affine (+ SiLU, kernel-uniform flag) of one value; m = {mean, scale, shift, .}
__device__ __forceinline__ float gn_apply(float v, const float4 m, bool silu) {
    float t = (v - m.x) * m.y + m.z;
    if (silu) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));
    return t;
}

// One K chunk of one wave: split this lane's B fragments (8 channels x 2 pixels) out of the staged fp32 rows, 24 (8 in f16x1) MFMAs.
// The LDS operands arrive as __restrict__ parameters ON PURPOSE: after inlining their loads carry alias-scope metadata, and the
// compiler's waitcnt pass then does not put `s_waitcnt vmcnt(0)` in front of every ds_read that follows an LDS-DMA instruction (it
// cannot tell which DMA a read depends on and waits for all of them, i.e. for the prefetch just issued); the counted waits in the
// caller are the real ordering.
template <bool HAS_PRM, bool X1, bool EMIT>
__device__ __forceinline__ bool conv5_chunk(const float* __restrict__ xs, const half8* __restrict__ wh, const float4* __restrict__ ps,
                                            const float4* __restrict__ eprm, const int (&pxl)[2], const int (&pimg)[2], bool silu, bool emit_wg,
                                            _Float16* __restrict__ ehi, _Float16* __restrict__ elo, const size_t (&eoff)[2], size_t echunk,
                                            floatx16 (&acc)[4][2]) {
    constexpr int WCO = 4, WPX = 2;
    const half8* wl = wh + 256;
    bool bad = false;
    half8 bh[WPX], bl[WPX];
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = xs[jj * 256 + pxl[j]];
        if (HAS_PRM) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[jj] = gn_apply(v[jj], ps[pimg[j] * 16 + jj], silu);
        }
        if (EMIT) {
            if (emit_wg) {
                half8 eh, el;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    float t = gn_apply(v[jj], eprm[jj], silu);
                    bad |= !(fabsf(t) <= 65000.f);
                    t = fminf(fmaxf(t, -65000.f), 65000.f);
                    const _Float16 hh = (_Float16)t;
                    eh[jj] = hh;
                    el[jj] = (_Float16)(t - (float)hh);
                }
                const size_t eo = (eoff[j] + echunk) * 8;
                *reinterpret_cast<half8*>(ehi + eo) = eh;
                if (!X1) *reinterpret_cast<half8*>(elo + eo) = el;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            bad |= !(fabsf(v[jj]) <= 65000.f);
            float x = fminf(fmaxf(v[jj], -65000.f), 65000.f);
            _Float16 hh = (_Float16)x;
            bh[j][jj] = hh;
            if (!X1) bl[j][jj] = (_Float16)(x - (float)hh);
        }
    }
    half8 ah[WCO], al[WCO];
#pragma unroll
    for (int i = 0; i < WCO; ++i) { ah[i] = wh[i * 32]; if (!X1) al[i] = wl[i * 32]; }
    if (!X1) {
#pragma unroll
        for (int i = 0; i < WCO; ++i)
#pragma unroll
            for (int j = 0; j < WPX; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < WCO; ++i)
#pragma unroll
            for (int j = 0; j < WPX; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WCO; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    return bad;
}

template <bool HAS_PRM, bool X1, bool EMIT>
__global__ __launch_bounds__(256, 2) void conv5_mfma_kernel(Conv5K p) {
    static_assert(!(HAS_PRM && EMIT), "the emitting variant multiplies the raw input");
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WCO = 4, WPX = 2;
    constexpr int NST = 3;                   // ring stages: DMA runs two chunks ahead of the MFMAs
    constexpr int XBYTES = 16 * 1024;        // [16 ch][256 px] fp32
    constexpr int WBYTES = 8 * 1024;         // [hi|lo][k-half][128 co][8] f16
    constexpr int PBYTES = HAS_PRM ? 1024 : 0;   // [4 images][16 ch] GroupNorm table rows of the chunk
    constexpr int STAGE = XBYTES + WBYTES + PBYTES;
    constexpr int NSTORE = EMIT ? (X1 ? WPX : 2 * WPX) : 0;     // plane stores per wave and chunk (emitting workgroups)
    // ONE LDS object on purpose: with a second __shared__ array in the kernel the compiler's waitcnt pass can no longer tell what an
    // LDS-DMA instruction writes and puts `s_waitcnt vmcnt(0)` in front of the first ds_read after every prefetch (seen in the ISA of
    // the first plane-emitting version: the prefetch distance was zero).
    __shared__ __attribute__((aligned(16))) char smem5[NST * STAGE + 512 + (EMIT ? kConv5EmitMaxC * 16 : 0)];
    float* bias_sh = reinterpret_cast<float*>(smem5 + NST * STAGE);        // this co-block's 128 bias values
    // EMIT: the GroupNorm table of the tile's image, staged once (a 256-pixel tile lies inside one image: HW % 256 == 0); read
    // back as half-wave broadcasts -- per-lane global loads of it (16 per chunk) made the kernel request-bound
    float4* eprm_sh = reinterpret_cast<float4*>(smem5 + NST * STAGE + 512);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int bid = blockIdx.x;             // plain tile order: measured faster than the XCD renumbering conv6.hip uses
    int co_blk, tile;
    if (p.pair_xcd) {
        // several co-blocks read the SAME 256 pixels of X.  Workgroup b runs on XCD b % 8 (own L2): the co-blocks of a tile are numbered 8 apart -- one XCD,
        // dispatched back to back -- so that the second ... n-th read of a row hits that L2 instead of going to HBM again (launch_conv5: DPIR_CONV5_PAIR)
        const int t = bid >> 3;
        co_blk = t % p.n_co_blocks;
        tile = 8 * (t / p.n_co_blocks) + (bid & 7);
        if (tile >= p.n_tiles) return;       // padding workgroups of the last group of eight (before any barrier)
    } else {
        co_blk = bid % p.n_co_blocks;
        tile = bid / p.n_co_blocks;
    }
    const int px0 = tile * 256;             // B * HW < 2^31 (launch_conv5): 32-bit pixel arithmetic
    const int C = p.ca + p.cb;
    const int HW = p.HW;
    const int total_px = (int)p.total_px;
    const int n_tile = px0 / HW;            // first image of the tile
    const bool emit_wg = EMIT && co_blk == 0;
    const bool prm_wave = HAS_PRM && wave == 0;

    // ---- DMA sources.  Activations: one descriptor per concat half, the channel row is the scalar offset, the lane's 4 pixels
    // the vector offset (chunk invariant); lanes past the last pixel and rows past the last channel read out of range = zeros.
    const __amdgpu_buffer_rsrc_t rs_a = rsrc_uniform(p.sa, p.bytes_a);
    const __amdgpu_buffer_rsrc_t rs_b = rsrc_uniform(p.sb ? p.sb : p.sa, p.sb ? p.bytes_b : 0u);
    const __amdgpu_buffer_rsrc_t rs_w = rsrc_uniform(p.w16, p.bytes_w);
    const __amdgpu_buffer_rsrc_t rs_p = rsrc_uniform(HAS_PRM ? (const void*)p.prm : (const void*)p.w16, HAS_PRM ? p.bytes_prm : 0u);
    const int gq = px0 + 4 * lane;
    const bool q_ok = gq < total_px;
    const int qn = q_ok ? gq / HW : 0;
    const int qp = q_ok ? gq - qn * HW : 0;
    const unsigned voffA = q_ok ? (unsigned)(((size_t)qn * p.ca * HW + qp) * 4) : kOutOfRange;
    const unsigned voffB = q_ok ? (unsigned)(((size_t)qn * p.cb * HW + qp) * 4) : kOutOfRange;
    const unsigned lane16 = (unsigned)lane * 16u;
    // GroupNorm rows of the chunk: lane -> (image n_tile + lane / 16, channel lane % 16)
    const unsigned voffP = (HAS_PRM && n_tile + (lane >> 4) < p.B) ? (unsigned)(((size_t)(n_tile + (lane >> 4)) * C + (lane & 15)) * 16) : kOutOfRange;

    auto issue_dma = [&](int chunk, int st) __attribute__((always_inline)) {
        char* stage = smem5 + st * STAGE;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = wave * 4 + u;
            const int c = chunk * 16 + cc;
            char* dst = stage + cc * 1024;
            if (c < p.ca) BLDS6(rs_a, dst, voffA, (unsigned)c * (unsigned)HW * 4u);
            else BLDS6(rs_b, dst, c < C ? voffB : kOutOfRange, (unsigned)(c - p.ca) * (unsigned)HW * 4u);
        }
        const unsigned wsoff = (unsigned)(chunk * p.n_co_blocks + co_blk) * (unsigned)WBYTES;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int piece = wave * 2 + u;
            BLDS6(rs_w, stage + XBYTES + piece * 1024, lane16 + piece * 1024, wsoff);
        }
        if (prm_wave) BLDS6(rs_p, stage + XBYTES + WBYTES, voffP, (unsigned)chunk * 256u);
    };

    // ---- this lane's B-fragment pixels
    int pxl[WPX]; int pimg[WPX];
    size_t eoff[WPX];                      // EMIT: entry index of (image, k-group 0, pixel) in the planes
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        pxl[j] = wave * 64 + j * 32 + l31;
        const int g = px0 + pxl[j];
        const int pn = g < total_px ? g / HW : n_tile;
        pimg[j] = pn - n_tile;             // 0 .. 3: image slot of the staged GroupNorm rows
        eoff[j] = EMIT ? ((size_t)pn * p.eC8 + half) * HW + (size_t)(g - pn * HW) : 0;
    }

    floatx16 acc[WCO][WPX];
#pragma unroll
    for (int i = 0; i < WCO; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bool bad = false;
    bool silu = false;                      // kernel-uniform: every row of a GroupNorm table carries the same flag
    if (EMIT) {
        for (int c = tid; c < C; c += 256) eprm_sh[c] = p.eprm[(size_t)n_tile * C + c];
        silu = __builtin_amdgcn_readfirstlane(p.eprm[0].w != 0.f ? 1 : 0) != 0;
    }
    if (HAS_PRM) silu = __builtin_amdgcn_readfirstlane(p.prm[0].w != 0.f ? 1 : 0) != 0;
    if (tid < 128) bias_sh[tid] = co_blk * 128 + tid < p.Cout ? p.bias[co_blk * 128 + tid] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    issue_dma(0, 0);
    if (p.n_chunks > 1) issue_dma(1, 1);
    int st = 0;                              // ring stage of the current chunk
    for (int chunk = 0; chunk < p.n_chunks; ++chunk) {
        // ---- this wave's pieces of the current chunk have landed.  Vector-memory operations complete in issue order and, counted
        // back from here, the wave has issued: [plane stores of chunk-1] <- [DMA of chunk+1] <- [plane stores of chunk-2] <- [DMA of chunk]
        __builtin_amdgcn_sched_barrier(0);
        if (chunk + 1 >= p.n_chunks) wait_vmcnt<0>();
        else if (emit_wg && chunk > 0) wait_vmcnt<6 + NSTORE>();
        else if (prm_wave) wait_vmcnt<7>();
        else wait_vmcnt<6>();
        barrier_lds_only();                      // ... and everyone's; every wave is also done reading the stage refilled next
        if (chunk + 2 < p.n_chunks) issue_dma(chunk + 2, st == 0 ? 2 : st - 1);
        __builtin_amdgcn_sched_barrier(0);

        const char* stage = smem5 + st * STAGE;
        bad |= conv5_chunk<HAS_PRM, X1, EMIT>(reinterpret_cast<const float*>(stage) + (8 * half) * 256,
                                              reinterpret_cast<const half8*>(stage + XBYTES) + half * 128 + l31,
                                              reinterpret_cast<const float4*>(stage + XBYTES + WBYTES) + 8 * half,
                                              eprm_sh + chunk * 16 + 8 * half, pxl, pimg, silu, emit_wg,
                                              p.ehi, p.elo, eoff, (size_t)chunk * 2 * HW, acc);
        st = st == NST - 1 ? 0 : st + 1;
    }

    {   // f16 operand range guard: see act.hip range_report
        const unsigned long long m = __ballot(bad);
        if (m != 0ull && p.range_ctr && lane == __builtin_ctzll(m)) atomicAdd(p.range_ctr, (unsigned long long)__builtin_popcountll(m));
    }
    // ---- epilogue: un-scale, bias (+ residual), NCHW stores.  The accumulator layout (lanes = pixels, registers = output
    // channels) would give 4-byte-per-lane stores: 128 store instructions per wave for its 128 co x 64 px, and the per-CU store
    // issue rate, not HBM, then bounds these short-K kernels.  Each 32-channel block is therefore transposed through a
    // wave-private LDS slab (the operand buffers are free) so that a lane owns 4 consecutive pixels of one channel:
    // 32 float4 stores per wave, each instruction writing four 256-byte runs.
    __syncthreads();                                   // every wave is done reading the operand buffers
    constexpr int SROW = 68;                           // slab row: 64 pixels + 4 floats of padding (16-byte aligned rows)
    static_assert(4 * 32 * SROW * 4 <= NST * STAGE, "epilogue slabs alias the ring");
    float* slab = reinterpret_cast<float*>(smem5) + wave * (32 * SROW);
    const int row_l = lane >> 4, c4 = lane & 15;
    const int g4 = px0 + wave * 64 + 4 * c4;           // HW % 4 == 0 (conv5_supported): the 4 pixels share an image
    const bool ok4 = g4 < total_px;
    const int n4 = ok4 ? g4 / HW : 0;
    const size_t base4 = (size_t)n4 * p.Cout * HW + (size_t)(g4 - n4 * HW);
    const int co0 = co_blk * 128;
    const float osc = p.out_scale_dev ? p.out_scale * p.out_scale_dev[0] : p.out_scale;
    // No global load between the stores: a load's `s_waitcnt vmcnt(0)` also waits for every store issued before it (the counter
    // is shared and completes in order), which chained the 32 stores of a wave behind one another's write latency.  The bias comes
    // from LDS; the residual rows of a 32-channel block are all requested before its first store.
#pragma unroll
    for (int i = 0; i < WCO; ++i) {
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * half) * SROW + j * 32 + l31] = acc[i][j][r];
        float4 rr[8];
        if (p.res) {
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int co = co0 + i * 32 + ps * 4 + row_l;
                rr[ps] = (ok4 && co < p.Cout) ? *reinterpret_cast<const float4*>(p.res + base4 + (size_t)co * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            wait_vmcnt<0>();             // once per block, so that the compiler does not wait (for the previous store too) before every add
        }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 4 + row_l;
            const int co = co0 + i * 32 + row;
            const float4 a4 = *reinterpret_cast<const float4*>(slab + row * SROW + 4 * c4);
            const float bz = bias_sh[i * 32 + row];
            if (ok4 && co < p.Cout) {
                float4 v = make_float4(a4.x * osc + bz, a4.y * osc + bz, a4.z * osc + bz, a4.w * osc + bz);
                if (p.res) v = make_float4(rr[ps].x + v.x, rr[ps].y + v.y, rr[ps].z + v.z, rr[ps].w + v.w);
                *reinterpret_cast<float4*>(p.out + base4 + (size_t)co * HW) = v;
            }
        }
    }
#endif
}

bool conv5_supported(int B, int Cout, int H, int W, bool has_prm) {
    const long long px = (long long)B * H * W;
    if ((H * W) % 4) return false;
    // the GroupNorm rows of a tile are staged for at most 4 images (conv5.hip PBYTES)
    if (has_prm && (H * W) % 64) return false;
    // enough workgroups to occupy the chip; smaller problems stay on the split-K fp32 kernel
    return ((px + 255) / 256) * ((Cout + 127) / 128) >= 32;
}

Status launch_conv5(hipStream_t s, const Conv5Args& a) {
    if (!conv5_supported(a.B, a.Cout, a.H, a.W, a.prm != nullptr)) return Status{DPIR_ERR_UNSUPPORTED, "conv5: shape not tiled"};
    Conv5K k;
    k.sa = a.src.a; k.sb = a.src.b; k.ca = a.src.ca; k.cb = a.src.b ? a.src.cb : 0; k.prm = a.prm;
    k.w16 = reinterpret_cast<const char*>(a.w16); k.bias = a.bias; k.out = a.out; k.res = a.res;
    k.B = a.B; k.Cout = a.Cout; k.HW = a.H * a.W;
    const int C = k.ca + k.cb;
    k.n_chunks = (C + 15) / 16;
    k.n_co_blocks = (a.Cout + 127) / 128;
    k.total_px = (long long)a.B * a.H * a.W;
    if (k.total_px >= (1ll << 31) - 256) return invalid("conv5: more than 2^31 pixels in the batch");
    // buffer descriptors address 32-bit byte offsets
    const unsigned long long ba = (unsigned long long)k.total_px * k.ca * 4, bb = (unsigned long long)k.total_px * k.cb * 4;
    const unsigned long long bw = (unsigned long long)k.n_chunks * k.n_co_blocks * 8192, bp = (unsigned long long)a.B * C * 16;
    if (ba >= (1ull << 32) || bb >= (1ull << 32) || bw >= (1ull << 32))
        return invalid("conv5: an input tensor exceeds the 4 GiB buffer-descriptor range; reduce the batch");
    k.bytes_a = (unsigned)ba; k.bytes_b = (unsigned)bb; k.bytes_w = (unsigned)bw; k.bytes_prm = (unsigned)bp;
    if (a.prm && C % 16) return invalid("conv5: the GroupNorm variant needs a multiple of 16 input channels");
    k.out_scale = 1.0f / a.w16_scale;
    k.out_scale_dev = a.out_scale_dev;
    k.range_ctr = a.range_ctr;
    k.eprm = a.emit_prm; k.ehi = reinterpret_cast<_Float16*>(a.emit_hi); k.elo = reinterpret_cast<_Float16*>(a.emit_lo);
    k.eC8 = 2 * k.n_chunks;
    k.n_tiles = (int)((k.total_px + 255) / 256);
    static const bool pair_env = !(getenv("DPIR_CONV5_PAIR") && atoi(getenv("DPIR_CONV5_PAIR")) == 0);
    k.pair_xcd = pair_env && k.n_co_blocks > 1 ? 1 : 0;
    const unsigned blocks = k.pair_xcd ? (unsigned)((k.n_tiles + 7) / 8 * 8 * k.n_co_blocks) : (unsigned)(k.n_tiles * k.n_co_blocks);
    if (a.emit_hi) {
        if (a.prm || !a.emit_prm || (!a.x1 && !a.emit_lo)) return invalid("conv5: the plane-emitting variant takes the raw input and a GroupNorm table for the planes");
        if ((a.H * a.W) % 256 || C % 16 || C > kConv5EmitMaxC) return invalid("conv5: the plane-emitting variant needs H*W % 256 == 0 and a multiple of 16, at most 384, input channels");
        if (a.x1) hipLaunchKernelGGL((conv5_mfma_kernel<false, true, true>), dim3(blocks), dim3(256), 0, s, k);
        else hipLaunchKernelGGL((conv5_mfma_kernel<false, false, true>), dim3(blocks), dim3(256), 0, s, k);
    } else if (a.x1) {
        if (a.prm) hipLaunchKernelGGL((conv5_mfma_kernel<true, true, false>), dim3(blocks), dim3(256), 0, s, k);
        else hipLaunchKernelGGL((conv5_mfma_kernel<false, true, false>), dim3(blocks), dim3(256), 0, s, k);
    } else if (a.prm) hipLaunchKernelGGL((conv5_mfma_kernel<true, false, false>), dim3(blocks), dim3(256), 0, s, k);
    else hipLaunchKernelGGL((conv5_mfma_kernel<false, false, false>), dim3(blocks), dim3(256), 0, s, k);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// Host: OI fp32 -> [chunk (16 ci)][co-block (128)][hi|lo][k-half][128 co][8] f16, scaled by a power of two so that
// max|w|*scale is in [512, 1024).  Returns the scale.
float pack_weights_f16x3_1x1(const float* w, int cout, int cin, std::vector<uint16_t>& out) {
    const int chunks = (cin + 15) / 16, cblocks = (cout + 127) / 128;
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin; ++i) mx = fmaxf(mx, fabsf(w[i]));
    float scale = 1.0f;
    if (mx > 0.f) scale = exp2f(floorf(log2f(1024.0f / mx)));
    while (mx * scale >= 1024.0f) scale *= 0.5f;
    const size_t plane = (size_t)2 * 128 * 8;
    out.assign((size_t)chunks * cblocks * 2 * plane, 0);
    for (int ch = 0; ch < chunks; ++ch)
        for (int cbk = 0; cbk < cblocks; ++cbk) {
            uint16_t* hi = out.data() + ((size_t)ch * cblocks + cbk) * 2 * plane;
            uint16_t* lo = hi + plane;
            for (int kh = 0; kh < 2; ++kh)
                for (int col = 0; col < 128; ++col)
                    for (int j = 0; j < 8; ++j) {
                        int co = cbk * 128 + col, ci = ch * 16 + kh * 8 + j;
                        float v = (co < cout && ci < cin) ? w[(size_t)co * cin + ci] * scale : 0.f;
                        _Float16 h = (_Float16)v;
                        _Float16 l = (_Float16)(v - (float)h);
                        size_t o = ((size_t)kh * 128 + col) * 8 + j;
                        __builtin_memcpy(&hi[o], &h, 2);
                        __builtin_memcpy(&lo[o], &l, 2);
                    }
        }
    return scale;
}

}  // namespace dpir
