// conv5: 1x1 convolution (pointwise GEMM  out[co, px] = W[co, ci] * X[ci, px]) on the f16 matrix pipe with operand
// splitting (arithmetic and accuracy: see conv6.hip).  Unlike the 3x3 case there is no 9-tap reuse to amortise a
// separate split pre-pass over, so the fp32 NCHW activations are DMA'd straight into LDS ([16 ch][256 px] per K chunk,
// one 1 KiB piece per channel row; the virtual concat is resolved per piece) and each lane splits its own B fragment
// (8 channels of one pixel) into f16 hi/lo on the VALU, which runs beside the f16 MFMA pipe.
// Tile: 128 output channels x 256 flattened pixels, 4 waves x (128 co x 64 px); LDS 48 KiB -> two workgroups per CU, so
// one workgroup's epilogue overlaps the other's MFMAs.  X is read from HBM once per 128 output channels.
#include "common.h"
#include <vector>

namespace dpir {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct Conv5K {
    const float* sa; const float* sb; int ca, cb;
    const float4* prm;
    const char* w16; const float* bias; float* out; const float* res;
    int B, Cout, HW;
    int n_chunks, n_co_blocks;
    long long total_px;
    const float* zeros;
    float out_scale;
    const float* out_scale_dev;       // optional device scalar multiplied into out_scale (dgrad)
    unsigned long long* range_ctr;
    // EMIT: the kernel also produces the split operand planes of the ResBlock's first 3x3 convolution from the SAME input rows
    // (GroupNorm affine + SiLU per eprm, f16 hi/lo, blocked [n][C8][HW][8] as act.hip writes them): the concat input of an
    // output-path ResBlock is read from HBM once for the 1x1 skip projection and for in_layers (unet.py:236-256).
    const float4* eprm; _Float16* ehi; _Float16* elo; int eC8;
};

#define GLDS5(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

template <bool HAS_PRM, bool X1, bool EMIT>
__global__ __launch_bounds__(256, 2) void conv5_mfma_kernel(Conv5K p) {
    static_assert(!(HAS_PRM && EMIT), "the emitting variant multiplies the raw input");
    constexpr int WCO = 4, WPX = 2;
    constexpr int XBYTES = 16 * 1024;        // [16 ch][256 px] fp32
    constexpr int WBYTES = 8 * 1024;         // [hi|lo][k-half][128 co][8] f16
    __shared__ __attribute__((aligned(16))) char smem5[2 * XBYTES + 2 * WBYTES];
    char* lds_x = smem5;
    char* lds_w = smem5 + 2 * XBYTES;
    // EMIT: the GroupNorm table of the tile's image, staged once (a 256-pixel tile lies inside one image: HW % 256 == 0); read
    // back as half-wave broadcasts -- per-lane global loads of it (16 per chunk) made the kernel request-bound
    __shared__ float4 eprm_sh[EMIT ? 1024 : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int bid = blockIdx.x;             // plain order: measured faster than the XCD renumbering conv6.hip uses
    const int co_blk = bid % p.n_co_blocks;
    const long long px0 = (long long)(bid / p.n_co_blocks) * 256;
    const int C = p.ca + p.cb;
    const int HW = p.HW;

    // ---- DMA source of this lane's 4 pixels (chunk invariant part)
    const long long gq = px0 + 4 * lane;
    const bool q_ok = gq < p.total_px;
    const int qn = q_ok ? (int)(gq / HW) : 0;
    const int qp = q_ok ? (int)(gq - (long long)qn * HW) : 0;
    const size_t offA = (size_t)qn * p.ca * HW + qp;
    const size_t offB = (size_t)qn * p.cb * HW + qp;

    auto issue_dma = [&](int chunk, int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = wave * 4 + u;
            const int c = chunk * 16 + cc;
            const float* src = p.zeros;
            if (q_ok && c < C) src = c < p.ca ? p.sa + offA + (size_t)c * HW : p.sb + offB + (size_t)(c - p.ca) * HW;
            GLDS5(src, lds_x + buf * XBYTES + cc * 1024);
        }
        const char* wsrc = p.w16 + ((size_t)chunk * p.n_co_blocks + co_blk) * WBYTES + lane * 16;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int piece = wave * 2 + u;
            GLDS5(wsrc + piece * 1024, lds_w + buf * WBYTES + piece * 1024);
        }
    };

    // ---- this lane's B-fragment pixels
    int pxl[WPX]; int pn[WPX];
    size_t eoff[WPX]; bool eok[WPX];       // EMIT: entry index of (image, k-group 0, pixel) in the planes
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        pxl[j] = wave * 64 + j * 32 + l31;
        long long g = px0 + pxl[j];
        pn[j] = g < p.total_px ? (int)(g / HW) : 0;
        eok[j] = EMIT && co_blk == 0 && g < p.total_px;
        eoff[j] = EMIT ? ((size_t)pn[j] * p.eC8 + half) * HW + (size_t)(g - (long long)pn[j] * HW) : 0;
    }

    floatx16 acc[WCO][WPX];
#pragma unroll
    for (int i = 0; i < WCO; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bool bad = false;
    if (EMIT) {
        const int n_tile = (int)(px0 / HW);
        for (int c = tid; c < C; c += 256) eprm_sh[c] = p.eprm[(size_t)n_tile * C + c];
    }
    issue_dma(0, 0);
    for (int chunk = 0; chunk < p.n_chunks; ++chunk) {
        const int cur = chunk & 1;
        __syncthreads();                         // this chunk's pieces have landed; the other buffer is free again
        if (chunk + 1 < p.n_chunks) issue_dma(chunk + 1, cur ^ 1);

        const float* xs = reinterpret_cast<const float*>(lds_x + cur * XBYTES) + (8 * half) * 256;
        const half8* wh = reinterpret_cast<const half8*>(lds_w + cur * WBYTES) + half * 128 + l31;
        const half8* wl = wh + 256;
        half8 bh[WPX], bl[WPX];
#pragma unroll
        for (int j = 0; j < WPX; ++j) {
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[jj] = xs[jj * 256 + pxl[j]];
            if (HAS_PRM) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int c = chunk * 16 + 8 * half + jj;
                    if (c < C) {
                        float4 m = p.prm[(size_t)pn[j] * C + c];
                        float t = (v[jj] - m.x) * m.y + m.z;
                        if (m.w != 0.f) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));
                        v[jj] = t;
                    }
                }
            }
            if (EMIT) {
                if (eok[j]) {
                    half8 eh, el;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int c = chunk * 16 + 8 * half + jj;
                        float t = 0.f;
                        if (c < C) {
                            const float4 m = eprm_sh[c];
                            t = (v[jj] - m.x) * m.y + m.z;
                            if (m.w != 0.f) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));
                        }
                        bad |= !(fabsf(t) <= 65000.f);
                        t = fminf(fmaxf(t, -65000.f), 65000.f);
                        const _Float16 hh = (_Float16)t;
                        eh[jj] = hh;
                        el[jj] = (_Float16)(t - (float)hh);
                    }
                    const size_t eo = (eoff[j] + (size_t)chunk * 2 * HW) * 8;
                    *reinterpret_cast<half8*>(p.ehi + eo) = eh;
                    if (!X1) *reinterpret_cast<half8*>(p.elo + eo) = el;
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
    float* slab = reinterpret_cast<float*>(smem5) + wave * (32 * SROW);
    const int row_l = lane >> 4, c4 = lane & 15;
    const long long g4 = px0 + wave * 64 + 4 * c4;     // HW % 4 == 0 (conv5_supported): the 4 pixels share an image
    const bool ok4 = g4 < p.total_px;
    const int n4 = ok4 ? (int)(g4 / HW) : 0;
    const size_t base4 = (size_t)n4 * p.Cout * HW + (size_t)(g4 - (long long)n4 * HW);
    const int co0 = co_blk * 128;
    const float osc = p.out_scale_dev ? p.out_scale * p.out_scale_dev[0] : p.out_scale;
#pragma unroll
    for (int i = 0; i < WCO; ++i) {
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * half) * SROW + j * 32 + l31] = acc[i][j][r];
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 4 + row_l;
            const int co = co0 + i * 32 + row;
            const float4 a4 = *reinterpret_cast<const float4*>(slab + row * SROW + 4 * c4);
            if (ok4 && co < p.Cout) {
                const float bz = p.bias[co];
                float4 v = make_float4(a4.x * osc + bz, a4.y * osc + bz, a4.z * osc + bz, a4.w * osc + bz);
                const size_t o = base4 + (size_t)co * HW;
                if (p.res) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.res + o);
                    v = make_float4(rr.x + v.x, rr.y + v.y, rr.z + v.z, rr.w + v.w);
                }
                *reinterpret_cast<float4*>(p.out + o) = v;
            }
        }
    }
}

const float* conv_zero_page();

bool conv5_supported(int B, int Cout, int H, int W) {
    const long long px = (long long)B * H * W;
    if ((H * W) % 4) return false;
    // enough workgroups to occupy the chip; smaller problems stay on the split-K fp32 kernel
    return ((px + 255) / 256) * ((Cout + 127) / 128) >= 32;
}

Status launch_conv5(hipStream_t s, const Conv5Args& a) {
    if (!conv5_supported(a.B, a.Cout, a.H, a.W)) return Status{DPIR_ERR_UNSUPPORTED, "conv5: shape not tiled"};
    Conv5K k;
    k.sa = a.src.a; k.sb = a.src.b; k.ca = a.src.ca; k.cb = a.src.cb; k.prm = a.prm;
    k.w16 = reinterpret_cast<const char*>(a.w16); k.bias = a.bias; k.out = a.out; k.res = a.res;
    k.B = a.B; k.Cout = a.Cout; k.HW = a.H * a.W;
    k.n_chunks = (a.src.ca + a.src.cb + 15) / 16;
    k.n_co_blocks = (a.Cout + 127) / 128;
    k.total_px = (long long)a.B * a.H * a.W;
    k.zeros = conv_zero_page();
    if (!k.zeros) return Status{DPIR_ERR_NOMEM, "conv5: cannot allocate the zero page"};
    k.out_scale = 1.0f / a.w16_scale;
    k.out_scale_dev = a.out_scale_dev;
    k.range_ctr = a.range_ctr;
    k.eprm = a.emit_prm; k.ehi = reinterpret_cast<_Float16*>(a.emit_hi); k.elo = reinterpret_cast<_Float16*>(a.emit_lo);
    k.eC8 = 2 * k.n_chunks;
    const unsigned blocks = (unsigned)(((k.total_px + 255) / 256) * k.n_co_blocks);
    if (a.emit_hi) {
        if (a.prm || !a.emit_prm || (!a.x1 && !a.emit_lo)) return invalid("conv5: the plane-emitting variant takes the raw input and a GroupNorm table for the planes");
        if ((a.H * a.W) % 256 || a.src.ca + a.src.cb > 1024) return invalid("conv5: the plane-emitting variant needs H*W % 256 == 0 and at most 1024 input channels");
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
