// Development-only: one convolution shape timed in isolation (include/diffpir_debug.h).  Part of libdiffpir_dbg.so, which links
// against the product library and uses its internal launchers; nothing here is on the product path.
#include "engine.h"
#include "conv6_params.h"
#include "../../include/diffpir_debug.h"
#include <vector>
using namespace dpir;
static int fail(dpir_engine* e, const Status& s) {
    if (e) e->last_error = s.msg;
    return s.code;
}
#define API_TRY(e, expr)                          \
    do {                                          \
        Status _s = (expr);                       \
        if (!_s.ok()) return fail((e), _s);       \
    } while (0)
#define API_HIP(e, expr)                                                                  \
    do {                                                                                  \
        hipError_t _h = (expr);                                                           \
        if (_h != hipSuccess)                                                             \
            return fail((e), Status{DPIR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_h)}); \
    } while (0)

// elements whose bit patterns differ (out[0]) and their largest absolute difference as ordered uint bits (out[1])
__global__ void dbg_bitdiff_kernel(const float* a, const float* b, size_t n, unsigned long long* out) {
    unsigned long long cnt = 0; unsigned mx = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = a[i], y = b[i];
        if (__builtin_bit_cast(unsigned, x) != __builtin_bit_cast(unsigned, y)) {
            ++cnt;
            const float d = fabsf(x - y);
            const unsigned u = d == d ? __builtin_bit_cast(unsigned, d) : 0x7fc00000u;
            mx = u > mx ? u : mx;
        }
    }
    if (cnt) { atomicAdd(&out[0], cnt); atomicMax(&out[1], (unsigned long long)mx); }
}

extern "C" {
// Times one convolution shape in isolation (synthetic operands already on the device).  mode bits: 0 = exact-fp32 kernels,
// 1 = f16x3 (conv6 for 3x3 incl. its act_split pre-pass, conv5 for 1x1), 2 = f16x3 without the pre-pass (planes prepared once).
// Not part of the product path; used by tools/ only.
int dpir_debug_conv_bench(dpir_engine* e, int B, int Cin, int Cout, int H, int W, int ks, int mode, int with_prm,
                          int dbg, int iters, double* ms_out) {
    if (!e || !ms_out || iters <= 0) return DPIR_ERR_INVALID;
    (void)hipSetDevice(e->device);
    int taps = ks * ks, coutp = round_up(Cout, 64), cinp = round_up(Cin, 16);
    float *x = nullptr, *w = nullptr, *bias = nullptr, *out = nullptr, *partial = nullptr; float4* prm = nullptr;
    API_TRY(e, e->ws.getT("dbg#x", (size_t)B * Cin * H * W, &x));
    API_TRY(e, e->ws.getT("dbg#w", (size_t)cinp * taps * coutp, &w));
    API_TRY(e, e->ws.getT("dbg#b", (size_t)coutp, &bias));
    API_TRY(e, e->ws.getT("dbg#o", (size_t)B * Cout * H * W, &out));
    API_TRY(e, e->ws.getT("dbg#prm", (size_t)B * Cin, &prm));
    API_TRY(e, e->ws.getT("conv#partial", (size_t)16 * 1024 * 1024, &partial));
    API_TRY(e, launch_randn(e->stream, x, 1, 1, 0, 1, (size_t)B * Cin * H * W));
    API_TRY(e, launch_randn(e->stream, w, 2, 1, 0, 1, (size_t)cinp * taps * coutp));
    API_TRY(e, launch_randn(e->stream, bias, 3, 1, 0, 1, (size_t)coutp));
    API_TRY(e, launch_randn(e->stream, reinterpret_cast<float*>(prm), 4, 1, 0, 1, (size_t)B * Cin * 4));
    ConvArgs a;
    a.src.a = x; a.src.ca = Cin; a.src.Hs = H; a.src.Ws = W; a.src.mode = 0; a.src.prm = with_prm ? prm : nullptr;
    a.w = w; a.bias = bias; a.out = out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.CoutP = coutp; a.H = H; a.W = W; a.ks = ks;
    a.partial = partial; a.partial_capacity = (size_t)16 * 1024 * 1024;
    const void* w16 = nullptr; float w16_scale = 1.f;
    if (dbg != 0) {   // operand-split f16 path with synthetic split weights
        std::vector<float> hw((size_t)Cout * Cin * taps);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) % 2001) / 1000.0f * 0.05f - 0.05f;
        std::vector<uint16_t> w16v;
        w16_scale = ks == 1 ? pack_weights_f16x3_1x1(hw.data(), Cout, Cin, w16v) : pack_weights_conv6(hw.data(), Cout, Cin, w16v);
        void* wp = nullptr;
        API_TRY(e, e->ws.get("dbg#w16", w16v.size() * 2, &wp));
        API_HIP(e, hipMemcpy(wp, w16v.data(), w16v.size() * 2, hipMemcpyHostToDevice));
        w16 = wp;
    }
    Conv5Args a5;
    Conv6Args a6;
    const bool use5 = dbg != 0 && ks == 1, use6 = dbg != 0 && ks == 3;
    if (use5) {
        a5.src = CatSrc{x, Cin, nullptr, 0}; a5.prm = a.src.prm; a5.w16 = w16; a5.w16_scale = w16_scale; a5.bias = bias; a5.out = out;
        a5.B = B; a5.Cout = Cout; a5.H = H; a5.W = W; a5.x1 = e->precision == 2;      // f16x1 engine: single-product kernels (hi planes only)
    }
    if (use6) {
        int C8 = 2 * ((Cin + 15) / 16);
        size_t plane = (size_t)B * C8 * H * W * 16;
        char* s16 = nullptr;
        API_TRY(e, e->ws.getT("dbg#s16", 2 * plane, &s16));
        API_TRY(e, launch_act_split(e->stream, CatSrc{x, Cin, nullptr, 0}, a.src.prm, 0, B, H, W, s16, s16 + plane));
        a6.xhi = s16; a6.xlo = s16 + plane; a6.w16 = w16; a6.w16_scale = w16_scale; a6.bias = bias; a6.out = out;
        a6.B = B; a6.Cin = Cin; a6.Cout = Cout; a6.H = H; a6.W = W; a6.partial = partial; a6.partial_capacity = a.partial_capacity;
        a6.x1 = e->precision == 2;
    }
    auto run_once = [&]() -> Status {
        if (use5) return launch_conv5(e->stream, a5);
        if (use6) {
            if (dbg == 1) DPIR_TRY(launch_act_split(e->stream, CatSrc{x, Cin, nullptr, 0}, a.src.prm, 0, B, H, W, const_cast<void*>(a6.xhi), const_cast<void*>(a6.xlo)));
            return launch_conv6(e->stream, a6);
        }
        return launch_conv(e->stream, a);
    };
    API_TRY(e, run_once());
    hipEvent_t e0, e1;
    API_HIP(e, hipEventCreate(&e0)); API_HIP(e, hipEventCreate(&e1));
    API_HIP(e, hipEventRecord(e0, e->stream));
    for (int i = 0; i < iters; ++i) API_TRY(e, run_once());
    API_HIP(e, hipEventRecord(e1, e->stream));
    API_HIP(e, hipEventSynchronize(e1));
    float ms = 0; API_HIP(e, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_out = ms / iters;
    return DPIR_OK;
}

// conv7 (csrc/conv7.hip) against conv6 (8 x 32 geometry: the only one conv6 is still built for) on the same split planes and weight
// pack: outputs + fused GroupNorm statistics (whole K) or the split-K partial slabs must agree bit for bit; then both are timed back to back.  x1: f16x1 planes / products; split: give both kernels a partial buffer so that the
// launch is split along K when launch_conv6's rule says so; scaled: a device output scale of 0.25 (the dgrad route).
int dpir_debug_conv7_check(dpir_engine* e, int B, int Cin, int Cout, int H, int W, int res_mode, int x1, int split, int scaled, int iters,
                            double* ms6_out, double* ms7_out, unsigned long long* mismatches_out, float* maxdiff_out, int* ksplit_out) {
    if (!e || !ms6_out || !ms7_out || !mismatches_out || !maxdiff_out || !ksplit_out || iters <= 0 || res_mode < -1 || res_mode > 2) return DPIR_ERR_INVALID;
    (void)hipSetDevice(e->device);
    if (!conv6_supported(H, W) || W < 32) return fail(e, invalid("conv7 check: conv6 only has the 8 x 32 geometry (W >= 32) to compare with"));
    const size_t nx = (size_t)B * Cin * H * W, no = (size_t)B * Cout * H * W;
    const size_t nres = res_mode == 1 ? no / 4 : (res_mode == 2 ? no * 4 : no);
    const int slots = conv6_stat_slots(H, W);
    const size_t nst = (size_t)B * Cout * slots;
    const size_t pcap = (size_t)16 * no;                       // room for 16 slabs
    float *x = nullptr, *bias = nullptr, *o6 = nullptr, *o7 = nullptr, *res = nullptr, *p6 = nullptr, *p7 = nullptr, *scale = nullptr;
    float2 *st6 = nullptr, *st7 = nullptr;
    unsigned long long* cmp = nullptr;
    API_TRY(e, e->ws.getT("c7#x", nx, &x));
    API_TRY(e, e->ws.getT("c7#b", (size_t)round_up(Cout, 64), &bias));
    API_TRY(e, e->ws.getT("c7#o6", no, &o6));
    API_TRY(e, e->ws.getT("c7#o7", no, &o7));
    API_TRY(e, e->ws.getT("c7#res", nres, &res));
    API_TRY(e, e->ws.getT("c7#st6", nst, &st6));
    API_TRY(e, e->ws.getT("c7#st7", nst, &st7));
    API_TRY(e, e->ws.getT("c7#cmp", (size_t)2, &cmp));
    API_TRY(e, e->ws.getT("c7#scale", (size_t)4, &scale));
    if (split) { API_TRY(e, e->ws.getT("c7#p6", pcap, &p6)); API_TRY(e, e->ws.getT("c7#p7", pcap, &p7)); }
    API_TRY(e, launch_randn(e->stream, x, 11, 1, 0, 1, nx));
    API_TRY(e, launch_randn(e->stream, bias, 12, 1, 0, 1, (size_t)round_up(Cout, 64)));
    API_TRY(e, launch_randn(e->stream, res, 13, 1, 0, 1, nres));
    const float quarter = 0.25f;
    API_HIP(e, hipMemcpyAsync(scale, &quarter, 4, hipMemcpyHostToDevice, e->stream));
    std::vector<float> hw((size_t)Cout * Cin * 9);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) % 2001) / 1000.0f * 0.05f - 0.05f;
    std::vector<uint16_t> w16v;
    const float w16_scale = pack_weights_conv6(hw.data(), Cout, Cin, w16v);
    void* wp = nullptr;
    API_TRY(e, e->ws.get("c7#w16", w16v.size() * 2, &wp));
    API_HIP(e, hipMemcpy(wp, w16v.data(), w16v.size() * 2, hipMemcpyHostToDevice));
    const int chunks = (Cin + 15) / 16, C8 = 2 * chunks;
    const size_t plane = (size_t)B * C8 * H * W * 16;
    char* s16 = nullptr;
    API_TRY(e, e->ws.getT("c7#s16", 2 * plane, &s16));
    API_TRY(e, launch_act_split(e->stream, CatSrc{x, Cin, nullptr, 0}, nullptr, 0, B, H, W, s16, x1 ? nullptr : s16 + plane));
    Conv6Args a6;
    a6.xhi = s16; a6.xlo = x1 ? nullptr : s16 + plane; a6.w16 = wp; a6.w16_scale = w16_scale; a6.bias = bias;
    a6.B = B; a6.Cin = Cin; a6.Cout = Cout; a6.H = H; a6.W = W; a6.x1 = x1 != 0;
    if (res_mode >= 0) { a6.res = res; a6.res_mode = res_mode; }
    if (scaled) a6.out_scale_dev = scale;
    a6.out = o6; a6.stat = st6; a6.force_kernel = 6;
    if (split) { a6.partial = p6; a6.partial_capacity = pcap; }
    // the parameter block launch_conv6 would build, for conv7 (same tiling and the same split rule)
    Conv6K k;
    k.xhi = reinterpret_cast<const char*>(a6.xhi); k.xlo = reinterpret_cast<const char*>(a6.xlo);
    k.w16 = reinterpret_cast<const char*>(wp); k.bias = bias; k.out = o7; k.res = a6.res; k.res_mode = a6.res_mode;
    k.B = B; k.Cout = Cout; k.H = H; k.W = W; k.n_chunks_total = chunks; k.C8 = C8;
    k.out_scale = 1.0f / w16_scale; k.out_scale_dev = a6.out_scale_dev; k.zeros = nullptr;
    const int geo = W >= 32 ? 0 : (W >= 16 ? 1 : 2);
    const int tw = geo == 0 ? 32 : (geo == 1 ? 16 : 8), th = geo == 0 ? 8 : (geo == 1 ? 16 : 8), ti = geo == 2 ? 4 : 1;
    k.tiles_x = (W + tw - 1) / tw; k.tiles_y = (H + th - 1) / th;
    const int n_ptiles = k.tiles_x * k.tiles_y * ((B + ti - 1) / ti);
    k.n_co_blocks = (Cout + 127) / 128;
    const int blocks = n_ptiles * k.n_co_blocks;
    int S = 1;
    if (split && blocks < 384) {
        S = (512 + blocks - 1) / blocks;
        if (S > chunks / 2) S = chunks / 2;
        if (S > 16) S = 16;
        if (S < 1) S = 1;
        if ((size_t)S * no > pcap) S = 1;
    }
    k.chunks_per_split = (chunks + S - 1) / S;
    S = (chunks + k.chunks_per_split - 1) / k.chunks_per_split;
    k.ksplit = S; k.partial = S > 1 ? p7 : nullptr;
    k.stat = S == 1 ? st7 : nullptr; k.stat_slots = S == 1 ? slots : 0;
    *ksplit_out = S;
    API_HIP(e, hipMemsetAsync(o6, 0xFF, no * 4, e->stream));
    API_HIP(e, hipMemsetAsync(o7, 0x7F, no * 4, e->stream));
    API_HIP(e, hipMemsetAsync(st6, 0xFF, nst * 8, e->stream));
    API_HIP(e, hipMemsetAsync(st7, 0x7F, nst * 8, e->stream));
    if (split) { API_HIP(e, hipMemsetAsync(p6, 0xFF, (size_t)S * no * 4, e->stream)); API_HIP(e, hipMemsetAsync(p7, 0x7F, (size_t)S * no * 4, e->stream)); }
    API_HIP(e, hipMemsetAsync(cmp, 0, 16, e->stream));
    int k6 = 0;
    PendingConv pend;
    API_TRY(e, launch_conv6(e->stream, a6, &k6, &pend));
    if ((S > 1) != (k6 == 3) || (S > 1 && pend.ksplit != S)) return fail(e, Status{DPIR_ERR_INVALID, "conv7 check: the split rule of launch_conv6 changed"});
    API_TRY(e, launch_conv7(e->stream, k, blocks * S, x1 != 0));
    API_HIP(e, hipGetLastError());
    if (S > 1) {
        hipLaunchKernelGGL(dbg_bitdiff_kernel, dim3(2048), dim3(256), 0, e->stream, p6, p7, (size_t)S * no, cmp);
    } else {
        hipLaunchKernelGGL(dbg_bitdiff_kernel, dim3(2048), dim3(256), 0, e->stream, o6, o7, no, cmp);
        hipLaunchKernelGGL(dbg_bitdiff_kernel, dim3(256), dim3(256), 0, e->stream, reinterpret_cast<const float*>(st6), reinterpret_cast<const float*>(st7), nst * 2, cmp);
    }
    unsigned long long h[2] = {0, 0};
    API_HIP(e, hipMemcpyAsync(h, cmp, 16, hipMemcpyDeviceToHost, e->stream));
    API_HIP(e, hipStreamSynchronize(e->stream));
    *mismatches_out = h[0];
    const unsigned mb = (unsigned)h[1];
    *maxdiff_out = __builtin_bit_cast(float, mb);
    hipEvent_t e0, e1, e2;
    API_HIP(e, hipEventCreate(&e0)); API_HIP(e, hipEventCreate(&e1)); API_HIP(e, hipEventCreate(&e2));
    for (int i = 0; i < 3; ++i) { API_TRY(e, launch_conv6(e->stream, a6, &k6, &pend)); API_TRY(e, launch_conv7(e->stream, k, blocks * S, x1 != 0)); }
    API_HIP(e, hipEventRecord(e0, e->stream));
    for (int i = 0; i < iters; ++i) API_TRY(e, launch_conv6(e->stream, a6, &k6, &pend));
    API_HIP(e, hipEventRecord(e1, e->stream));
    for (int i = 0; i < iters; ++i) API_TRY(e, launch_conv7(e->stream, k, blocks * S, x1 != 0));
    API_HIP(e, hipEventRecord(e2, e->stream));
    API_HIP(e, hipEventSynchronize(e2));
    float m6 = 0, m7 = 0;
    API_HIP(e, hipEventElapsedTime(&m6, e0, e1)); API_HIP(e, hipEventElapsedTime(&m7, e1, e2));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    *ms6_out = m6 / iters; *ms7_out = m7 / iters;
    return DPIR_OK;
}

int dpir_debug_conv7_emit_supported(dpir_engine* e, int B, int Cout, int H, int W, int* capacity_out) {
    if (!e) return 0;
    (void)hipSetDevice(e->device);
    if (capacity_out) *capacity_out = conv7_emit_capacity();
    return conv7_emit_supported(B, Cout, H, W) ? 1 : 0;
}

}  // extern "C"
