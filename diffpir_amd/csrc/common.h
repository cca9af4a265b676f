// Shared declarations for the DiffPIR MI355X engine (gfx950 only; no compatibility layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <mutex>

#include "../../include/diffpir_engine.h"

namespace dpir {

// ---------------------------------------------------------------------------------------------
// error plumbing: nothing throws across the C ABI
struct Status {
    int code = DPIR_OK;
    std::string msg;
    bool ok() const { return code == DPIR_OK; }
};

#define DPIR_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            char _b[512];                                                                      \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                     __FILE__, __LINE__);                                                      \
            return dpir::Status{DPIR_ERR_HIP, _b};                                             \
        }                                                                                      \
    } while (0)

#define DPIR_TRY(expr)                           \
    do {                                         \
        dpir::Status _s = (expr);                \
        if (!_s.ok()) return _s;                 \
    } while (0)

inline Status invalid(const std::string& m) { return Status{DPIR_ERR_INVALID, m}; }

// One-time hipFuncSetAttribute(MaxDynamicSharedMemorySize) per kernel AND device.  Engines are driven from several host threads
// (ctypes releases the GIL), so the lazy first-launch initialisation is a std::call_once, one flag per device ordinal.
struct LdsAttrOnce {
    std::once_flag flags[16];
    hipError_t errs[16] = {};
    hipError_t set(const void* fn, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        std::call_once(flags[dev], [&] { errs[dev] = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
        return errs[dev];
    }
};

// ---------------------------------------------------------------------------------------------
// profiling classes (include/diffpir_engine.h)
enum ProfClass { PC_CONV3 = 0, PC_CONV1 = 1, PC_GN = 2, PC_ATTN = 3, PC_FFT = 4, PC_ELEM = 5, PC_UNET = 6, PC_LOOP = 7 };

struct Profiler {
    bool on = false;
    hipStream_t stream = nullptr;
    struct Rec { hipEvent_t a, b; int cls; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double ms[DPIR_PROF_CLASSES] = {0};
    int64_t cnt[DPIR_PROF_CLASSES] = {0};

    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    unsigned long long serial = 0;      // every scoped enqueue bumps it, profiling on or off: "has anything been enqueued since X?" (api.hip replay guard)
    int begin(int cls) {
        ++serial;
        if (!on) return -1;
        Rec r; r.a = get(); r.b = get(); r.cls = cls;
        (void)hipEventRecord(r.a, stream);
        recs.push_back(r);
        return (int)recs.size() - 1;
    }
    void end(int id) {
        if (id < 0) return;
        (void)hipEventRecord(recs[id].b, stream);
    }
    void collect() {
        if (recs.empty()) return;
        (void)hipStreamSynchronize(stream);
        for (auto& r : recs) {
            float t = 0; (void)hipEventElapsedTime(&t, r.a, r.b);
            ms[r.cls] += t; cnt[r.cls] += 1;
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
    }
    void reset() { collect(); for (int i = 0; i < DPIR_PROF_CLASSES; ++i) { ms[i] = 0; cnt[i] = 0; } }
};

struct ProfScope {
    Profiler* p; int id;
    ProfScope(Profiler* p_, int cls) : p(p_), id(p_ ? p_->begin(cls) : -1) {}
    ~ProfScope() { if (p) p->end(id); }
};

// ---------------------------------------------------------------------------------------------
// conv launcher (conv.hip)
struct ConvSrc {
    const float* a = nullptr; int ca = 0;     // first tensor  [B, ca, Hs, Ws]
    const float* b = nullptr; int cb = 0;     // second tensor [B, cb, Hs, Ws] (virtual concat) or null
    int Hs = 0, Ws = 0;                       // source resolution
    int mode = 0;                             // 0 plain, 1 nearest-up x2, 2 avg-pool 2x2
    const float4* prm = nullptr;              // [B, ca+cb] {mean, a, b, act}: v = (v-mean)*a+b, then SiLU if act != 0; null = identity
};
struct ConvArgs {
    ConvSrc src;
    const float* w = nullptr;     // packed [Cin][taps][CoutP]
    const float* bias = nullptr;  // [Cout]
    float* out = nullptr;         // [B, Cout, H, W]
    const float* res = nullptr;   // residual [B, Cout, Hr, Wr] or null
    int res_mode = 0;             // 0 plain, 1 up, 2 down (same meaning as src.mode)
    int B = 0, Cin = 0, Cout = 0, CoutP = 0, H = 0, W = 0;
    int ks = 3;                   // 3 or 1
    float* partial = nullptr;     // split-K slab (optional) and its capacity in floats
    size_t partial_capacity = 0;
    int dbg = 0;                  // ablation bits (debug bench only)
};
Status launch_conv(hipStream_t s, const ConvArgs& a);
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// norm.hip
struct CatSrc { const float* a; int ca; const float* b; int cb; };

// act.hip: the activation pre-pass of the operand-split f16 path (GroupNorm apply / FiLM / SiLU / resampling / concat / split)
// hi / lo: blocked [B][C8][H][W][8] f16 planes, C8 = 2*ceil(C/16); mode: 0 plain, 1 nearest-up source, 2 avg-pool source
// range_ctr: device counter of operand values outside the f16 range (see act.hip range_report), or null
// lo == nullptr: single-product mode, only the hi plane is produced
Status launch_act_split(hipStream_t s, CatSrc src, const float4* prm, int mode, int B, int H, int W, void* hi, void* lo,
                        unsigned long long* range_ctr = nullptr);
// A split-K convolution whose slabs have not been combined yet (launch_conv6 with defer = true): value = sum_k partial[k]
// (slab order) + bias + residual.  Either gn_act_small (the fused low-resolution prologue of the consumer) or
// launch_conv6_resolve finishes it into `out`.
struct PendingConv {
    const float* partial = nullptr; int ksplit = 0; const float* bias = nullptr;
    const float* res = nullptr; int res_mode = 0; float* out = nullptr; int B = 0, Cout = 0, H = 0, W = 0;
    double2* stat_plane = nullptr;      // where the combine kernel would put the per-plane fp64 GroupNorm sums
};
// act.hip: fused low-resolution elementwise chain [split-K combine +] GroupNorm statistics + affine/FiLM fold + SiLU + resample +
// f16 split, one workgroup per (image, group); src tensors at Hs x Ws, planes at the conv's output resolution
struct StepDev;
struct GnActArgs {
    CatSrc src; PendingConv pend;
    const float* gamma = nullptr; const float* beta = nullptr;
    const float* film = nullptr; int film_stride = 0; int film_off = 0; const StepDev* fstep = nullptr; int frows = 0;
    bool silu = true; int mode = 0; int B = 0, Hs = 0, Ws = 0;
    void* hi = nullptr; void* lo = nullptr; unsigned long long* range_ctr = nullptr;
    // gradient mode: what the backward pass reads -- per-(image, channel) {mean, a, b, act} and per-(image, group) {mean, rstd}
    float4* prm_out = nullptr; float2* stats_out = nullptr;
};
bool gn_act_small_supported(int C, int Hs, int Ws, int mode);
Status launch_gn_act_small(hipStream_t s, const GnActArgs& a);
// conv6.hip: 3x3, f16x3 (or f16x1), two workgroups per CU (private weight rings)
struct Conv6Args {
    const void* xhi = nullptr; const void* xlo = nullptr;
    const void* w16 = nullptr; float w16_scale = 1.f;       // pack_weights_conv6 layout
    const float* bias = nullptr; float* out = nullptr; const float* res = nullptr; int res_mode = 0;
    int B = 0, Cin = 0, Cout = 0, H = 0, W = 0;
    float* partial = nullptr; size_t partial_capacity = 0;
    float2* stat = nullptr;        // optional [B][Cout][conv6_stat_slots(H, W)] epilogue partial sums (no split-K)
    double2* stat_plane = nullptr; // optional [B][Cout] fp64 {sum, sum of squares}, written by the split-K combine
    bool x1 = false;               // single-product mode (f16x1): hi planes / hi weight halves only
    const float* out_scale_dev = nullptr;      // optional device scalar folded into the output scale (dgrad, unet_bwd.hip)
    int force_kernel = 0;          // tests only: 6 = conv6 even where conv7 applies, 7 = conv7 or an error; 0 = launch_conv6 decides
    const struct Conv6Emit* emit = nullptr;    // conv6_params.h: fused emission of the NEXT convolution's operand planes instead of `out`
};
// true when a launch with these arguments runs whole-K on the 8 x 32 geometry with every tile inside the image (what Conv6Emit needs)
bool conv7_emit_supported(int B, int Cout, int H, int W);
bool conv6_supported(int H, int W);
int conv6_stat_slots(int H, int W);
// pend_out != null: a split-K launch leaves its slabs uncombined and describes them in *pend_out (stat kind 3); the caller must
// have them finished (gn_act_small or launch_conv6_resolve) before the slab buffer is reused
Status launch_conv6(hipStream_t s, const Conv6Args& a, int* stat_kind_out = nullptr, PendingConv* pend_out = nullptr);
Status launch_conv6_resolve(hipStream_t s, const PendingConv& p);
float pack_weights_conv6(const float* w_oihw, int cout, int cin, std::vector<uint16_t>& out);
// conv5.hip: 1x1 convolution, f16x3 with the operand split done in-kernel from the fp32 NCHW (virtual concat) input
struct Conv5Args {
    CatSrc src; const float4* prm = nullptr;
    const void* w16 = nullptr; float w16_scale = 1.f;
    const float* bias = nullptr; float* out = nullptr; const float* res = nullptr;   // res: same shape as out, or null
    int B = 0, Cout = 0, H = 0, W = 0;
    unsigned long long* range_ctr = nullptr;   // f16 operand range guard (act.hip range_report)
    bool x1 = false;                           // single-product mode (f16x1)
    // optional second product: the split operand planes of the following 3x3 convolution, silu(GroupNorm(src)) per emit_prm,
    // blocked [B][2*ceil(C/16)][HW][8] f16 (act.hip's layout); requires prm == null (the 1x1 itself multiplies the raw input)
    const float4* emit_prm = nullptr; void* emit_hi = nullptr; void* emit_lo = nullptr;
    const float* out_scale_dev = nullptr;      // optional device scalar folded into the output scale (dgrad)
};
constexpr int kConv5EmitMaxC = 384;            // channels of the GroupNorm table the plane-emitting variant stages in LDS
bool conv5_supported(int B, int Cout, int H, int W, bool has_prm = false);
Status launch_conv5(hipStream_t s, const Conv5Args& a);
float pack_weights_f16x3_1x1(const float* w_oi, int cout, int cin, std::vector<uint16_t>& out);
// conv8.hip: GroupNorm affine + SiLU + f16 split + 3x3 convolution to <= 16 output channels in one kernel (the network's output layer)
struct Conv8Args {
    const float* x = nullptr;                  // [B][C][H][W] fp32, single source, same resolution
    const float4* prm = nullptr;               // [B][C] {mean, scale, shift, silu flag} (launch_gn_prm)
    const void* w = nullptr; float w_scale = 1.f;    // pack_weights_conv8 layout
    const float* bias = nullptr; float* out = nullptr;
    int B = 0, C = 0, Cout = 0, H = 0, W = 0;
    unsigned long long* range_ctr = nullptr;
    bool x1 = false;                           // single-product mode (f16x1)
    bool silu = true;                          // the table's activation flag (prm.w); the kernel is built for SiLU
};
bool conv8_supported(int B, int C, int Cout, int H, int W);
Status launch_conv8(hipStream_t s, const Conv8Args& a);
float pack_weights_conv8(const float* w_oihw, int cout, int cin, std::vector<uint16_t>& out);
// part[n*C+c] = fp64 {sum, sum of squares} of one channel plane of the (virtual-concat) input
Status launch_gn_stats(hipStream_t s, CatSrc src, int B, int HW, double2* part);
// statistics of one tensor of a virtual concat: conv6 epilogue slots [B][c][nslots] (float2) or, when slots == null,
// gn_stats partials [B][c] (double2)
struct GnStatSrc { const float2* slots = nullptr; int nslots = 0; const double2* part = nullptr; int c = 0; };
// prm[n*C+c] = {mean, rstd*gamma*(1+scale), beta*(1+scale)+shift, silu?1:0}; film = [B, film_stride] rows with
// scale at film[n*film_stride + film_off + c], shift at +C; film == null -> plain GroupNorm.
// fstep != null: film is the hoisted table [n_steps][frows] of dpir_run_loop; the row of the current step (fstep->i, read on
// the device) is used for every image (film_stride = 0)
Status launch_gn_prm(hipStream_t s, GnStatSrc sa, GnStatSrc sb, int HW, const float* gamma, const float* beta,
                     const float* film, int film_stride, int film_off, int B, int C, bool silu, float4* prm,
                     const StepDev* fstep = nullptr, int frows = 0, float2* stats_out = nullptr);
// time embedding MLP: t_dev [B] int32 -> semb [B, ted] = silu(time_embed(timestep_embedding(t)) + label_emb[y])
Status launch_time_embed(hipStream_t s, const int* t_dev, const int* y_dev, const float* freqs, const float* w0, const float* b0,
                         const float* w2, const float* b2, const float* label_emb, int B, int mc, float* tmp, float* semb);
// out[n, r] = dot(W[r,:], semb[n,:]) + bias[r], W [R, K]
Status launch_rows_gemv(hipStream_t s, const float* W, const float* bias, const float* x, int B, int R, int K, float* out);

// attn.hip: qkv [B, 3C, T] (legacy head-major order) -> out [B, C, T]
Status launch_attention(hipStream_t s, const float* qkv, float* out, int B, int C, int T, int head_ch);

}  // namespace dpir
