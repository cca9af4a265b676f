// C ABI (include/diffpir_engine.h): lifecycle, memory plumbing, operator entry points and the
// restoration loop with hipGraph capture.  Each entry cites the reference interface it replaces in
// the header; this file only validates, dispatches to the launchers and keeps the error string.
#include "engine.h"
#include "grad.h"
#include "fft4_wave.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

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

static void prox_release(dpir::ProxState* st);

// f16x3 operand range guard: called where the ABI synchronises anyway (dpir_sync, D2H copies).  A non-zero count means
// at least that many wave-lanes clamped an activation to the f16 range since the last check: the images are wrong.
// The failure is STICKY: every later dpir_sync / dpir_d2h / dpir_allgather_results keeps returning DPIR_ERR_RANGE (the device
// results stay wrong) until the next UNet forward or restoration loop starts (range_clear).
// Fused-hop time-out (bit 40 and up of the guard word): degrade, do not die.  Latches the hop off for this engine (the captured step graphs
// hold its kernels: dropped), clears the guard word and says so once on stderr.  Returns true when a time-out was seen.
static bool fuse_timeout_latch(dpir_engine* e, unsigned long long n) {
    if (n < (1ull << 40)) return false;
    e->fuse_h1_off = true;
    e->invalidate_graphs();
    (void)hipMemsetAsync(e->range_ctr, 0, sizeof(unsigned long long), e->stream);
    (void)hipStreamSynchronize(e->stream);
    fprintf(stderr, "diffpir: conv7's fused GroupNorm hop timed out (a workgroup waited too long for the other workgroups of its image: the GPU is shared "
                    "with other engines or processes).  The hop is switched off for this engine (unfused path from here on, ~2 %% slower) and the "
                    "affected work is re-run.\n");
    return true;
}
static int read_range(dpir_engine* e, unsigned long long* n) {
    *n = 0;
    if (hipMemcpyAsync(n, e->range_ctr, sizeof(*n), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess)
        return fail(e, Status{DPIR_ERR_HIP, "reading the operand range counter failed"});
    return DPIR_OK;
}
int dpir_check_range(dpir_engine* e) {
    if (!e->range_ctr || e->precision == 0) { e->fwd_since_sync = 0; e->replay_last = nullptr; return DPIR_OK; }
    unsigned long long n = 0;
    if (int rc = read_range(e, &n)) return rc;
    if (fuse_timeout_latch(e, n)) {
        // the results of the forwards issued since the last synchronisation are invalid.  One forward: re-issue it (its inputs are
        // caller-owned and untouched) on the unfused path; a burst of several cannot be re-issued from here
        const int burst = e->fwd_since_sync;
        std::function<int()> replay = std::move(e->replay_last);
        // work enqueued BEHIND the forward (finalize, a prox step, a copy) has consumed its invalid output: re-issuing the forward alone would leave
        // that work wrong (or overwrite what it wrote in place) -- only a forward that is still the last thing on the stream is replayed
        const bool trailing = e->enqueue_serial() != e->replay_serial;
        e->fwd_since_sync = 0; e->replay_last = nullptr;
        if (burst != 1 || !replay || trailing)
            return fail(e, Status{DPIR_ERR_HIP, "conv7 fused emission timed out " + (burst == 0 ? std::string("inside a loop call (dpir_run_dps_loop)") :
                                                burst == 1 ? std::string("in a forward that other work was already queued behind (or whose buffers were freed)") :
                                                "during a burst of " + std::to_string(burst) + " un-synchronised forwards") + ": the results since the last "
                                                "synchronisation are invalid.  The hop is now off for this engine; re-issue the call(s) (not a sticky error)"});
        if (int rc = replay()) return rc;
        e->fwd_since_sync = 0; e->replay_last = nullptr;
        if (int rc = read_range(e, &n)) return rc;
    }
    e->fwd_since_sync = 0; e->replay_last = nullptr;
    if (n == 0) return DPIR_OK;
    return fail(e, Status{DPIR_ERR_RANGE, std::string(e->precision == 2 ? "f16x1" : "f16x3") + " precision mode: " + std::to_string(n) +
                                          " activation lane(s) exceeded the f16 operand range (|v| > 65000 or NaN) and were clamped since the "
                                          "last forward / loop started -- results are invalid; rerun with precision f32 (engine_precision: f32)"});
}
static int check_range(dpir_engine* e) { return dpir_check_range(e); }
// a new forward / loop starts a new accounting period (stream-ordered)
// Only the range count (bits 0..39) restarts: the fused-hop time-out count above it stays until dpir_check_range has seen it, so a time-out in an
// EARLIER forward of an un-synchronised burst is not wiped by the next forward's clear.
__global__ void range_clear_kernel(unsigned long long* ctr) { *ctr &= ~((1ull << 40) - 1ull); }
static void range_clear(dpir_engine* e) {
    if (e->range_ctr && e->precision != 0) hipLaunchKernelGGL(range_clear_kernel, dim3(1), dim3(1), 0, e->stream, e->range_ctr);
}
static int ilog2u(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

Status dpir_engine::fft_plan(int N, FftPlan* out) {
    auto it = fft_plans.find(N);
    if (it != fft_plans.end()) { *out = it->second; return Status{}; }
    if (N < 2 || (N & (N - 1))) return Status{DPIR_ERR_UNSUPPORTED, "FFT size must be a power of two"};
    std::vector<float2> tw(N / 2);
    for (int k = 0; k < N / 2; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)N;
        tw[k] = make_float2((float)cos(a), (float)sin(a));
    }
    FftPlan p; p.N = N; p.logN = ilog2u(N);
    void* d = nullptr;
    DPIR_HIP(hipMalloc(&d, tw.size() * sizeof(float2)));
    DPIR_HIP(hipMemcpy(d, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    p.tw = reinterpret_cast<float2*>(d);
    fft_plans[N] = p;
    *out = p;
    return Status{};
}

Status dpir_engine::fft2_table(int N, const float2** out) {
    auto it = fft2_tw.find(N);
    if (it != fft2_tw.end()) { *out = it->second; return Status{}; }
    std::vector<float2> tw(N);
    for (int m = 0; m < N; ++m) {
        double a = -2.0 * M_PI * (double)m / (double)N;
        tw[m] = make_float2((float)cos(a), (float)sin(a));
    }
    if (N == 256 || N == 512) { tw.resize(N + wave_tw_count(N)); wave_tw_fill(N, tw.data(), tw.data() + N); }   // fft4_wave.h: per-lane constants behind the table
    void* d = nullptr;
    DPIR_HIP(hipMalloc(&d, tw.size() * sizeof(float2)));
    DPIR_HIP(hipMemcpy(d, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    fft2_tw[N] = reinterpret_cast<float2*>(d);
    *out = fft2_tw[N];
    return Status{};
}

Status dpir_engine::fft2_map(int N, int sf, const Fft2Map** out, bool colmajor) {
    auto key = std::make_pair(N, sf + (colmajor ? 16 : 0));
    auto it = fft2_maps.find(key);
    if (it == fft2_maps.end()) {
        Fft2Map m;
        if (colmajor) fft4_build_map(N, sf, m.h_slot_col, m.h_col_slot);
        else fft2_build_map(N, sf, m.h_slot_col, m.h_col_slot);
        DPIR_HIP(hipMalloc((void**)&m.slot_col, m.h_slot_col.size() * sizeof(int)));
        DPIR_HIP(hipMalloc((void**)&m.col_slot, m.h_col_slot.size() * sizeof(int)));
        DPIR_HIP(hipMemcpy(m.slot_col, m.h_slot_col.data(), m.h_slot_col.size() * sizeof(int), hipMemcpyHostToDevice));
        DPIR_HIP(hipMemcpy(m.col_slot, m.h_col_slot.data(), m.h_col_slot.size() * sizeof(int), hipMemcpyHostToDevice));
        it = fft2_maps.emplace(key, std::move(m)).first;
    }
    *out = &it->second;
    return Status{};
}

Status dpir_engine::resizer(int in_len, int sf, ResizerTab* out) {
    auto key = std::make_pair(in_len, sf);
    auto it = resizers.find(key);
    if (it != resizers.end()) { *out = it->second; return Status{}; }
    if (sf < 1 || in_len % sf) return invalid("Resizer: length not divisible by sf");
    ResizerTab t; t.in_len = in_len; t.out_len = in_len / sf;
    std::vector<float> w; std::vector<int> idx;
    resizer_band(in_len, t.out_len, 1.0 / sf, w, idx, t.taps);
    void *dw = nullptr, *di = nullptr;
    DPIR_HIP(hipMalloc(&dw, w.size() * sizeof(float)));
    DPIR_HIP(hipMalloc(&di, idx.size() * sizeof(int)));
    DPIR_HIP(hipMemcpy(dw, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    DPIR_HIP(hipMemcpy(di, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    t.w = reinterpret_cast<float*>(dw); t.idx = reinterpret_cast<int*>(di);
    resizers[key] = t;
    *out = t;
    return Status{};
}

extern "C" {

int dpir_version(void) { return DPIR_ABI_VERSION; }

int dpir_device_count(int* n_out) {
    if (!n_out) return DPIR_ERR_INVALID;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
    *n_out = count;
    return DPIR_OK;
}

int dpir_device_info(dpir_engine* e, char* buf, size_t cap) {
    if (!e || !buf || cap == 0) return DPIR_ERR_INVALID;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, e->device) != hipSuccess) return fail(e, Status{DPIR_ERR_HIP, "hipGetDeviceProperties failed"});
    char bus[32] = "?";
    (void)hipDeviceGetPCIBusId(bus, sizeof(bus), e->device);
    snprintf(buf, cap, "%s | %s | pci %s | %d CUs | ordinal %d", p.gcnArchName, p.name, bus, p.multiProcessorCount, e->device);
    return DPIR_OK;
}

int dpir_create(int device, dpir_engine** out) {
    if (!out) return DPIR_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return DPIR_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return DPIR_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return DPIR_ERR_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return DPIR_ERR_UNSUPPORTED;   // kernels are built for gfx950 only
    dpir_engine* e = new (std::nothrow) dpir_engine();
    if (!e) return DPIR_ERR_NOMEM;
    e->device = device;
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return DPIR_ERR_HIP; }
    e->prof.stream = e->stream;
    if (hipMalloc((void**)&e->range_ctr, sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(e->range_ctr, 0, sizeof(unsigned long long)) != hipSuccess) { (void)hipStreamDestroy(e->stream); delete e; return DPIR_ERR_NOMEM; }
    e->cus = prop.multiProcessorCount;
    if (const char* ev = getenv("DPIR_PROX_MODE")) { const int m = atoi(ev); if (m == 0 || m == 1) e->prox_mode = m; }
    *out = e;
    return DPIR_OK;
}

void dpir_destroy(dpir_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    unet_free(e);
    e->ws.release();
    for (auto& kv : e->fft_plans) (void)hipFree(kv.second.tw);
    for (auto& kv : e->fft2_tw) (void)hipFree(kv.second);
    for (auto& kv : e->fft2_maps) { (void)hipFree(kv.second.slot_col); (void)hipFree(kv.second.col_slot); }
    for (auto& kv : e->resizers) { (void)hipFree(kv.second.w); (void)hipFree(kv.second.idx); }
    for (void* p : e->user_allocs) (void)hipFree(p);
    e->invalidate_graphs();
    prox_release(&e->loop_prox);
    (void)dpir_comm_destroy(e);
    if (e->range_ctr) (void)hipFree(e->range_ctr);
    (void)hipStreamDestroy(e->stream);
    delete e;
}

const char* dpir_last_error(const dpir_engine* e) { return e ? e->last_error.c_str() : "null engine"; }

int dpir_sync(dpir_engine* e) {
    if (!e) return DPIR_ERR_INVALID;
    API_HIP(e, hipStreamSynchronize(e->stream));
    return check_range(e);
}
void* dpir_stream(dpir_engine* e) { return e ? (void*)e->stream : nullptr; }

int dpir_malloc(dpir_engine* e, size_t bytes, void** dev_out) {
    if (!e || !dev_out) return DPIR_ERR_INVALID;
    (void)hipSetDevice(e->device);
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return fail(e, Status{DPIR_ERR_NOMEM, "dpir_malloc: hipMalloc failed"});
    e->user_allocs.push_back(p);
    *dev_out = p;
    return DPIR_OK;
}
int dpir_free(dpir_engine* e, void* dev) {
    if (!e) return DPIR_ERR_INVALID;
    for (size_t i = 0; i < e->user_allocs.size(); ++i)
        if (e->user_allocs[i] == dev) {
            (void)hipStreamSynchronize(e->stream);
            ++e->copy_serial;                      // a pending forward replay may hold this pointer: it is no longer valid (dpir_check_range)
            (void)hipFree(dev);
            e->user_allocs.erase(e->user_allocs.begin() + i);
            return DPIR_OK;
        }
    return fail(e, invalid("dpir_free: pointer was not allocated by dpir_malloc"));
}
int dpir_h2d(dpir_engine* e, void* d, const void* h, size_t bytes) {
    if (!e) return DPIR_ERR_INVALID;
    ++e->copy_serial;
    API_HIP(e, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, e->stream));
    API_HIP(e, hipStreamSynchronize(e->stream));   // the host buffer may be pageable / short-lived
    return DPIR_OK;
}
int dpir_d2h(dpir_engine* e, void* h, const void* d, size_t bytes) {
    if (!e) return DPIR_ERR_INVALID;
    // the guard word FIRST: a fused-hop time-out re-issues the invalidated forward (dpir_check_range), and the copy must see its result
    if (int rc = check_range(e)) return rc;
    API_HIP(e, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, e->stream));
    API_HIP(e, hipStreamSynchronize(e->stream));
    return DPIR_OK;
}
int dpir_d2d(dpir_engine* e, void* dd, const void* ds, size_t bytes) {
    if (!e) return DPIR_ERR_INVALID;
    ++e->copy_serial;
    API_HIP(e, hipMemcpyAsync(dd, ds, bytes, hipMemcpyDeviceToDevice, e->stream));
    return DPIR_OK;
}

// ------------------------------------------------------------------------------------------ UNet
int dpir_load_unet(dpir_engine* e, const dpir_unet_desc* desc, const dpir_tensor* weights, int n_weights) {
    if (!e || !desc || !weights || n_weights <= 0) return fail(e, invalid("dpir_load_unet: null argument"));
    (void)hipSetDevice(e->device);
    API_HIP(e, hipStreamSynchronize(e->stream));
    e->invalidate_graphs();
    Status s = unet_load(e, desc, weights, n_weights);
    if (!s.ok()) { unet_free(e); return fail(e, s); }
    return DPIR_OK;
}

static Status upload_ints(dpir_engine* e, const char* name, const int64_t* host, int B, int** dev) {
    if (!host) { *dev = nullptr; return Status{}; }
    DPIR_TRY(e->ws.getT(name, (size_t)B, dev));
    std::vector<int> tmp(B);
    for (int i = 0; i < B; ++i) tmp[i] = (int)host[i];
    DPIR_HIP(hipMemcpyAsync(*dev, tmp.data(), B * sizeof(int), hipMemcpyHostToDevice, e->stream));
    DPIR_HIP(hipStreamSynchronize(e->stream));
    return Status{};
}

// One timestep for the whole batch (what model_fn passes): written on the device by a 1-workgroup kernel -- no host buffer, no copy and, unlike
// upload_ints, no hipStreamSynchronize, so back-to-back eager forwards are queued while the previous one still runs.
__global__ void fill_const_kernel(int* p, int v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
static Status fill_ints(dpir_engine* e, const char* name, int value, int B, int** dev) {
    DPIR_TRY(e->ws.getT(name, (size_t)B, dev));
    hipLaunchKernelGGL(fill_const_kernel, dim3((B + 255) / 256), dim3(256), 0, e->stream, *dev, value, B);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

int dpir_set_precision(dpir_engine* e, int mode) {
    if (!e) return DPIR_ERR_INVALID;
    if (mode < 0 || mode > 2) return fail(e, invalid("dpir_set_precision: mode must be 0 (fp32 MFMA), 1 (operand-split f16x3 MFMA) or 2 (f16x1: f16 operands, fp32 accumulate)"));
    if (e->net.loaded && mode != e->precision) return fail(e, Status{DPIR_ERR_STATE, "dpir_set_precision must be called before dpir_load_unet"});
    e->precision = mode;
    return DPIR_OK;
}

int dpir_unet_forward(dpir_engine* e, const float* x, const int64_t* t_host, const int64_t* y_host, float* out, int B, int H, int W) {
    if (!e || !x || !t_host || !out) return fail(e, invalid("dpir_unet_forward: null argument"));
    (void)hipSetDevice(e->device);
    int *t_dev = nullptr, *y_dev = nullptr;
    range_clear(e);
    bool uni = true;
    for (int i = 1; i < B; ++i) uni = uni && t_host[i] == t_host[0];
    if (uni && !y_host) API_TRY(e, fill_ints(e, "api#t", (int)t_host[0], B, &t_dev));
    else API_TRY(e, upload_ints(e, "api#t", t_host, B, &t_dev));
    API_TRY(e, upload_ints(e, "api#y", y_host, B, &y_dev));
    if (y_host && e->net.loaded)
        for (int i = 0; i < B; ++i)
            if (y_host[i] < 0 || y_host[i] >= e->net.desc.num_classes) return fail(e, invalid("class label out of range"));
    API_TRY(e, unet_forward(e, x, t_dev, y_dev, out, B, H, W, nullptr, nullptr, uni));
    if (!e->fuse_h1_off && !e->grad_enabled && e->precision != 0) {
        std::vector<int64_t> tv(t_host, t_host + B), yv;
        if (y_host) yv.assign(y_host, y_host + B);
        if (x != out) e->arm_replay([=]() { return dpir_unet_forward(e, x, tv.data(), yv.empty() ? nullptr : yv.data(), out, B, H, W); });
        else { ++e->fwd_since_sync; e->replay_last = nullptr; }         // in place: the input is gone, nothing to re-issue from
    }
    return DPIR_OK;
}

int dpir_model_fn_xstart(dpir_engine* e, const float* x, int t, float c1, float c2, const int64_t* y_host, float* x0, int B, int H, int W) {
    if (!e || !x || !x0) return fail(e, invalid("dpir_model_fn_xstart: null argument"));
    (void)hipSetDevice(e->device);
    if (!e->net.loaded) return fail(e, Status{DPIR_ERR_STATE, "dpir_load_unet has not been called"});
    int *t_dev = nullptr, *y_dev = nullptr;
    range_clear(e);
    API_TRY(e, fill_ints(e, "api#t", t, B, &t_dev));
    API_TRY(e, upload_ints(e, "api#y", y_host, B, &y_dev));
    float* out6 = nullptr;
    API_TRY(e, e->ws.getT("api#out6", (size_t)B * e->net.desc.out_channels * H * W, &out6));
    API_TRY(e, unet_forward(e, x, t_dev, y_dev, out6, B, H, W, nullptr, nullptr, true));
    ProfScope ps(&e->prof, PC_ELEM);
    API_TRY(e, launch_xstart(e->stream, x, out6, e->net.desc.out_channels, c1, c2, x0, B, H * W));
    if (!e->fuse_h1_off && !e->grad_enabled && e->precision != 0) {
        std::vector<int64_t> yv;
        if (y_host) yv.assign(y_host, y_host + B);
        if (x != x0) e->arm_replay([=]() { return dpir_model_fn_xstart(e, x, t, c1, c2, yv.empty() ? nullptr : yv.data(), x0, B, H, W); });
        else { ++e->fwd_since_sync; e->replay_last = nullptr; }
    }
    return DPIR_OK;
}

int dpir_unet_read_tap(dpir_engine* e, const char* layer, float* host_dst, size_t cap, size_t* numel_out) {
    if (!e || !layer) return fail(e, invalid("dpir_unet_read_tap: null argument"));
    auto it = e->taps.find(layer);
    if (it == e->taps.end()) return fail(e, invalid(std::string("no such tap: ") + layer));
    if (numel_out) *numel_out = it->second.numel;
    if (!host_dst) return DPIR_OK;
    if (cap < it->second.numel) return fail(e, invalid("dpir_unet_read_tap: destination too small"));
    return dpir_d2h(e, host_dst, it->second.p, it->second.numel * sizeof(float));
}

double dpir_unet_flops(dpir_engine* e, int H, int W) { return e ? unet_flops(e->net, H, W) : 0.0; }
double dpir_unet_flops_class(dpir_engine* e, int H, int W, int cls) { return e ? unet_flops(e->net, H, W, cls) : 0.0; }

// ------------------------------------------------------------------------------------------ gradient mode (8f-4)
int dpir_enable_grad(dpir_engine* e, int on) {
    if (!e) return DPIR_ERR_INVALID;
    if (e->net.loaded && (on != 0) != e->grad_enabled) return fail(e, Status{DPIR_ERR_STATE, "dpir_enable_grad must be called before dpir_load_unet"});
    e->grad_enabled = on != 0;
    return DPIR_OK;
}

int dpir_unet_vjp(dpir_engine* e, const float* x, const int64_t* t_host, const int64_t* y_host, const float* gout, float* out, float* dx,
                  int B, int H, int W) {
    if (!e || !x || !t_host || !gout || !dx) return fail(e, invalid("dpir_unet_vjp: null argument"));
    (void)hipSetDevice(e->device);
    if (!e->grad_enabled) return fail(e, Status{DPIR_ERR_STATE, "dpir_unet_vjp: gradient mode is off (dpir_enable_grad before dpir_load_unet)"});
    int *t_dev = nullptr, *y_dev = nullptr;
    range_clear(e);
    API_TRY(e, upload_ints(e, "api#t", t_host, B, &t_dev));
    API_TRY(e, upload_ints(e, "api#y", y_host, B, &y_dev));
    float* o6 = out;
    if (!o6) API_TRY(e, e->ws.getT("api#out6", (size_t)B * e->net.desc.out_channels * H * W, &o6));
    API_TRY(e, unet_forward(e, x, t_dev, y_dev, o6, B, H, W));
    API_TRY(e, unet_backward(e, gout, dx));
    return DPIR_OK;
}

// ------------------------------------------------------------------------------------------ FFT prox
static Status prox_precalc(dpir_engine* e, const float* y, const float* k, int kh, int kw, int sf, int B, int H, int W, ProxState* st) {
    if (sf < 1 || H % sf || W % sf) return invalid("pre_calculate: image size not divisible by sf");
    hipStream_t s = e->stream;
    ProfScope ps(&e->prof, PC_FFT);
    if (st->half) {
        // half-spectrum register-FFT path (fft2.hip): FB = rfft2(p2o-embedded PSF), FBFy = conj(FB) * rfft2(y)
        const float2* tw = nullptr;
        DPIR_TRY(e->fft2_table(W, &tw));
        float* psf = nullptr;
        DPIR_TRY(e->ws.getT("prox#psf", (size_t)B * H * W, &psf));
        SolveArgs none{};
        DPIR_TRY(launch_psf_embed_real(s, k, kh, kw, psf, B, H, W));
        if (st->colmajor) {      // wave-per-transform kernels, column-major spectra (fft4.hip)
            const int NC = st->WP;
            DPIR_TRY(launch_rfft4_rows(s, tw, W, psf, 1.f, 0.f, 1.f, nullptr, st->FB, B, NC, nullptr, 0, st->slot_col));
            DPIR_TRY(launch_cfft4_cols(s, tw, W, st->FB, none, false, B, NC));
            const float* ysrc4 = y;
            if (sf > 1) {
                float* yup = nullptr;
                DPIR_TRY(e->ws.getT("prox#yup", (size_t)B * 3 * H * W, &yup));
                DPIR_TRY(launch_upsample_real(s, y, sf, yup, B * 3, H / sf, W / sf));
                ysrc4 = yup;
            }
            DPIR_TRY(launch_rfft4_rows(s, tw, W, ysrc4, 1.f, 0.f, 1.f, nullptr, st->FBFy, B * 3, NC, nullptr, 0, st->slot_col));
            DPIR_TRY(launch_cfft4_cols(s, tw, W, st->FBFy, none, false, B * 3, NC));
            DPIR_TRY(launch_precalc_finish2(s, st->FB, st->FBFy, st->F2B, B, (size_t)H * NC));
            if (sf > 1) DPIR_TRY(launch_fold_f2b4(s, st->F2B, st->slot_col, W, NC, sf, st->invW, B));
            return Status{};
        }
        DPIR_TRY(launch_rfft_rows(s, tw, psf, 1.f, 0.f, 1.f, nullptr, st->FB, B, W, nullptr, 0, st->slot_col));
        DPIR_TRY(launch_cfft_cols(s, tw, st->FB, none, false, B, H));
        const float* ysrc = y;
        if (sf > 1) {      // F(zero-stuffed y) (utils_sisr.py:84-85)
            float* yup = nullptr;
            DPIR_TRY(e->ws.getT("prox#yup", (size_t)B * 3 * H * W, &yup));
            DPIR_TRY(launch_upsample_real(s, y, sf, yup, B * 3, H / sf, W / sf));
            ysrc = yup;
        }
        DPIR_TRY(launch_rfft_rows(s, tw, ysrc, 1.f, 0.f, 1.f, nullptr, st->FBFy, B * 3, W, nullptr, 0, st->slot_col));
        DPIR_TRY(launch_cfft_cols(s, tw, st->FBFy, none, false, B * 3, H));
        DPIR_TRY(launch_precalc_finish2(s, st->FB, st->FBFy, st->F2B, B, (size_t)H * st->WP));
        if (sf > 1) DPIR_TRY(launch_fold_f2b(s, st->F2B, st->slot_col, H, sf, st->invW, B));
        return Status{};
    }
    FftPlan ph, pw;
    DPIR_TRY(e->fft_plan(H, &ph));
    DPIR_TRY(e->fft_plan(W, &pw));
    DPIR_TRY(launch_psf_embed(s, k, kh, kw, st->FB, B, H, W));
    DPIR_TRY(launch_fft_rows(s, pw, st->FB, nullptr, 1.f, 0.f, B, H, W, false));
    DPIR_TRY(launch_fft_cols(s, ph, st->FB, B, H, W, false));
    DPIR_TRY(launch_upsample_embed(s, y, sf, st->FBFy, B * 3, H / sf, W / sf));
    DPIR_TRY(launch_fft_rows(s, pw, st->FBFy, nullptr, 1.f, 0.f, B * 3, H, W, false));
    DPIR_TRY(launch_fft_cols(s, ph, st->FBFy, B * 3, H, W, false));
    DPIR_TRY(launch_precalc_finish(s, st->FB, st->FBFy, st->F2B, B, H, W));
    return Status{};
}

static Status prox_alloc(dpir_engine* e, int sf, int B, int H, int W, ProxState* st) {
    st->B = B; st->H = H; st->W = W; st->sf = sf;
    st->half = fft2_supported(H, W, sf);
    st->colmajor = e->prox_mode == 1 && fft4_supported(H, W, sf);
    st->WP = st->colmajor ? fft4_columns(W, sf) : (st->half ? fft2_padded_width(W) : W);
    size_t hw = (size_t)H * st->WP;
    if (hipMalloc((void**)&st->FB, B * hw * sizeof(float2)) != hipSuccess ||
        hipMalloc((void**)&st->F2B, B * hw * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&st->FBFy, 3 * B * hw * sizeof(float2)) != hipSuccess)
        return Status{DPIR_ERR_NOMEM, "pre_calculate: hipMalloc failed"};
    st->invW = nullptr; st->slot_col = nullptr; st->col_slot = nullptr; st->h_col_slot = nullptr;
    if (st->half && sf > 1) {
        const dpir_engine::Fft2Map* m = nullptr;
        DPIR_TRY(e->fft2_map(W, sf, &m, st->colmajor));
        st->slot_col = m->slot_col; st->col_slot = m->col_slot; st->h_col_slot = &m->h_col_slot;
        if (hipMalloc((void**)&st->invW, (size_t)B * (H / sf) * (W / sf / 2 + 1) * sizeof(float)) != hipSuccess)
            return Status{DPIR_ERR_NOMEM, "pre_calculate: hipMalloc failed"};
    }
    return Status{};
}
static void prox_release(ProxState* st) {
    if (st->FB) (void)hipFree(st->FB);
    if (st->F2B) (void)hipFree(st->F2B);
    if (st->FBFy) (void)hipFree(st->FBFy);
    if (st->invW) (void)hipFree(st->invW);
    st->FB = nullptr; st->F2B = nullptr; st->FBFy = nullptr; st->invW = nullptr;
}

int dpir_prox_fft_precalc(dpir_engine* e, const float* y, const float* k, int kh, int kw, int sf, int B, int H, int W, dpir_prox** out) {
    if (!e || !y || !k || !out) return fail(e, invalid("dpir_prox_fft_precalc: null argument"));
    (void)hipSetDevice(e->device);
    *out = nullptr;
    dpir_prox* p = new (std::nothrow) dpir_prox();
    if (!p) return fail(e, Status{DPIR_ERR_NOMEM, "out of host memory"});
    Status s = prox_alloc(e, sf, B, H, W, &p->st);
    if (s.ok()) s = prox_precalc(e, y, k, kh, kw, sf, B, H, W, &p->st);
    if (!s.ok()) { prox_release(&p->st); delete p; return fail(e, s); }
    *out = p;
    return DPIR_OK;
}
void dpir_prox_free(dpir_engine* e, dpir_prox* p) {
    if (!p) return;
    if (e) (void)hipStreamSynchronize(e->stream);
    prox_release(&p->st);
    delete p;
}

static unsigned brev(unsigned v, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}

int dpir_prox_read(dpir_engine* e, const dpir_prox* p, int which, void* host_dst, size_t cap_bytes) {
    if (!e || !p || !host_dst) return fail(e, invalid("dpir_prox_read: null argument"));
    const ProxState& st = p->st;
    size_t hw = (size_t)st.H * st.W;
    size_t planes = which == 2 ? (size_t)3 * st.B : (size_t)st.B;
    if (st.half) {
        if (which < 0 || which > 2) return fail(e, invalid("dpir_prox_read: which must be 0, 1 or 2"));
        size_t esz = which == 1 ? sizeof(float) : sizeof(float2);
        if (cap_bytes < planes * hw * esz) return fail(e, invalid("dpir_prox_read: destination too small"));
        size_t shw = (size_t)st.H * st.WP;
        std::vector<char> tmp(planes * shw * esz);
        const void* src = which == 0 ? (const void*)st.FB : (which == 1 ? (const void*)st.F2B : (const void*)st.FBFy);
        int rc = dpir_d2h(e, tmp.data(), src, tmp.size());
        if (rc != DPIR_OK) return rc;
        // natural[u][v] = stored[u][v] for v <= W/2, conj(stored[(H-u)%H][W-v]) beyond (Hermitian spectra of real signals)
        for (size_t pl = 0; pl < planes; ++pl)
            for (int u = 0; u < st.H; ++u)
                for (int v = 0; v < st.W; ++v) {
                    bool mir = v > st.W / 2;
                    int su = mir ? (st.H - u) % st.H : u, sv = mir ? st.W - v : v;
                    if (st.h_col_slot) sv = (*st.h_col_slot)[sv];           // sf > 1: alias-grouped column order
                    const char* sp = tmp.data() + (pl * shw + (st.colmajor ? (size_t)sv * st.H + fft4_row_pos(su) : (size_t)su * st.WP + sv)) * esz;
                    char* dp = reinterpret_cast<char*>(host_dst) + (pl * hw + (size_t)u * st.W + v) * esz;
                    memcpy(dp, sp, esz);
                    if (mir && which != 1) reinterpret_cast<float*>(dp)[1] = -reinterpret_cast<float*>(dp)[1];
                }
        return DPIR_OK;
    }
    size_t esz = which == 1 ? sizeof(float) : sizeof(float2);
    const void* src = which == 0 ? (const void*)st.FB : (which == 1 ? (const void*)st.F2B : (const void*)st.FBFy);
    if (which < 0 || which > 2) return fail(e, invalid("dpir_prox_read: which must be 0, 1 or 2"));
    if (cap_bytes < planes * hw * esz) return fail(e, invalid("dpir_prox_read: destination too small"));
    std::vector<char> tmp(planes * hw * esz);
    int rc = dpir_d2h(e, tmp.data(), src, tmp.size());
    if (rc != DPIR_OK) return rc;
    // stored layout is bit-reversed along both axes (fft.hip): natural[u][v] = stored[brev(u)][brev(v)]
    int lh = ilog2u(st.H), lw = ilog2u(st.W);
    char* dst = reinterpret_cast<char*>(host_dst);
    for (size_t pl = 0; pl < planes; ++pl)
        for (int u = 0; u < st.H; ++u)
            for (int v = 0; v < st.W; ++v)
                memcpy(dst + (pl * hw + (size_t)u * st.W + v) * esz,
                       tmp.data() + (pl * hw + (size_t)brev(u, lh) * st.W + brev(v, lw)) * esz, esz);
    return DPIR_OK;
}

// The half-spectrum prox passes described by `a` (rows forward -> columns with the solve -> rows inverse) on the layout the spectra were built in:
// wave-per-transform kernels on the column-major spectrum (fft4.hip, 256 x 256) or the two-pass register kernels (fft2.hip).
static Status prox_passes(dpir_engine* e, const ProxState& st, const ProxPassArgs& a) {
    hipStream_t s = e->stream;
    const int P = st.B * 3, N = st.W;
    const RenoiseArgs ra{a.rn.xt, a.rn.sp, a.rn.lp, a.rn.n1, a.rn.n2, a.rn.stride, a.rn.with_n1};
    if (st.colmajor) {
        DPIR_TRY(launch_rfft4_rows(s, a.tw, st.W, a.x, a.pa, a.pb, a.pm, a.sp, a.hbuf, P, st.WP, a.fu.eps6, a.fu.out_ch, a.slot_col));
        DPIR_TRY(launch_cfft4_cols(s, a.tw, st.W, a.hbuf, a.solve, true, P, st.WP));
        return launch_irfft4_rows(s, a.tw, st.W, a.hbuf, a.out, a.scale, a.oa, a.ob, a.blend_base, a.g, P, st.WP, a.rn.xt ? &ra : nullptr, a.col_slot);
    }
    DPIR_TRY(launch_rfft_rows(s, a.tw, a.x, a.pa, a.pb, a.pm, a.sp, a.hbuf, P, N, a.fu.eps6, a.fu.out_ch, a.slot_col));
    DPIR_TRY(launch_cfft_cols(s, a.tw, a.hbuf, a.solve, true, P, st.H));
    return launch_irfft_rows(s, a.tw, a.hbuf, a.out, a.scale, a.oa, a.ob, a.blend_base, a.g, P, N, a.rn.xt ? &ra : nullptr, a.col_slot);
}

// out = blend ? base + g*((ifft)*oa+ob - base) : (ifft)*oa+ob ; input pre-map v = (x*pa+pb)*alpha
static Status data_solution_impl(dpir_engine* e, const ProxState& st, const float* x, float pa, float pb, float alpha, float* out,
                                 float oa, float ob, const float* blend_base, float g, const StepDev* sp = nullptr) {
    if (!sp && !(alpha > 0.f)) return invalid("data_solution: alpha must be > 0");
    if (st.half) {
        const float2* tw = nullptr;
        DPIR_TRY(e->fft2_table(st.W, &tw));
        float2* hbuf = nullptr;
        DPIR_TRY(e->ws.getT("prox#hbuf", (size_t)st.B * 3 * st.H * st.WP, &hbuf));
        ProfScope ps2(&e->prof, PC_FFT);
        ProxPassArgs a{};
        a.x = x; a.pa = pa; a.pb = pb; a.pm = alpha; a.sp = sp; a.fu = RowsFuse{nullptr, 0}; a.slot_col = st.slot_col;
        a.solve = SolveArgs{st.FB, st.F2B, st.FBFy, alpha, st.sf, sp, st.invW, st.slot_col};
        a.out = out; a.scale = 1.0f / ((float)st.H * (float)st.W); a.oa = oa; a.ob = ob;
        a.blend_base = (blend_base && g != 1.0f) ? blend_base : nullptr; a.g = g;
        a.rn = RenoiseFuse{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}; a.col_slot = st.col_slot;
        a.hbuf = hbuf; a.tw = tw;
        return prox_passes(e, st, a);
    }
    FftPlan ph, pw;
    DPIR_TRY(e->fft_plan(st.H, &ph));
    DPIR_TRY(e->fft_plan(st.W, &pw));
    float2* buf = nullptr;
    DPIR_TRY(e->ws.getT("prox#buf", (size_t)st.B * 3 * st.H * st.W, &buf));
    hipStream_t s = e->stream;
    ProfScope ps(&e->prof, PC_FFT);
    DPIR_TRY(launch_fft_rows_real3(s, pw, buf, x, pa, pb, alpha, st.B * 3, st.H, st.W, sp));
    SolveArgs a{st.FB, st.F2B, st.FBFy, alpha, st.sf, sp};
    DPIR_TRY(launch_fft_cols_solve(s, ph, buf, a, st.B, st.H, st.W));
    float scale = 1.0f / ((float)st.H * (float)st.W);
    DPIR_TRY(launch_ifft_rows_real(s, pw, buf, out, scale, oa, ob, blend_base, g, st.B * 3, st.H, st.W));
    return Status{};
}

int dpir_set_prox_launch(dpir_engine* e, int mode) {
    if (!e || (mode != 0 && mode != 1)) return fail(e, invalid("dpir_set_prox_launch: mode must be 0 or 1"));
    (void)hipSetDevice(e->device);
    if (mode != e->prox_mode) { (void)hipStreamSynchronize(e->stream); e->invalidate_graphs(); }
    e->prox_mode = mode;
    return DPIR_OK;
}
int dpir_data_solution(dpir_engine* e, const dpir_prox* p, const float* x, float alpha, float* out) {
    if (!e || !p || !x || !out) return fail(e, invalid("dpir_data_solution: null argument"));
    (void)hipSetDevice(e->device);
    API_TRY(e, data_solution_impl(e, p->st, x, 1.f, 0.f, alpha, out, 1.f, 0.f, nullptr, 0.f));
    return DPIR_OK;
}
int dpir_prox_fft_apply(dpir_engine* e, const dpir_prox* p, float* x0, float tau, float guidance) {
    if (!e || !p || !x0) return fail(e, invalid("dpir_prox_fft_apply: null argument"));
    (void)hipSetDevice(e->device);
    API_TRY(e, data_solution_impl(e, p->st, x0, 0.5f, 0.5f, tau, x0, 2.f, -1.f, x0, guidance));
    return DPIR_OK;
}

// measurement: n back-to-back applies between two events on the engine stream, eagerly or as ONE captured graph (what the restoration
// loop replays: no host launch cost, no per-apply event records) -> device microseconds per apply, launch boundaries included
int dpir_prox_fft_apply_timed(dpir_engine* e, const dpir_prox* p, float* x0, float tau, float guidance, int n, int use_graph, float* us_per_apply) {
    if (!e || !p || !x0 || !us_per_apply || n < 1) return fail(e, invalid("dpir_prox_fft_apply_timed: bad argument"));
    (void)hipSetDevice(e->device);
    API_TRY(e, data_solution_impl(e, p->st, x0, 0.5f, 0.5f, tau, x0, 2.f, -1.f, x0, guidance));       // allocates the workspace, warms the code
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    API_HIP(e, hipEventCreate(&ev0));
    API_HIP(e, hipEventCreate(&ev1));
    hipGraphExec_t exec = nullptr;
    Status st;
    const bool prof_on = e->prof.on;
    e->prof.on = false;
    if (use_graph) {
        hipGraph_t graph = nullptr;
        e->ws.frozen = true;
        hipError_t herr = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal);
        if (herr != hipSuccess) st = Status{DPIR_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(herr)};
        else {
            for (int i = 0; i < n && st.ok(); ++i) st = data_solution_impl(e, p->st, x0, 0.5f, 0.5f, tau, x0, 2.f, -1.f, x0, guidance);
            herr = hipStreamEndCapture(e->stream, &graph);
            if (st.ok() && herr != hipSuccess) st = Status{DPIR_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(herr)};
        }
        e->ws.frozen = false;
        if (st.ok() && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) st = Status{DPIR_ERR_HIP, "hipGraphInstantiate failed"};
        if (graph) (void)hipGraphDestroy(graph);
        if (st.ok() && hipGraphLaunch(exec, e->stream) != hipSuccess) st = Status{DPIR_ERR_HIP, "hipGraphLaunch failed"};      // warm-up replay
    }
    if (st.ok()) {
        (void)hipEventRecord(ev0, e->stream);
        if (use_graph) { if (hipGraphLaunch(exec, e->stream) != hipSuccess) st = Status{DPIR_ERR_HIP, "hipGraphLaunch failed"}; }
        else for (int i = 0; i < n && st.ok(); ++i) st = data_solution_impl(e, p->st, x0, 0.5f, 0.5f, tau, x0, 2.f, -1.f, x0, guidance);
        (void)hipEventRecord(ev1, e->stream);
        if (hipEventSynchronize(ev1) != hipSuccess) st = Status{DPIR_ERR_HIP, "hipEventSynchronize failed"};
        float ms = 0.f;
        if (st.ok() && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) *us_per_apply = ms * 1e3f / (float)n;
    }
    e->prof.on = prof_on;
    if (exec) (void)hipGraphExecDestroy(exec);
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
    API_TRY(e, st);
    return DPIR_OK;
}

int dpir_prox_mask(dpir_engine* e, float* x0, const float* y, const uint8_t* mask, float tau, float guidance, int B, int H, int W) {
    if (!e || !x0 || !y || !mask) return fail(e, invalid("dpir_prox_mask: null argument"));
    ProfScope ps(&e->prof, PC_ELEM);
    API_TRY(e, launch_prox_mask(e->stream, x0, y, mask, tau, guidance, (size_t)B * 3 * H * W));
    return DPIR_OK;
}

static Status resize_down_impl(dpir_engine* e, const float* x, float pa, float pb, float* out, int sf, int B, int H, int W) {
    ResizerTab th, tw;
    DPIR_TRY(e->resizer(H, sf, &th));
    DPIR_TRY(e->resizer(W, sf, &tw));
    float* mid = nullptr;
    DPIR_TRY(e->ws.getT("resize#mid", (size_t)B * 3 * (H / sf) * W, &mid));
    ProfScope ps(&e->prof, PC_ELEM);
    // dim 2 (H) first, then dim 3 (W): utils_resizer.py:29-30 (stable argsort of equal scale factors)
    DPIR_TRY(launch_band_resample(e->stream, x, th.w, th.idx, th.taps, B * 3, H, H / sf, W, pa, pb, mid));
    DPIR_TRY(launch_band_resample(e->stream, mid, tw.w, tw.idx, tw.taps, B * 3 * (H / sf), W, W / sf, 1, 1.f, 0.f, out));
    return Status{};
}

int dpir_resize_down(dpir_engine* e, const float* x, float* out, int sf, int B, int H, int W) {
    if (!e || !x || !out) return fail(e, invalid("dpir_resize_down: null argument"));
    (void)hipSetDevice(e->device);
    API_TRY(e, resize_down_impl(e, x, 1.f, 0.f, out, sf, B, H, W));
    return DPIR_OK;
}

static Status prox_ibp_impl(dpir_engine* e, float* x0, const float* y, float rho, float gamma, int in_iter, int sf, int B, int H, int W,
                            const StepDev* sp = nullptr, const LoopDev* lp = nullptr) {
    float* d = nullptr;
    DPIR_TRY(e->ws.getT("ibp#down", (size_t)B * 3 * (H / sf) * (W / sf), &d));
    for (int it = 0; it < in_iter; ++it) {
        DPIR_TRY(resize_down_impl(e, x0, 0.5f, 0.5f, d, sf, B, H, W));      // down(x0/2+.5)
        ProfScope ps(&e->prof, PC_ELEM);
        DPIR_TRY(launch_ibp_update(e->stream, x0, y, d, gamma, rho, sf, B * 3, H, W, sp, lp));
    }
    return Status{};
}
int dpir_prox_ibp(dpir_engine* e, float* x0, const float* y, float rho, float gamma, int in_iter, int sf, int B, int H, int W) {
    if (!e || !x0 || !y) return fail(e, invalid("dpir_prox_ibp: null argument"));
    (void)hipSetDevice(e->device);
    API_TRY(e, prox_ibp_impl(e, x0, y, rho, gamma, in_iter, sf, B, H, W));
    return DPIR_OK;
}

int dpir_bicubic_up(dpir_engine* e, const float* y, float* out, int sf, int B, int h, int w) {
    if (!e || !y || !out) return fail(e, invalid("dpir_bicubic_up: null argument"));
    ProfScope ps(&e->prof, PC_ELEM);
    API_TRY(e, launch_bicubic_up(e->stream, y, out, B * 3, h, w, sf));
    return DPIR_OK;
}

// ------------------------------------------------------------------------------------------ loop arithmetic
// First-order data step of the DiffPIR loop (sub_1_analytic: false, main_ddpir.py:420-430; task sr):
//   x0 <- x0 - d|| (2y - 1) - Resizer(x0) || / dx0 * ||.|| / rho       -- the gradient needs no network backward
static Status prox_first_order_impl(dpir_engine* e, float* x0, const float* y, float rho, int sf, int B, int H, int W,
                                    const StepDev* sp = nullptr, const LoopDev* lp = nullptr) {
    const int h = H / sf, w = W / sf;
    const size_t total = (size_t)B * 3 * H * W, small = (size_t)B * 3 * h * w;
    float *down = nullptr, *diff = nullptr, *gmid = nullptr, *gup = nullptr, *normv = nullptr; double* part = nullptr;
    DPIR_TRY(e->ws.getT("fo#down", small, &down));
    DPIR_TRY(e->ws.getT("fo#diff", small, &diff));
    DPIR_TRY(e->ws.getT("fo#gmid", (size_t)B * 3 * h * W, &gmid));
    DPIR_TRY(e->ws.getT("fo#gup", total, &gup));
    DPIR_TRY(e->ws.getT("fo#part", (size_t)256, &part));
    DPIR_TRY(e->ws.getT("fo#norm", (size_t)4, &normv));
    ResizerTab th, tw;
    DPIR_TRY(e->resizer(H, sf, &th));
    DPIR_TRY(e->resizer(W, sf, &tw));
    DPIR_TRY(resize_down_impl(e, x0, 1.f, 0.f, down, sf, B, H, W));
    ProfScope ps(&e->prof, PC_ELEM);
    hipStream_t s = e->stream;
    DPIR_TRY(launch_diff_norm(s, y, 2.f, -1.f, down, diff, small, part, 256, normv, 1.f, 0.f, nullptr, lp));
    DPIR_TRY(launch_band_resample_T(s, diff, tw.w, tw.idx, tw.taps, B * 3 * h, W, w, 1, 1.f, gmid));
    DPIR_TRY(launch_band_resample_T(s, gmid, th.w, th.idx, th.taps, B * 3, H, h, W, 1.f, gup));
    DPIR_TRY(launch_grad_step(s, x0, gup, normv, 1.f, rho, 1.f, x0, total, sp));
    return Status{};
}

static RenoiseCoef coef_of(const dpir_step& s) { return RenoiseCoef{s.sa_t, s.s1m_t, s.sa_p, s.k1, s.q, s.es, s.k2}; }

int dpir_renoise(dpir_engine* e, float* x, const float* x0, const dpir_step* s, const float* n1, const float* n2, int B, int H, int W) {
    if (!e || !x || !x0 || !s || !n2) return fail(e, invalid("dpir_renoise: null argument"));
    if (s->es != 0.f && !n1) return fail(e, invalid("dpir_renoise: eta_sigma != 0 needs n1"));
    ProfScope ps(&e->prof, PC_ELEM);
    API_TRY(e, launch_renoise(e->stream, x, x0, coef_of(*s), s->es != 0.f ? n1 : nullptr, n2, (size_t)B * 3 * H * W));
    return DPIR_OK;
}
int dpir_repaint_mix(dpir_engine* e, float* x, const float* y, const uint8_t* mask, const dpir_step* s, const float* n, int B, int H, int W) {
    if (!e || !x || !y || !mask || !s || !n) return fail(e, invalid("dpir_repaint_mix: null argument"));
    ProfScope ps(&e->prof, PC_ELEM);
    API_TRY(e, launch_repaint_mix(e->stream, x, y, mask, n, s->sa_t, s->s1m_t, (size_t)B * 3 * H * W));
    return DPIR_OK;
}
int dpir_finalize(dpir_engine* e, const float* x, float* of, uint8_t* ou, int B, int H, int W) {
    if (!e || !x) return fail(e, invalid("dpir_finalize: null argument"));
    ProfScope ps(&e->prof, PC_ELEM);
    API_TRY(e, launch_finalize(e->stream, x, of, ou, B, H * W));
    return DPIR_OK;
}
int dpir_randn(dpir_engine* e, float* out, uint64_t seed, uint64_t stream_id, int64_t image_offset, int B, int C, int H, int W) {
    if (!e || !out) return fail(e, invalid("dpir_randn: null argument"));
    ProfScope ps(&e->prof, PC_ELEM);
    API_TRY(e, launch_randn(e->stream, out, seed, stream_id, image_offset, B, (size_t)C * H * W));
    return DPIR_OK;
}

int dpir_ewise(dpir_engine* e, int op, const float* x_dev, const float* y_dev, size_t y_numel, float scalar, float* out_dev, size_t numel) {
    if (!e || !x_dev || !out_dev) return fail(e, invalid("dpir_ewise: null argument"));
    (void)hipSetDevice(e->device);
    ProfScope ps(&e->prof, PC_ELEM);
    API_TRY(e, launch_ewise(e->stream, op, x_dev, y_dev, y_numel, scalar, out_dev, numel));
    return DPIR_OK;
}

// ------------------------------------------------------------------------------------------ degradation + metrics
int dpir_degrade(dpir_engine* e, const dpir_degrade_desc* d, const uint8_t* gt, const float* k, const uint8_t* mask, const float* noise, float* y) {
    if (!e || !d || !gt || !y) return fail(e, invalid("dpir_degrade: null argument"));
    (void)hipSetDevice(e->device);
    const int B = d->B, H = d->H, W = d->W, sf = d->sf < 1 ? 1 : d->sf;
    if (B <= 0 || H <= 0 || W <= 0 || H % sf || W % sf) return fail(e, invalid("dpir_degrade: bad shape"));
    hipStream_t s = e->stream;
    ProfScope ps(&e->prof, PC_ELEM);
    const int h = H / sf, w = W / sf;
    if (d->task == DPIR_TASK_DEBLUR) {
        if (!k) return fail(e, invalid("dpir_degrade: deblurring needs a PSF"));
        API_TRY(e, launch_blur_wrap_u8(s, gt, k, d->kh, d->kw, B, H, W, y));
    } else if (d->task == DPIR_TASK_INPAINT) {
        if (!mask) return fail(e, invalid("dpir_degrade: inpainting needs a mask"));      // img_H * mask / 255 is formed in the finish kernel (float64)
    } else if (d->task == DPIR_TASK_SR_BLUR || d->task == DPIR_TASK_SR_CUBIC) {
        float* full = nullptr;
        API_TRY(e, e->ws.getT("degrade#full", (size_t)B * 3 * H * W, &full));
        API_TRY(e, launch_u8_to_single(s, gt, nullptr, B, H * W, full));
        API_TRY(e, resize_down_impl(e, full, 1.f, 0.f, y, sf, B, H, W));
    } else return fail(e, invalid("dpir_degrade: unknown task"));
    const size_t total = (size_t)B * 3 * h * w;
    const float* nz = noise;
    if (!nz && d->noise_level_img != 0.f) {
        float* nb = nullptr;
        API_TRY(e, e->ws.getT("degrade#noise", total, &nb));
        // Philox stream 2^40: the loop's streams are draw + 4 * step (draw 0..3), so no (seed, image) pair can meet this one
        API_TRY(e, launch_randn(s, nb, d->seed, 1ull << 40, d->image_offset, B, (size_t)3 * h * w));
        nz = nb;
    }
    const bool inp = d->task == DPIR_TASK_INPAINT;
    API_TRY(e, launch_degrade_finish(s, y, d->noise_level_img != 0.f ? nz : nullptr, (double)d->noise_level_img * 2.0,
                                     inp ? mask : nullptr, inp ? gt : nullptr, H * W, total));
    return DPIR_OK;
}

int dpir_metrics(dpir_engine* e, const float* x0, const uint8_t* gt, int B, int H, int W, float* psnr_host, float* psnr_y_host) {
    if (!e || !x0 || !gt || !psnr_host || B <= 0) return fail(e, invalid("dpir_metrics: null argument"));
    (void)hipSetDevice(e->device);
    double2* acc = nullptr;
    API_TRY(e, e->ws.getT("metrics#acc", (size_t)B, &acc));
    {
        ProfScope ps(&e->prof, PC_ELEM);
        API_TRY(e, launch_metrics(e->stream, x0, gt, B, H * W, acc));
    }
    std::vector<double2> h(B);
    int rc = dpir_d2h(e, h.data(), acc, sizeof(double2) * B);
    if (rc != DPIR_OK) return rc;
    const float cnt = (float)(3.0 * H * W);
    for (int i = 0; i < B; ++i) {
        // utils_image.calculate_psnr_batch in float32: inf when mse == 0, else 20*log10(max_pixel / sqrt(mse + eps))
        const float mse = (float)(h[i].x / (double)cnt), mse_y = (float)(h[i].y / (double)cnt);
        psnr_host[i] = mse == 0.f ? INFINITY : 20.0f * log10f(2.0f / sqrtf(mse + 1e-10f));
        if (psnr_y_host) psnr_y_host[i] = mse_y == 0.f ? INFINITY : 20.0f * log10f(2.0f / sqrtf(mse_y + 1e-10f));
    }
    return DPIR_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ whole loop
namespace {
struct LoopBufs { float *x, *x0, *out6, *n1, *n2, *init_src; int *t_dev, *y_dev; StepDev *steps_dev, *cur; LoopDev* lp;
                  float* film; };   // film: hoisted FiLM table [n_steps][film_rows] (class-unconditional models) or null

__global__ void fill_t_kernel(int* p, const StepDev* sp, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = sp->t;
}

// init (main_ddpir.py:291-320): x_T from y, spectra of the batch
Status loop_init(dpir_engine* e, const dpir_loop_desc& d, const LoopBufs& b, ProxState* prox) {
    hipStream_t s = e->stream;
    const int B = d.B, H = d.H, W = d.W;
    const size_t total = (size_t)B * 3 * H * W;
    const float* src = d.y_dev;
    if (d.task == DPIR_TASK_SR_BLUR || d.task == DPIR_TASK_SR_CUBIC) {
        ProfScope ps(&e->prof, PC_ELEM);
        DPIR_TRY(launch_bicubic_up(s, d.y_dev, b.init_src, B * 3, H / d.sf, W / d.sf, d.sf));   // main_ddpir.py:295
        src = b.init_src;
    }
    const float* n0 = d.noise_init_dev;
    if (!n0) { DPIR_TRY(launch_randn(s, b.n2, d.seed, 0, d.image_offset, B, (size_t)3 * H * W)); n0 = b.n2; }
    {
        ProfScope ps(&e->prof, PC_ELEM);
        DPIR_TRY(launch_init_x(s, src, d.task == DPIR_TASK_INPAINT ? d.mask_dev : nullptr, n0, d.sa_start, d.s1m_start, b.x, total));
    }
    if (d.task == DPIR_TASK_DEBLUR || d.task == DPIR_TASK_SR_BLUR)
        DPIR_TRY(prox_precalc(e, d.y_dev, d.k_dev, d.kh, d.kw, d.sf, B, H, W, prox));
    return Status{};
}

// one iteration of main_ddpir.py:341-470; every per-step scalar is read on the device from b.cur
Status loop_step(dpir_engine* e, const dpir_loop_desc& d, const LoopBufs& b, ProxState* prox, bool last, bool with_n1) {
    hipStream_t s = e->stream;
    const int B = d.B, H = d.H, W = d.W;
    const size_t total = (size_t)B * 3 * H * W;
    if (d.generate_mode == 1) {          // repaint: re-draw the known region at the current noise level (main_ddpir.py:355-358)
        ProfScope ps(&e->prof, PC_ELEM);
        const float* nr = d.noise_rp_dev;
        size_t rstride = total;
        if (!nr) { DPIR_TRY(launch_randn(s, b.n1, d.seed, 3, d.image_offset, B, (size_t)3 * H * W, b.cur, b.lp)); nr = b.n1; rstride = 0; }
        DPIR_TRY(launch_repaint_mix(s, b.x, d.y_dev, d.mask_dev, nr, 0.f, 0.f, total, b.cur, rstride, b.lp));
    }
    if (!b.film) hipLaunchKernelGGL(fill_t_kernel, dim3((B + 255) / 256), dim3(256), 0, s, b.t_dev, b.cur, B);
    DPIR_TRY(unet_forward(e, b.x, b.t_dev, b.y_dev, b.out6, B, H, W, b.film, b.film ? b.cur : nullptr));
    // FFT data step on the half-spectrum path, fused into three launches: eps -> clamped x0 in the row-FFT prologue, spectral
    // solve between the column FFTs, re-noise (+ Philox) in the inverse row-FFT epilogue.  x0 is never materialised.
    if (!last && d.generate_mode == 0 && !d.first_order && (d.task == DPIR_TASK_DEBLUR || d.task == DPIR_TASK_SR_BLUR) && prox->half && d.guidance == 1.0f) {
        if (with_n1 && d.noise_n1_dev && !d.noise_n2_dev) return invalid("host n1 noise requires host n2 noise");
        const float2* tw = nullptr;
        DPIR_TRY(e->fft2_table(prox->W, &tw));
        float2* hbuf = nullptr;
        DPIR_TRY(e->ws.getT("prox#hbuf", (size_t)prox->B * 3 * prox->H * prox->WP, &hbuf));
        ProfScope ps(&e->prof, PC_FFT);
        ProxPassArgs a{};
        a.x = b.x; a.pa = 0.5f; a.pb = 0.5f; a.pm = 1.f; a.sp = b.cur; a.fu = RowsFuse{b.out6, e->net.desc.out_channels}; a.slot_col = prox->slot_col;
        a.solve = SolveArgs{prox->FB, prox->F2B, prox->FBFy, 1.f, prox->sf, b.cur, prox->invW, prox->slot_col};
        a.out = b.x0; a.scale = 1.0f / ((float)H * (float)W); a.oa = 2.f; a.ob = -1.f; a.blend_base = nullptr; a.g = 1.f;
        a.rn = RenoiseFuse{b.x, b.cur, b.lp, d.noise_n1_dev, d.noise_n2_dev, d.noise_n2_dev ? total : 0, with_n1 ? 1 : 0}; a.col_slot = prox->col_slot;
        a.hbuf = hbuf; a.tw = tw;
        DPIR_TRY(prox_passes(e, *prox, a));
        DPIR_HIP(hipGetLastError());
        return Status{};
    }
    {
        ProfScope ps(&e->prof, PC_ELEM);
        DPIR_TRY(launch_xstart(s, b.x, b.out6, e->net.desc.out_channels, 0.f, 0.f, b.x0, B, H * W, b.cur));
    }
    if (last) return Status{};
    if (d.generate_mode != 0) {
        // repaint / vanilla: no data-fidelity step (main_ddpir.py:385 is DiffPIR only)
    } else if (d.first_order) {
        DPIR_TRY(prox_first_order_impl(e, b.x0, d.y_dev, 0.f, d.sf, B, H, W, b.cur, b.lp));
    } else if (d.task == DPIR_TASK_INPAINT) {
        ProfScope ps(&e->prof, PC_ELEM);
        DPIR_TRY(launch_prox_mask(s, b.x0, d.y_dev, d.mask_dev, 0.f, d.guidance, total, b.cur, b.lp));
    } else if (d.task == DPIR_TASK_SR_CUBIC) {
        DPIR_TRY(prox_ibp_impl(e, b.x0, d.y_dev, 0.f, d.gamma, d.in_iter, d.sf, B, H, W, b.cur, b.lp));
    } else {
        DPIR_TRY(data_solution_impl(e, *prox, b.x0, 0.5f, 0.5f, 1.f, b.x0, 2.f, -1.f, b.x0, d.guidance, b.cur));
    }
    const float *n1 = nullptr, *n2 = nullptr;
    size_t stride = 0;
    ProfScope ps(&e->prof, PC_ELEM);
    if (with_n1) {
        if (d.noise_n1_dev) n1 = d.noise_n1_dev;
        else { DPIR_TRY(launch_randn(s, b.n1, d.seed, 1, d.image_offset, B, (size_t)3 * H * W, b.cur, b.lp)); n1 = b.n1; }
    }
    if (d.noise_n2_dev) { n2 = d.noise_n2_dev; stride = total; }
    else { DPIR_TRY(launch_randn(s, b.n2, d.seed, 2, d.image_offset, B, (size_t)3 * H * W, b.cur, b.lp)); n2 = b.n2; }
    if (with_n1 && d.noise_n1_dev && !d.noise_n2_dev) return invalid("host n1 noise requires host n2 noise");
    DPIR_TRY(launch_renoise(s, b.x, b.x0, RenoiseCoef{}, n1, n2, total, b.cur, stride, b.lp));
    DPIR_HIP(hipGetLastError());
    return Status{};
}

Status capture_step(dpir_engine* e, const dpir_loop_desc& d, const LoopBufs& b, ProxState* prox, bool last, bool with_n1,
                    hipGraphExec_t* out) {
    bool prof_on = e->prof.on, taps_on = e->collect_taps;
    e->prof.on = false; e->collect_taps = false; e->ws.frozen = true;
    hipGraph_t graph = nullptr;
    Status cs;
    hipError_t herr = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal);
    if (herr != hipSuccess) cs = Status{DPIR_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(herr)};
    else {
        cs = loop_step(e, d, b, prox, last, with_n1);
        herr = hipStreamEndCapture(e->stream, &graph);
        if (cs.ok() && herr != hipSuccess) cs = Status{DPIR_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(herr)};
    }
    e->ws.frozen = false; e->prof.on = prof_on; e->collect_taps = taps_on;
    if (!cs.ok()) { if (graph) (void)hipGraphDestroy(graph); return cs; }
    herr = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (herr != hipSuccess) return Status{DPIR_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(herr)};
    return Status{};
}
}  // namespace

extern "C" {

static int run_loop_once(dpir_engine* e, const dpir_loop_desc* dd, const dpir_step* steps, int n_steps, float* out_f32, uint8_t* out_u8);

int dpir_run_loop(dpir_engine* e, const dpir_loop_desc* dd, const dpir_step* steps, int n_steps, float* out_f32, uint8_t* out_u8) {
    int rc = run_loop_once(e, dd, steps, n_steps, out_f32, out_u8);
    static const bool fuse_env_off = getenv("DPIR_FUSE_H1") && atoi(getenv("DPIR_FUSE_H1")) == 0;       // unet.hip: the hop is not used at all
    if (rc != DPIR_OK || !e->range_ctr || e->precision == 0 || e->fuse_h1_off || fuse_env_off || e->grad_enabled) return rc;
    // the fused hop may have run: look at the guard word now (the caller synchronises right after the loop anyway) and, on a time-out,
    // run the whole loop again on the unfused path -- same inputs, same noise (device Philox is keyed by seed; host noise buffers are the caller's)
    unsigned long long n = 0;
    if (int r2 = read_range(e, &n)) return r2;
    e->fwd_since_sync = 0; e->replay_last = nullptr;
    if (!fuse_timeout_latch(e, n)) return DPIR_OK;      // plain range excursions stay in the counter for dpir_sync / dpir_d2h to report
    return run_loop_once(e, dd, steps, n_steps, out_f32, out_u8);
}

static int run_loop_once(dpir_engine* e, const dpir_loop_desc* dd, const dpir_step* steps, int n_steps, float* out_f32, uint8_t* out_u8) {
    if (!e || !dd || !steps || n_steps <= 0) return fail(e, invalid("dpir_run_loop: null argument"));
    (void)hipSetDevice(e->device);
    const dpir_loop_desc& d = *dd;
    if (!e->net.loaded) return fail(e, Status{DPIR_ERR_STATE, "dpir_load_unet has not been called"});
    if (d.B <= 0 || d.H <= 0 || d.W <= 0 || d.sf < 1 || d.H % d.sf || d.W % d.sf) return fail(e, invalid("dpir_run_loop: bad shape"));
    if (!d.y_dev) return fail(e, invalid("dpir_run_loop: y is required"));
    if ((d.task == DPIR_TASK_DEBLUR || d.task == DPIR_TASK_SR_BLUR) && !d.k_dev) return fail(e, invalid("dpir_run_loop: task needs a PSF"));
    if (d.task == DPIR_TASK_INPAINT && !d.mask_dev) return fail(e, invalid("dpir_run_loop: inpainting needs a mask"));
    if (d.task < 0 || d.task > 3) return fail(e, invalid("dpir_run_loop: unknown task"));
    if (d.generate_mode < 0 || d.generate_mode > 2) return fail(e, invalid("dpir_run_loop: generate_mode must be 0 (DiffPIR), 1 (repaint) or 2 (vanilla)"));
    if (d.generate_mode != 0 && d.task != DPIR_TASK_INPAINT)
        return fail(e, invalid("dpir_run_loop: repaint / vanilla are inpainting modes (main_ddpir.py:448 re-noises only for inpainting or DiffPIR)"));
    if ((e->net.desc.num_classes > 0) != (d.labels_host != nullptr)) return fail(e, invalid("labels iff class-conditional model"));
    if (d.first_order && (d.generate_mode != 0 || (d.task != DPIR_TASK_SR_BLUR && d.task != DPIR_TASK_SR_CUBIC)))
        return fail(e, Status{DPIR_ERR_UNSUPPORTED, "the first-order data step (sub_1_analytic: false) is implemented for the DiffPIR mode of the "
                                                    "super-resolution tasks (the reference's deblurring operator raises at main_ddpir.py:302)"});
    const int B = d.B, H = d.H, W = d.W;
    const size_t total = (size_t)B * 3 * H * W;
    range_clear(e);
    LoopBufs b{};
    API_TRY(e, e->ws.getT("loop#x", total, &b.x));
    API_TRY(e, e->ws.getT("loop#x0", total, &b.x0));
    API_TRY(e, e->ws.getT("loop#out6", (size_t)B * e->net.desc.out_channels * H * W, &b.out6));
    API_TRY(e, e->ws.getT("loop#n1", total, &b.n1));
    API_TRY(e, e->ws.getT("loop#n2", total, &b.n2));
    API_TRY(e, e->ws.getT("loop#init", total, &b.init_src));
    API_TRY(e, e->ws.getT("loop#t", (size_t)B, &b.t_dev));
    API_TRY(e, e->ws.getT("loop#steps", (size_t)n_steps, &b.steps_dev));
    API_TRY(e, e->ws.getT("loop#cur", (size_t)1, &b.cur));
    API_TRY(e, e->ws.getT("loop#lp", (size_t)1, &b.lp));
    API_TRY(e, upload_ints(e, "loop#y", d.labels_host, B, &b.y_dev));
    const bool hoist_film = e->net.desc.num_classes == 0;
    if (hoist_film) API_TRY(e, e->ws.getT("loop#film", (size_t)n_steps * e->net.film_rows, &b.film));
    bool need_prox = d.task == DPIR_TASK_DEBLUR || d.task == DPIR_TASK_SR_BLUR;
    ProxState& prox = e->loop_prox;
    // sf decides the spectrum layout (half-spectrum register FFT vs bit-reversed c2c): a change of sf re-allocates too
    if (need_prox && (prox.B != B || prox.H != H || prox.W != W || prox.sf != d.sf || prox.half != fft2_supported(H, W, d.sf) ||
                      prox.colmajor != (e->prox_mode == 1 && fft4_supported(H, W, d.sf)) || !prox.FB)) {
        API_HIP(e, hipStreamSynchronize(e->stream));
        prox_release(&prox);
        e->invalidate_graphs();
        API_TRY(e, prox_alloc(e, d.sf, B, H, W, &prox));
    }

    // per-step scalar table and the per-batch device block -> device (two small H2D copies per batch)
    bool with_n1 = false;
    {
        std::vector<StepDev> hs(n_steps);
        for (int i = 0; i < n_steps; ++i) {
            const dpir_step& st = steps[i];
            hs[i] = StepDev{st.t, st.last, i, 0, st.c1, st.c2, st.tau, st.sa_t, st.s1m_t, st.sa_p, st.k1, st.q, st.es, st.k2};
            if (!st.last && st.es != 0.f) with_n1 = true;
            // the reference marks EVERY step with seq[i] == seq[-1] as final (two of them for quad skipping with
            // iter_num > T/2): each is a dead denoiser call, prox and re-noise are skipped (main_ddpir.py:384, 448)
            if (i > 0 && steps[i - 1].last && !st.last) return fail(e, invalid("dpir_run_loop: a non-final step may not follow a final step"));
        }
        LoopDev hl{d.y_dev, d.mask_dev, d.noise_n1_dev, d.noise_n2_dev, d.noise_rp_dev, (unsigned long long)d.seed, (long long)d.image_offset};
        API_HIP(e, hipMemcpyAsync(b.steps_dev, hs.data(), sizeof(StepDev) * n_steps, hipMemcpyHostToDevice, e->stream));
        API_HIP(e, hipMemcpyAsync(b.lp, &hl, sizeof(hl), hipMemcpyHostToDevice, e->stream));
        API_HIP(e, hipStreamSynchronize(e->stream));
    }
    if (with_n1 && d.noise_n2_dev && !d.noise_n1_dev) return fail(e, invalid("dpir_run_loop: eta != 0 with host noise needs noise_n1_dev"));

    if (b.film) {   // time embedding + every ResBlock's FiLM projection for all steps, once per batch (batch-uniform timestep)
        std::vector<int64_t> ts(n_steps);
        for (int i = 0; i < n_steps; ++i) ts[i] = steps[i].t;
        int* ts_dev = nullptr;
        API_TRY(e, upload_ints(e, "loop#ts", ts.data(), n_steps, &ts_dev));
        API_TRY(e, unet_film_table(e, ts_dev, n_steps, b.film));
    }
    API_TRY(e, loop_init(e, d, b, &prox));
    hipGraphExec_t g_step = nullptr, g_last = nullptr;
    for (int i = 0; i < n_steps; ++i) {
        const bool last = steps[i].last != 0;
        if (last && d.skip_dead_final_eval) continue;
        API_HIP(e, hipMemcpyAsync(b.cur, b.steps_dev + i, sizeof(StepDev), hipMemcpyDeviceToDevice, e->stream));
        if (!d.use_graph) {
            API_TRY(e, loop_step(e, d, b, &prox, last, with_n1));
            continue;
        }
        hipGraphExec_t& g = last ? g_last : g_step;
        if (!g) {
            auto make_key = [&]() {
                dpir_engine::GraphKey k{};
                k.task = d.task; k.B = B; k.H = H; k.W = W; k.sf = d.sf; k.in_iter = d.in_iter; k.generate_mode = d.generate_mode;
                k.kind = (last ? 1 : 0) | (with_n1 ? 2 : 0) | (d.first_order ? 4 : 0);
                k.host_n1 = d.noise_n1_dev != nullptr; k.host_n2 = d.noise_n2_dev != nullptr; k.host_rp = d.noise_rp_dev != nullptr;
                k.has_labels = d.labels_host != nullptr;
                k.gamma = d.gamma; k.guidance = d.guidance; k.ws_generation = e->ws.generation;
                return k;
            };
            auto find = [&](const dpir_engine::GraphKey& k) -> dpir_engine::GraphEntry* {
                for (auto& ge : e->graphs) if (memcmp(&ge.key, &k, sizeof(k)) == 0) return &ge;
                return nullptr;
            };
            dpir_engine::GraphEntry* hit = find(make_key());
            if (!hit) {
                // warm-up: run this step eagerly once so that every workspace buffer exists before capture
                // (it is a real step of the loop: its result is kept and the graph is used from the next one)
                API_TRY(e, loop_step(e, d, b, &prox, last, with_n1));
                API_HIP(e, hipStreamSynchronize(e->stream));
                hipGraphExec_t exec = nullptr;
                API_TRY(e, capture_step(e, d, b, &prox, last, with_n1, &exec));
                if (e->graphs.size() >= dpir_engine::kMaxGraphs) {        // evict the least recently used graph
                    size_t lru = 0;
                    for (size_t q = 1; q < e->graphs.size(); ++q) if (e->graphs[q].last_use < e->graphs[lru].last_use) lru = q;
                    // never the partner graph of this very loop
                    if (e->graphs[lru].exec == g_step || e->graphs[lru].exec == g_last) lru = (lru + 1) % e->graphs.size();
                    (void)hipGraphExecDestroy(e->graphs[lru].exec);
                    e->graphs.erase(e->graphs.begin() + lru);
                }
                dpir_engine::GraphEntry ge; ge.key = make_key(); ge.exec = exec; ge.last_use = ++e->graph_clock;   // key: the settled generation
                e->graphs.push_back(ge);
                g = exec;
                continue;   // this step was executed eagerly
            }
            hit->last_use = ++e->graph_clock;
            g = hit->exec;
        }
        ProfScope ps(&e->prof, PC_LOOP);
        API_HIP(e, hipGraphLaunch(g, e->stream));
    }
    {
        ProfScope ps(&e->prof, PC_ELEM);
        API_TRY(e, launch_finalize(e->stream, b.x, out_f32, out_u8, B, H * W));
    }
    return DPIR_OK;
}

// ------------------------------------------------------------------------------------------ gradient-mode plugs + DPS loop (8f-4)
// The batch-wide residual norm from the per-workgroup partial sums.  With a communicator of more than one rank the squared sums are
// all-reduced first: DPS_y0's update x = xt - d norm / d x has no `* norm` factor, so the GLOBAL batch norm of the reference
// (torch.linalg.norm over the whole batch, utils_model.py:392) must be used or the result would depend on the sharding.
static Status dps_norm(dpir_engine* e, const double* part, int nparts, float* normv) {
    double* ssq = nullptr;
    DPIR_TRY(e->ws.getT("dps#ssq", (size_t)2, &ssq));
    DPIR_TRY(launch_norm_fold_ssq(e->stream, part, nparts, ssq));
    if (e->comm && e->comm_world > 1) DPIR_TRY(comm_allreduce_sum_f64(e, ssq, 1));
    return launch_norm_sqrt(e->stream, ssq, normv);
}

// model_fn(..., 'pred_x_prev_and_start') = one denoiser call + p_sample / ddim_sample(eta = 0) (utils_model.py:219-243).  In gradient mode the
// forward leaves its tape and the clamp mask behind for dps_grad_through_network.
static Status p_sample_impl(dpir_engine* e, const float* x, const int* t_dev, const int* y_dev, const PSampleCoef& cf, const float* noise, float* out6,
                            float* xt, float* x0, int B, int H, int W) {
    const int oc = e->net.desc.out_channels;
    if (oc != 6) return Status{DPIR_ERR_UNSUPPORTED, "p_sample needs a learn_sigma model (out_channels == 6): the learned-range variance is read from channels 3..5"};
    unsigned char* inside = nullptr;
    DPIR_TRY(e->ws.getT("dps#inside", (size_t)B * 3 * H * W, &inside));
    DPIR_TRY(unet_forward(e, x, t_dev, y_dev, out6, B, H, W, nullptr, nullptr, true));      // every caller passes one timestep for the whole batch
    ProfScope ps(&e->prof, PC_ELEM);
    DPIR_TRY(launch_psample(e->stream, x, out6, oc, noise, cf, x0, xt, inside, B, H * W));
    e->ps_c1 = cf.c1; e->ps_c2 = cf.c2; e->ps_B = B; e->ps_H = H; e->ps_W = W; e->ps_x0 = x0; e->ps_serial = e->fwd_serial;
    return Status{};
}

// d || m - Resizer(x_hat) || / d x_hat, un-normalised: diff = m - Resizer(x_hat) (kept), the norm (device float), gup = Resizer^T diff.
// measurement = sa (ma y + mb) + s1m noise.
static Status dps_residual(dpir_engine* e, const float* x_hat, const float* y, float ma, float mb, float sa, float s1m, const float* noise, int sf,
                           int B, int H, int W, float** gup_out, float** norm_out) {
    const int h = H / sf, w = W / sf;
    const size_t total = (size_t)B * 3 * H * W, small = (size_t)B * 3 * h * w;
    hipStream_t s = e->stream;
    float *down = nullptr, *diff = nullptr, *gmid = nullptr, *gup = nullptr, *normv = nullptr; double* part = nullptr;
    DPIR_TRY(e->ws.getT("dps#down", small, &down));
    DPIR_TRY(e->ws.getT("dps#diff", small, &diff));
    DPIR_TRY(e->ws.getT("dps#gmid", (size_t)B * 3 * h * W, &gmid));
    DPIR_TRY(e->ws.getT("dps#gup", total, &gup));
    DPIR_TRY(e->ws.getT("dps#part", (size_t)256, &part));
    DPIR_TRY(e->ws.getT("dps#norm", (size_t)4, &normv));
    ResizerTab th, tw;
    DPIR_TRY(e->resizer(H, sf, &th));
    DPIR_TRY(e->resizer(W, sf, &tw));
    DPIR_TRY(resize_down_impl(e, x_hat, 1.f, 0.f, down, sf, B, H, W));
    DPIR_TRY(launch_diff_norm(s, y, ma, mb, down, diff, small, part, 256, nullptr, sa, s1m, noise));
    DPIR_TRY(dps_norm(e, part, 256, normv));
    // Resizer^T: the forward resamples dim 2 (H) first, then dim 3 (W) -> adjoint W first, then H
    DPIR_TRY(launch_band_resample_T(s, diff, tw.w, tw.idx, tw.taps, B * 3 * h, W, w, 1, 1.f, gmid));
    DPIR_TRY(launch_band_resample_T(s, gmid, th.w, th.idx, th.taps, B * 3, H, h, W, 1.f, gup));
    *gup_out = gup; *norm_out = normv;
    return Status{};
}

// torch.autograd.grad(norm, x) with x_hat = x0 of the last p_sample_impl(x): the clamp mask, the direct c1 term and the network backward.
// Leaves direct / dx_net in the workspace (norm_grad = direct + dx_net).
static Status dps_grad_through_network(dpir_engine* e, const float* gup, const float* normv, float** direct_out, float** dxn_out) {
    const int B = e->ps_B, H = e->ps_H, W = e->ps_W, oc = e->net.desc.out_channels;
    const size_t total = (size_t)B * 3 * H * W;
    float *dout6 = nullptr, *direct = nullptr, *dxn = nullptr; unsigned char* inside = nullptr;
    DPIR_TRY(e->ws.getT("dps#dout6", (size_t)B * oc * H * W, &dout6));
    DPIR_TRY(e->ws.getT("dps#direct", total, &direct));
    DPIR_TRY(e->ws.getT("dps#dxn", total, &dxn));
    DPIR_TRY(e->ws.getT("dps#inside", total, &inside));
    DPIR_TRY(launch_dps_seed(e->stream, gup, normv, inside, e->ps_c1, e->ps_c2, oc, dout6, direct, B, H * W));
    DPIR_TRY(unet_backward(e, dout6, dxn));
    *direct_out = direct; *dxn_out = dxn;
    return Status{};
}

int dpir_p_sample(dpir_engine* e, const float* x_dev, int t, const dpir_psample_coef* c, const float* noise_dev, const int64_t* y_host,
                  float* xt_out_dev, float* x0_out_dev, int B, int H, int W) {
    if (!e || !x_dev || !c || !noise_dev || !xt_out_dev || !x0_out_dev || B <= 0 || H <= 0 || W <= 0) return fail(e, invalid("dpir_p_sample: bad argument"));
    (void)hipSetDevice(e->device);
    if (!e->net.loaded) return fail(e, Status{DPIR_ERR_STATE, "dpir_load_unet has not been called"});
    if ((e->net.desc.num_classes > 0) != (y_host != nullptr)) return fail(e, invalid("labels iff class-conditional model"));
    range_clear(e);
    int *t_dev = nullptr, *y_dev = nullptr;
    float* out6 = nullptr;
    API_TRY(e, fill_ints(e, "api#t", t, B, &t_dev));
    API_TRY(e, upload_ints(e, "api#y", y_host, B, &y_dev));
    API_TRY(e, e->ws.getT("loop#out6", (size_t)B * e->net.desc.out_channels * H * W, &out6));
    PSampleCoef cf{c->c1, c->c2, c->pc1, c->pc2, c->min_log, c->max_log, t != 0 ? 1.0f : 0.0f, c->ddim, c->sa_prev, c->s1m_prev};
    API_TRY(e, p_sample_impl(e, x_dev, t_dev, y_dev, cf, noise_dev, out6, xt_out_dev, x0_out_dev, B, H, W));
    if (!e->fuse_h1_off && !e->grad_enabled && e->precision != 0) {
        std::vector<int64_t> yv;
        if (y_host) yv.assign(y_host, y_host + B);
        const dpir_psample_coef cc = *c;
        if (x_dev != xt_out_dev && x_dev != x0_out_dev)
            e->arm_replay([=]() { return dpir_p_sample(e, x_dev, t, &cc, noise_dev, yv.empty() ? nullptr : yv.data(), xt_out_dev, x0_out_dev, B, H, W); });
        else { ++e->fwd_since_sync; e->replay_last = nullptr; }
    }
    return DPIR_OK;
}

int dpir_eps_from_xstart(dpir_engine* e, const float* x_dev, const float* x0_dev, float sqrt_ac, float sqrt_1m_ac, int score, float* out_dev, size_t numel) {
    if (!e || !x_dev || !x0_dev || !out_dev) return fail(e, invalid("dpir_eps_from_xstart: null argument"));
    (void)hipSetDevice(e->device);
    ProfScope ps(&e->prof, PC_ELEM);
    API_TRY(e, launch_eps_from_xstart(e->stream, x_dev, x0_dev, sqrt_ac, sqrt_1m_ac, score, out_dev, numel));
    return DPIR_OK;
}

int dpir_grad_and_value(dpir_engine* e, int through_network, const float* x_hat_dev, const float* measurement_dev, int sf, float* norm_grad_out_dev,
                        float* norm_out_dev, int B, int H, int W) {
    if (!e || !x_hat_dev || !measurement_dev || !norm_grad_out_dev || B <= 0 || sf < 1 || H % sf || W % sf)
        return fail(e, invalid("dpir_grad_and_value: bad argument"));
    (void)hipSetDevice(e->device);
    float *gup = nullptr, *normv = nullptr;
    const size_t total = (size_t)B * 3 * H * W;
    if (through_network) {
        if (!e->grad_enabled) return fail(e, Status{DPIR_ERR_STATE, "grad_and_value through the denoiser needs gradient mode: dpir_enable_grad before dpir_load_unet"});
        if (!e->tape.valid || e->tape.serial != e->ps_serial || e->ps_serial != e->fwd_serial || e->ps_x0 != x_hat_dev || e->ps_B != B || e->ps_H != H || e->ps_W != W)
            return fail(e, Status{DPIR_ERR_STATE, "grad_and_value(x, x_hat): x_hat must be the pred_xstart output of the LAST dpir_p_sample call on this engine "
                                                  "(the tape of that forward is what the gradient runs through)"});
    }
    ProfScope ps(&e->prof, PC_ELEM);
    API_TRY(e, dps_residual(e, x_hat_dev, measurement_dev, 1.f, 0.f, 1.f, 0.f, nullptr, sf, B, H, W, &gup, &normv));
    if (through_network) {
        float *direct = nullptr, *dxn = nullptr;
        API_TRY(e, dps_grad_through_network(e, gup, normv, &direct, &dxn));
        API_TRY(e, launch_dps_update(e->stream, nullptr, direct, dxn, 0.f, nullptr, norm_grad_out_dev, total));
    } else {
        // norm_grad = -gup / norm: grad_step with src = 0, lam * nv / rho * tail = 1  ->  dst = 0 - ng  ... evaluated as a plain scale instead
        API_TRY(e, launch_neg_scale_by_norm(e->stream, gup, normv, norm_grad_out_dev, total));
    }
    if (norm_out_dev) API_HIP(e, hipMemcpyAsync(norm_out_dev, normv, sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    return DPIR_OK;
}

int dpir_run_dps_loop(dpir_engine* e, const dpir_loop_desc* dd, const dpir_step* steps, const dpir_dps_coef* coefs, int n_steps,
                      int variant, float lambda_, const float* noise_ps_dev, const float* noise_yt_dev, float step_scale, float* out_f32,
                      uint8_t* out_u8) {
    if (!e || !dd || !steps || !coefs || n_steps <= 0) return fail(e, invalid("dpir_run_dps_loop: null argument"));
    if (variant != 0 && variant != 1) return fail(e, invalid("dpir_run_dps_loop: variant must be 0 (DPS_y0) or 1 (DPS_yt)"));
    (void)hipSetDevice(e->device);
    const dpir_loop_desc& d = *dd;
    if (!e->net.loaded) return fail(e, Status{DPIR_ERR_STATE, "dpir_load_unet has not been called"});
    if (variant == 0 && !e->grad_enabled) return fail(e, Status{DPIR_ERR_STATE, "DPS_y0 needs gradient mode: dpir_enable_grad before dpir_load_unet"});
    if (d.task != DPIR_TASK_SR_BLUR && d.task != DPIR_TASK_SR_CUBIC)
        return fail(e, Status{DPIR_ERR_UNSUPPORTED, "DPS_y0 is implemented for the super-resolution tasks (the reference's deblurring operator raises at "
                                                    "main_ddpir.py:302 and its inpainting branch never defines xt)"});
    if (d.B <= 0 || d.H <= 0 || d.W <= 0 || d.sf < 1 || d.H % d.sf || d.W % d.sf || !d.y_dev) return fail(e, invalid("dpir_run_dps_loop: bad shape"));
    if ((e->net.desc.num_classes > 0) != (d.labels_host != nullptr)) return fail(e, invalid("labels iff class-conditional model"));
    const int B = d.B, H = d.H, W = d.W, sf = d.sf, h = H / sf, w = W / sf, oc = e->net.desc.out_channels;
    const size_t total = (size_t)B * 3 * H * W, small = (size_t)B * 3 * h * w;
    hipStream_t s = e->stream;
    range_clear(e);
    LoopBufs b{};
    float *xprev = nullptr, *diff = nullptr;
    API_TRY(e, e->ws.getT("loop#x", total, &b.x));
    API_TRY(e, e->ws.getT("loop#x0", total, &b.x0));
    API_TRY(e, e->ws.getT("loop#out6", (size_t)B * oc * H * W, &b.out6));
    API_TRY(e, e->ws.getT("loop#n2", total, &b.n2));
    API_TRY(e, e->ws.getT("loop#init", total, &b.init_src));
    API_TRY(e, e->ws.getT("loop#t", (size_t)B, &b.t_dev));
    API_TRY(e, e->ws.getT("dps#xprev", total, &xprev));
    API_TRY(e, e->ws.getT("dps#diff", small, &diff));
    API_TRY(e, upload_ints(e, "loop#y", d.labels_host, B, &b.y_dev));
    // init (main_ddpir.py:293-315): bicubic up-sampling of y, forward noising to t_start
    {
        ProfScope ps(&e->prof, PC_ELEM);
        API_TRY(e, launch_bicubic_up(s, d.y_dev, b.init_src, B * 3, h, w, sf));
        const float* n0 = d.noise_init_dev;
        if (!n0) { API_TRY(e, launch_randn(s, b.n2, d.seed, 0, d.image_offset, B, (size_t)3 * H * W)); n0 = b.n2; }
        API_TRY(e, launch_init_x(s, b.init_src, nullptr, n0, d.sa_start, d.s1m_start, b.x, total));
    }
    for (int i = 0; i < n_steps; ++i) {
        const dpir_step& st = steps[i];
        if (st.last && d.skip_dead_final_eval) continue;
        std::vector<int64_t> tv(B, st.t);
        int* t_dev = nullptr;
        API_TRY(e, upload_ints(e, "loop#tt", tv.data(), B, &t_dev));
        if (st.last) {                                            // the final denoiser call is dead (main_ddpir.py:384, 470)
            API_TRY(e, unet_forward(e, b.x, t_dev, b.y_dev, b.out6, B, H, W));
            continue;
        }
        const float* nz = noise_ps_dev ? noise_ps_dev + (size_t)i * total : nullptr;
        if (!nz) { ProfScope ps(&e->prof, PC_ELEM); API_TRY(e, launch_randn(s, b.n2, d.seed, (uint64_t)4 * (i + 1), d.image_offset, B, (size_t)3 * H * W)); nz = b.n2; }
        PSampleCoef cf{st.c1, st.c2, coefs[i].pc1, coefs[i].pc2, coefs[i].min_log, coefs[i].max_log, st.t != 0 ? 1.0f : 0.0f,
                       d.ddim_sample, coefs[i].sa_prev, coefs[i].s1m_prev};
        API_TRY(e, p_sample_impl(e, b.x, t_dev, b.y_dev, cf, nz, b.out6, xprev, b.x0, B, H, W));
        ProfScope ps(&e->prof, PC_ELEM);
        float *gup = nullptr, *normv = nullptr;
        if (variant == 1) {
            // DPS_yt (main_ddpir.py:439-445): the measurement is noised to level t, the residual is taken at xt = p_sample's sample and
            // differentiated w.r.t. xt itself -- no backward through the network.  The norm is multiplied back in (:444); it is still the
            // whole batch's (all-reduced in dps_norm when a communicator is attached), so the last bit does not depend on the sharding
            const float* ny = noise_yt_dev ? noise_yt_dev + (size_t)i * small : nullptr;
            if (!ny) { API_TRY(e, launch_randn(s, diff, d.seed, (uint64_t)4 * (i + 1) + 1, d.image_offset, B, (size_t)3 * h * w)); ny = diff; }
            float* nyb = nullptr;
            API_TRY(e, e->ws.getT("dps#ny", small, &nyb));
            API_HIP(e, hipMemcpyAsync(nyb, ny, small * sizeof(float), hipMemcpyDeviceToDevice, s));
            API_TRY(e, dps_residual(e, xprev, d.y_dev, 2.f, -1.f, st.sa_t, st.s1m_t, nyb, sf, B, H, W, &gup, &normv));
            API_TRY(e, launch_grad_step(s, xprev, gup, normv, lambda_, st.tau, 0.35f, b.x, total));
            continue;
        }
        // difference = (2y - 1) - Resizer(x0), norm over the whole batch (utils_model.py:391-392)
        float *direct = nullptr, *dxn = nullptr;
        API_TRY(e, dps_residual(e, b.x0, d.y_dev, 2.f, -1.f, 1.f, 0.f, nullptr, sf, B, H, W, &gup, &normv));
        API_TRY(e, dps_grad_through_network(e, gup, normv, &direct, &dxn));
        API_TRY(e, launch_dps_update(s, xprev, direct, dxn, step_scale, b.x, nullptr, total));
    }
    {
        ProfScope ps(&e->prof, PC_ELEM);
        API_TRY(e, launch_finalize(s, b.x, out_f32, out_u8, B, H * W));
    }
    return DPIR_OK;
}

// ------------------------------------------------------------------------------------------ profiling
int dpir_graph_cache_size(dpir_engine* e) { return e ? (int)e->graphs.size() : 0; }
int dpir_prof_enable(dpir_engine* e, int on) {
    if (!e) return DPIR_ERR_INVALID;
    e->prof.collect();
    e->prof.on = on != 0;
    return DPIR_OK;
}
int dpir_prof_reset(dpir_engine* e) {
    if (!e) return DPIR_ERR_INVALID;
    e->prof.reset();
    return DPIR_OK;
}
int dpir_prof_read(dpir_engine* e, double* ms_out, int64_t* count_out) {
    if (!e || !ms_out || !count_out) return DPIR_ERR_INVALID;
    e->prof.collect();
    for (int i = 0; i < DPIR_PROF_CLASSES; ++i) { ms_out[i] = e->prof.ms[i]; count_out[i] = e->prof.cnt[i]; }
    return DPIR_OK;
}

}  // extern "C"
