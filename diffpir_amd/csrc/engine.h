// Engine state: device arena, UNet plan + weights, workspaces.
#pragma once
#include <functional>
#include "common.h"
#include "elem.h"

namespace dpir {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// name -> device buffer, grown on demand (never during graph capture: a warm-up pass allocates)
struct Workspace {
    std::map<std::string, DevBuf> bufs;
    size_t total = 0;
    bool frozen = false;   // true while a graph is being captured: allocation is an error
    uint64_t generation = 0;   // bumped whenever a buffer is (re)allocated: captured graphs become stale
    Status get(const std::string& name, size_t bytes, void** out);
    template <class T> Status getT(const std::string& name, size_t count, T** out) {
        void* p = nullptr;
        DPIR_TRY(get(name, count * sizeof(T), &p));
        *out = reinterpret_cast<T*>(p);
        return Status{};
    }
    void release();
};

struct ConvW { float* w = nullptr; float* bias = nullptr; int cin = 0, cout = 0, coutp = 0, ks = 3;
               void* w16 = nullptr; float w16_scale = 1.f;      // operand-split f16 copy (precision modes 1, 2): conv6 layout for 3x3, conv5 layout for 1x1
               float* wT = nullptr; int coutpT = 0;             // grad mode: the dgrad operand [coutP16][taps flipped][cinP64] (unet_bwd.hip)
               void* w8 = nullptr; float w8_scale = 1.f;        // 3x3 with cout <= 16 and cin % 32 == 0: conv8 layout (fused GroupNorm/SiLU/split prologue)
               void* w16T = nullptr; float w16T_scale = 1.f; }; // grad mode in the f16 precisions: the same operand in conv6 / conv5 split layout
struct GnW { float* gamma = nullptr; float* beta = nullptr; int c = 0; };
struct ResW {
    std::string name;
    int cin = 0, cout = 0, mode = 0;   // mode: 0 plain, 1 up, 2 down
    GnW gn1; ConvW conv1; GnW gn2; ConvW conv2;
    bool has_skip = false; ConvW skip;
    int film_off = 0;
};
struct AttnW { std::string name; int c = 0; GnW norm; ConvW qkv; ConvW proj; };
struct Layer { int kind; int idx; std::string name; };   // kind 0 conv_in, 1 res, 2 attn
typedef std::vector<Layer> Block;

struct UNet {
    bool loaded = false;
    dpir_unet_desc desc{};
    std::vector<float> cm;       // resolved channel_mult
    std::vector<Block> in_blocks, out_blocks;
    Block mid;
    ConvW conv_in;
    std::vector<ResW> res;
    std::vector<AttnW> attn;
    GnW out_gn; ConvW out_conv;
    float *te_w0 = nullptr, *te_b0 = nullptr, *te_w2 = nullptr, *te_b2 = nullptr, *label_emb = nullptr, *freqs = nullptr;
    float *film_w = nullptr, *film_b = nullptr;
    int film_rows = 0;
    std::vector<void*> allocs;   // every weight allocation (freed on reload / destroy)
};

struct TapInfo { const float* p; size_t numel; };

// What the last forward left behind for the input-gradient pass (grad mode only; unet_bwd.hip walks it in reverse)
struct TapeRes { int idx; CatSrc in; int inH, inW, Ho, Wo; float* h1; float* sk; float* out; float4* prm1; float2* st1; float4* prm2; float2* st2; };
struct TapeAttn { int idx; const float* in; int H, W; float* qkv; float* att; float* out; float4* prm; float2* st; };
struct TapeNode { int kind; int idx; };      // 1: res[idx], 2: attn[idx]
struct Tape {
    bool valid = false;
    unsigned long long serial = 0;      // which forward recorded it (dpir_engine::fwd_serial at that time)
    int B = 0, H = 0, W = 0;
    float* conv_in_out = nullptr;
    std::vector<TapeNode> nodes; std::vector<TapeRes> res; std::vector<TapeAttn> attn;
    const float* final_h = nullptr; float4* final_prm = nullptr; float2* final_st = nullptr;
    void clear() { valid = false; nodes.clear(); res.clear(); attn.clear(); }
};

struct ProxState {   // dpir_prox
    int B = 0, H = 0, W = 0, sf = 1;
    float2* FB = nullptr; float* F2B = nullptr; float2* FBFy = nullptr;
    bool half = false;     // true: half spectrum [.., H, WP] (fft2.hip; columns alias-grouped when sf > 1); false: bit-reversed full c2c (fft.hip)
    bool colmajor = false; // half spectrum stored COLUMN-major [.., WP slots, H] for the wave-per-transform kernels (fft4.hip, 256 x 256 and 512 x 512)
    int WP = 0;            // stored row length (complex elements); colmajor: stored columns (slots) per plane
    float* invW = nullptr; // half && sf > 1: alias mean of F2B [B][H/sf][W/sf/2+1]
    const int* slot_col = nullptr; const int* col_slot = nullptr;     // half && sf > 1: device slot maps (engine-owned, fft2_map)
    const std::vector<int>* h_col_slot = nullptr;                     // host copy (dpir_prox_read)
};

struct ResizerTab { int in_len = 0, out_len = 0, taps = 0; float* w = nullptr; int* idx = nullptr; };

}  // namespace dpir

struct dpir_prox { dpir::ProxState st; };

struct dpir_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    dpir::Profiler prof;
    dpir::UNet net;
    dpir::Workspace ws;          // UNet activations + loop state
    std::map<std::string, dpir::TapInfo> taps;
    std::map<int, dpir::FftPlan> fft_plans;
    std::map<int, float2*> fft2_tw;              // W_N^m tables (N entries) for fft2.hip
    struct Fft2Map { int* slot_col = nullptr; int* col_slot = nullptr; std::vector<int> h_slot_col, h_col_slot; };
    std::map<std::pair<int, int>, Fft2Map> fft2_maps;   // (N, sf [+ 16 for the column-major layout]) -> alias-grouped column permutation of the half-spectrum layout
    std::map<std::pair<int, int>, dpir::ResizerTab> resizers;   // (in_len, sf)
    std::vector<void*> user_allocs;
    bool collect_taps = true;
    int precision = 0;           // 0: exact fp32 MFMA kernels; 1: operand-split f16x3 MFMA (fp32-equivalent accuracy)
    bool grad_enabled = false;   // dpir_enable_grad before dpir_load_unet: dgrad weight packs + a tape per forward (DPS modes, 8f-4)
    dpir::Tape tape;
    // the last p_sample (dpir_p_sample / the DPS loop): what dpir_grad_and_value(x, x_hat = that call's pred_xstart) differentiates through
    float ps_c1 = 0.f, ps_c2 = 0.f; int ps_B = 0, ps_H = 0, ps_W = 0; const float* ps_x0 = nullptr;
    unsigned long long fwd_serial = 0, ps_serial = 0;   // forwards so far; the forward p_sample ran (any later forward rebuilds the tape: ps_serial goes stale)
    // Captured restoration steps.  A graph depends only on what is baked into its kernel arguments: the shape / task /
    // mode fields below and the workspace generation; per-batch pointers, seed and image offset live in a device block
    // (dpir::LoopDev), so every batch of a test set replays the same graph.  Entries are compared field by field on a
    // hit (no hash-only match) and the cache is capped (least recently used entry is destroyed).
    struct GraphKey {
        int32_t task, B, H, W, sf, in_iter, generate_mode, kind;      // kind: bit0 final step, bit1 eta draw
        int32_t host_n1, host_n2, host_rp, has_labels;
        float gamma, guidance;
        uint64_t ws_generation;
    };
    struct GraphEntry { GraphKey key; hipGraphExec_t exec = nullptr; uint64_t last_use = 0; };
    std::vector<GraphEntry> graphs;
    uint64_t graph_clock = 0;
    static constexpr size_t kMaxGraphs = 16;
    dpir::ProxState loop_prox;                   // spectra owned by dpir_run_loop
    void invalidate_graphs() {
        for (auto& g : graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
        graphs.clear();
    }
    unsigned long long* range_ctr = nullptr;     // f16x3 operand range guard (act.hip range_report), device
    // how the half-spectrum data step is run (dpir_set_prox_launch): 1 = wave-per-transform kernels on a column-major spectrum (fft4.hip; 256 x 256 and 512 x 512, default),
    // 0 = the two-pass register kernels (fft2.hip; every other size always).  cus: CU count of the device.
    int prox_mode = 1; int cus = 256;
    // conv7's fused hop (Conv6Emit) is an inter-workgroup wait; when it times out (the GPU is shared with other engines / processes) the
    // engine does not fail: it switches the hop off for its lifetime and re-runs what the time-out invalidated -- the restoration loop
    // (dpir_run_loop) or the ONE eager forward issued since the last synchronisation (replay_last); see dpir_check_range
    bool fuse_h1_off = false;
    int fwd_since_sync = 0;
    std::function<int()> replay_last;
    // the replay is valid only while NOTHING has been enqueued behind that forward (later work consumed its invalid output) and none of its buffers
    // has been freed: the enqueue serial (Profiler::serial + copies) at the moment the forward returned
    unsigned long long replay_serial = 0;
    unsigned long long enqueue_serial() const { return prof.serial + copy_serial; }
    unsigned long long copy_serial = 0;
    void arm_replay(std::function<int()> fn) { ++fwd_since_sync; replay_last = std::move(fn); replay_serial = enqueue_serial(); }
    void* comm = nullptr; int comm_world = 1, comm_rank = 0;     // RCCL communicator (comm.cpp), or null

    dpir::Status fft_plan(int N, dpir::FftPlan* out);
    dpir::Status fft2_table(int N, const float2** out);
    dpir::Status fft2_map(int N, int sf, const Fft2Map** out, bool colmajor = false);
    dpir::Status resizer(int in_len, int sf, dpir::ResizerTab* out);
};

namespace dpir {
Status unet_load(dpir_engine* e, const dpir_unet_desc* desc, const dpir_tensor* weights, int n);
void unet_free(dpir_engine* e);
// t_dev/y_dev: device int32 [B]
// film_table / film_step: hoisted FiLM projections of a whole schedule (unet_film_table) and the device-resident current step
Status unet_forward(dpir_engine* e, const float* x, const int* t_dev, const int* y_dev, float* out, int B, int H, int W,
                    const float* film_table = nullptr, const StepDev* film_step = nullptr, bool uniform_t = false);
// uniform_t: every image of the batch has the same timestep (what model_fn always passes: utils_model.py:217 `[t_step] * x.shape[0]`) and the model is
// class-unconditional -> the time embedding and the FiLM projection of all ResBlocks are evaluated for ONE row and shared (rows_gemv over B rows was 90 us at B = 16)
Status unet_film_table(dpir_engine* e, const int* t_dev, int n_steps, float* table);
// vector-Jacobian product of the LAST forward (grad mode): gout [B, out_channels, H, W] -> dx [B, 3, H, W]
Status unet_backward(dpir_engine* e, const float* gout, float* dx);
double unet_flops(const UNet& net, int H, int W, int cls = -1);
// comm.cpp: SUM all-reduce of n device doubles in place over the engine's communicator, on the engine stream (DPS_y0's batch-wide norm)
Status comm_allreduce_sum_f64(dpir_engine* e, double* dev, size_t n);
}  // namespace dpir
