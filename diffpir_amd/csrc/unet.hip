// UNet plan builder, weight repacker and forward executor.
//
// Replaces script_util.create_model (guided_diffusion/script_util.py:130-184), UNetModel.__init__'s
// topology walk (unet.py:480-616), load_state_dict (main_ddpir.py:231-240) and UNetModel.forward
// (unet.py:634-663) with ResBlock._forward (unet.py:236-256) and AttentionBlock._forward
// (unet.py:299-305).  Every layer is a handful of launches of the kernels in conv.hip / norm.hip /
// attn.hip; concat, pooling, upsampling, GroupNorm-apply, SiLU, FiLM, bias and residual adds never
// exist as separate passes (see conv.hip).
#include "engine.h"
#include "conv6_params.h"
#include <unordered_map>
#include <cstdint>
#include <math.h>
#include <string.h>
#include <stdlib.h>

namespace dpir {

// ------------------------------------------------------------------------------------------ workspace
Status Workspace::get(const std::string& name, size_t bytes, void** out) {
    auto it = bufs.find(name);
    if (it != bufs.end() && it->second.bytes >= bytes) { *out = it->second.p; return Status{}; }
    if (frozen) return Status{DPIR_ERR_STATE, "workspace allocation of '" + name + "' during graph capture"};
    if (it != bufs.end()) { (void)hipFree(it->second.p); total -= it->second.bytes; bufs.erase(it); }
    void* p = nullptr;
    size_t rb = (bytes + 255) & ~(size_t)255;
    hipError_t err = hipMalloc(&p, rb);
    if (err != hipSuccess) return Status{DPIR_ERR_NOMEM, "hipMalloc(" + std::to_string(rb) + ") for '" + name + "' failed"};
    bufs[name] = DevBuf{p, rb};
    total += rb;
    ++generation;
    *out = p;
    return Status{};
}
void Workspace::release() {
    for (auto& kv : bufs) (void)hipFree(kv.second.p);
    bufs.clear();
    total = 0;
}

// ------------------------------------------------------------------------------------------ load
namespace {
struct WeightMap {
    std::map<std::string, const dpir_tensor*> m;
    Status find(const std::string& key, std::vector<int64_t> shape, const float** out) const {
        auto it = m.find(key);
        if (it == m.end()) return invalid("state-dict key missing: " + key);
        const dpir_tensor* t = it->second;
        if (t->ndim != (int)shape.size()) return invalid("state-dict key " + key + ": wrong rank");
        for (size_t i = 0; i < shape.size(); ++i)
            if (t->shape[i] != shape[i]) return invalid("state-dict key " + key + ": wrong shape");
        if (!t->data) return invalid("state-dict key " + key + ": null data");
        *out = t->data;
        return Status{};
    }
};

Status upload(dpir_engine* e, const float* host, size_t n, float** dev) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) return Status{DPIR_ERR_NOMEM, "hipMalloc for weights failed"};
    e->net.allocs.push_back(p);
    DPIR_HIP(hipMemcpy(p, host, n * sizeof(float), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<float*>(p);
    return Status{};
}

// OIHW (or OI1 for conv1d) -> [CinP][taps][CoutP], zero padded: CinP % 16 == 0, CoutP % 64 == 0 (conv2.hip's LDS-DMA
// copies whole K chunks and 64-channel column blocks without bounds checks)
Status load_conv(dpir_engine* e, const WeightMap& wm, const std::string& p, int cin, int cout, int ks, bool one_d, ConvW* out) {
    const float *w = nullptr, *b = nullptr;
    std::vector<int64_t> shape = one_d ? std::vector<int64_t>{cout, cin, 1} : std::vector<int64_t>{cout, cin, ks, ks};
    DPIR_TRY(wm.find(p + ".weight", shape, &w));
    DPIR_TRY(wm.find(p + ".bias", {cout}, &b));
    int taps = ks * ks;
    int coutp = round_up(cout, 64);
    int cinp = round_up(cin, 16);
    std::vector<float> packed((size_t)cinp * taps * coutp, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t)
                packed[((size_t)ci * taps + t) * coutp + co] = w[((size_t)co * cin + ci) * taps + t];
    out->cin = cin; out->cout = cout; out->coutp = coutp; out->ks = ks;
    DPIR_TRY(upload(e, packed.data(), packed.size(), &out->w));
    DPIR_TRY(upload(e, b, cout, &out->bias));
    if (e->grad_enabled) {
        // dgrad operand: dx[ci, q] = sum_co sum_t w[co, ci, taps-1-t] dy[co, q + off(t)] -- the forward kernel on the transposed,
        // spatially flipped weights, roles of Cin / Cout exchanged (same [CinP][taps][CoutP] packing)
        const int cinT = round_up(cout, 16), coutT = round_up(cin, 64);
        std::vector<float> pt((size_t)cinT * taps * coutT, 0.f);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < taps; ++t)
                    pt[((size_t)co * taps + t) * coutT + ci] = w[((size_t)co * cin + ci) * taps + (taps - 1 - t)];
        out->coutpT = coutT;
        DPIR_TRY(upload(e, pt.data(), pt.size(), &out->wT));
        if (e->precision >= 1 && (ks == 3 || ks == 1)) {
            // the same operand for the f16 kernels: OIHW of the transposed, flipped filter -> conv6 / conv5 packing
            std::vector<float> wt((size_t)cin * cout * taps);
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int t = 0; t < taps; ++t)
                        wt[((size_t)ci * cout + co) * taps + t] = w[((size_t)co * cin + ci) * taps + (taps - 1 - t)];
            std::vector<uint16_t> w16t;
            out->w16T_scale = ks == 3 ? pack_weights_conv6(wt.data(), cin, cout, w16t) : pack_weights_f16x3_1x1(wt.data(), cin, cout, w16t);
            void* pp = nullptr;
            if (hipMalloc(&pp, w16t.size() * 2) != hipSuccess) return Status{DPIR_ERR_NOMEM, "hipMalloc for the split dgrad weights failed"};
            e->net.allocs.push_back(pp);
            DPIR_HIP(hipMemcpy(pp, w16t.data(), w16t.size() * 2, hipMemcpyHostToDevice));
            out->w16T = pp;
        }
    }
    if (e->precision >= 1 && (ks == 3 || ks == 1)) {
        std::vector<uint16_t> w16;
        out->w16_scale = ks == 3 ? pack_weights_conv6(w, cout, cin, w16) : pack_weights_f16x3_1x1(w, cout, cin, w16);
        void* p = nullptr;
        if (hipMalloc(&p, w16.size() * 2) != hipSuccess) return Status{DPIR_ERR_NOMEM, "hipMalloc for split weights failed"};
        e->net.allocs.push_back(p);
        DPIR_HIP(hipMemcpy(p, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
        out->w16 = p;
        if (ks == 3 && !one_d && cout <= 16 && cin % 32 == 0) {
            std::vector<uint16_t> w8;
            out->w8_scale = pack_weights_conv8(w, cout, cin, w8);
            void* p8 = nullptr;
            if (hipMalloc(&p8, w8.size() * 2) != hipSuccess) return Status{DPIR_ERR_NOMEM, "hipMalloc for the conv8 weights failed"};
            e->net.allocs.push_back(p8);
            DPIR_HIP(hipMemcpy(p8, w8.data(), w8.size() * 2, hipMemcpyHostToDevice));
            out->w8 = p8;
        }
    }
    return Status{};
}
Status load_gn(dpir_engine* e, const WeightMap& wm, const std::string& p, int c, GnW* out) {
    const float *g = nullptr, *b = nullptr;
    if (c % 32) return invalid("GroupNorm32 channels not divisible by 32 at " + p);
    DPIR_TRY(wm.find(p + ".weight", {c}, &g));
    DPIR_TRY(wm.find(p + ".bias", {c}, &b));
    out->c = c;
    DPIR_TRY(upload(e, g, c, &out->gamma));
    DPIR_TRY(upload(e, b, c, &out->beta));
    return Status{};
}
}  // namespace

void unet_free(dpir_engine* e) {
    for (void* p : e->net.allocs) (void)hipFree(p);
    e->net = UNet{};
}

Status unet_load(dpir_engine* e, const dpir_unet_desc* d, const dpir_tensor* weights, int n) {
    unet_free(e);
    UNet& net = e->net;
    net.desc = *d;
    if (d->in_channels != 3) return invalid("in_channels must be 3");
    if (d->model_channels <= 0 || d->model_channels % 32) return invalid("model_channels must be a positive multiple of 32");
    if (d->num_head_channels != 64) return Status{DPIR_ERR_UNSUPPORTED, "only num_head_channels=64 is supported"};
    if (d->n_channel_mult > 8 || d->n_attention_ds > 8) return invalid("too many channel_mult / attention_ds entries");
    if (d->n_channel_mult > 0) net.cm.assign(d->channel_mult, d->channel_mult + d->n_channel_mult);
    else if (d->image_size == 512) net.cm = {0.5f, 1, 1, 2, 2, 4, 4};
    else if (d->image_size == 256) net.cm = {1, 1, 2, 2, 4, 4};
    else if (d->image_size == 128) net.cm = {1, 1, 2, 3, 4};
    else if (d->image_size == 64) net.cm = {1, 2, 3, 4};
    else return invalid("unsupported image_size for default channel_mult");
    WeightMap wm;
    for (int i = 0; i < n; ++i) wm.m[weights[i].name] = &weights[i];

    const int mc = d->model_channels, ted = 4 * mc;
    auto has_attn = [&](int ds) { for (int i = 0; i < d->n_attention_ds; ++i) if (d->attention_ds[i] == ds) return true; return false; };
    int film_rows = 0;
    std::vector<std::pair<std::string, int>> film_parts;   // (key prefix, cout)

    auto add_res = [&](const std::string& p, int cin, int cout, int mode, Block& blk) -> Status {
        ResW r; r.name = p; r.cin = cin; r.cout = cout; r.mode = mode;
        DPIR_TRY(load_gn(e, wm, p + ".in_layers.0", cin, &r.gn1));
        DPIR_TRY(load_conv(e, wm, p + ".in_layers.2", cin, cout, 3, false, &r.conv1));
        DPIR_TRY(load_gn(e, wm, p + ".out_layers.0", cout, &r.gn2));
        DPIR_TRY(load_conv(e, wm, p + ".out_layers.3", cout, cout, 3, false, &r.conv2));
        r.has_skip = cin != cout;
        if (r.has_skip) {
            if (mode != 0) return invalid("resampling ResBlock with channel change is not part of this architecture");
            DPIR_TRY(load_conv(e, wm, p + ".skip_connection", cin, cout, 1, false, &r.skip));
        }
        r.film_off = film_rows;
        film_rows += 2 * cout;
        film_parts.push_back({p + ".emb_layers.1", cout});
        blk.push_back(Layer{1, (int)net.res.size(), p});
        net.res.push_back(r);
        return Status{};
    };
    auto add_attn = [&](const std::string& p, int c, Block& blk) -> Status {
        AttnW a; a.name = p; a.c = c;
        if (c % 64) return invalid("attention channels not divisible by 64");
        DPIR_TRY(load_gn(e, wm, p + ".norm", c, &a.norm));
        DPIR_TRY(load_conv(e, wm, p + ".qkv", c, 3 * c, 1, true, &a.qkv));
        DPIR_TRY(load_conv(e, wm, p + ".proj_out", c, c, 1, true, &a.proj));
        blk.push_back(Layer{2, (int)net.attn.size(), p});
        net.attn.push_back(a);
        return Status{};
    };

    // unet.py:480-536
    int ch = (int)(net.cm[0] * mc);
    const int input_ch = ch;
    {
        Block b0;
        DPIR_TRY(load_conv(e, wm, "input_blocks.0.0", 3, ch, 3, false, &net.conv_in));
        b0.push_back(Layer{0, 0, "input_blocks.0.0"});
        net.in_blocks.push_back(b0);
    }
    std::vector<int> chans{ch};
    int ds = 1;
    const int nlev = (int)net.cm.size();
    for (int level = 0; level < nlev; ++level) {
        for (int k = 0; k < d->num_res_blocks; ++k) {
            Block blk;
            std::string p = "input_blocks." + std::to_string(net.in_blocks.size());
            int cout = (int)(net.cm[level] * mc);
            DPIR_TRY(add_res(p + ".0", ch, cout, 0, blk));
            ch = cout;
            if (has_attn(ds)) DPIR_TRY(add_attn(p + ".1", ch, blk));
            net.in_blocks.push_back(blk);
            chans.push_back(ch);
        }
        if (level != nlev - 1) {
            Block blk;
            std::string p = "input_blocks." + std::to_string(net.in_blocks.size());
            DPIR_TRY(add_res(p + ".0", ch, ch, 2, blk));
            net.in_blocks.push_back(blk);
            chans.push_back(ch);
            ds *= 2;
        }
    }
    // unet.py:539-563
    DPIR_TRY(add_res("middle_block.0", ch, ch, 0, net.mid));
    DPIR_TRY(add_attn("middle_block.1", ch, net.mid));
    DPIR_TRY(add_res("middle_block.2", ch, ch, 0, net.mid));
    // unet.py:566-610
    for (int level = nlev - 1; level >= 0; --level) {
        for (int i = 0; i <= d->num_res_blocks; ++i) {
            Block blk;
            std::string p = "output_blocks." + std::to_string(net.out_blocks.size());
            int ich = chans.back(); chans.pop_back();
            int cout = (int)(mc * net.cm[level]);
            int j = 0;
            DPIR_TRY(add_res(p + "." + std::to_string(j++), ch + ich, cout, 0, blk));
            ch = cout;
            if (has_attn(ds)) DPIR_TRY(add_attn(p + "." + std::to_string(j++), ch, blk));
            if (level && i == d->num_res_blocks) {
                DPIR_TRY(add_res(p + "." + std::to_string(j++), ch, ch, 1, blk));
                ds /= 2;
            }
            net.out_blocks.push_back(blk);
        }
    }
    DPIR_TRY(load_gn(e, wm, "out.0", ch, &net.out_gn));
    DPIR_TRY(load_conv(e, wm, "out.2", input_ch, d->out_channels, 3, false, &net.out_conv));
    if (ch != input_ch) return invalid("final channel count mismatch");

    // time embedding (unet.py:470-478) + all FiLM projections concatenated into one [R, ted] matrix
    const float* p0 = nullptr;
    DPIR_TRY(wm.find("time_embed.0.weight", {ted, mc}, &p0)); DPIR_TRY(upload(e, p0, (size_t)ted * mc, &net.te_w0));
    DPIR_TRY(wm.find("time_embed.0.bias", {ted}, &p0)); DPIR_TRY(upload(e, p0, ted, &net.te_b0));
    DPIR_TRY(wm.find("time_embed.2.weight", {ted, ted}, &p0)); DPIR_TRY(upload(e, p0, (size_t)ted * ted, &net.te_w2));
    DPIR_TRY(wm.find("time_embed.2.bias", {ted}, &p0)); DPIR_TRY(upload(e, p0, ted, &net.te_b2));
    if (d->num_classes > 0) {
        DPIR_TRY(wm.find("label_emb.weight", {d->num_classes, ted}, &p0));
        DPIR_TRY(upload(e, p0, (size_t)d->num_classes * ted, &net.label_emb));
    }
    {
        std::vector<float> fw((size_t)film_rows * ted), fb(film_rows);
        size_t row = 0;
        for (auto& fp : film_parts) {
            const float *w = nullptr, *b = nullptr;
            DPIR_TRY(wm.find(fp.first + ".weight", {2 * fp.second, ted}, &w));
            DPIR_TRY(wm.find(fp.first + ".bias", {2 * fp.second}, &b));
            memcpy(&fw[row * ted], w, (size_t)2 * fp.second * ted * sizeof(float));
            memcpy(&fb[row], b, (size_t)2 * fp.second * sizeof(float));
            row += 2 * fp.second;
        }
        DPIR_TRY(upload(e, fw.data(), fw.size(), &net.film_w));
        DPIR_TRY(upload(e, fb.data(), fb.size(), &net.film_b));
        net.film_rows = film_rows;
    }
    {   // nn.py:114-116, correctly rounded on the host
        int half = mc / 2;
        std::vector<float> fr(half);
        for (int i = 0; i < half; ++i) {
            float arg = -(float)log(10000.0) * (float)i / (float)half;   // float32 arithmetic as in torch
            fr[i] = (float)exp((double)arg);
        }
        DPIR_TRY(upload(e, fr.data(), half, &net.freqs));
    }
    net.loaded = true;
    return Status{};
}

// ------------------------------------------------------------------------------------------ forward
namespace {
struct Act {   // an activation tensor (possibly a virtual concat of two)
    const float* a = nullptr; int ca = 0;
    const float* b = nullptr; int cb = 0;
    int H = 0, W = 0;
    int C() const { return ca + cb; }
};

struct Fwd {
    dpir_engine* e;
    hipStream_t s;
    Workspace& ws;
    int B;
    const float* film;     // [B, film_rows], or the hoisted table [n_steps, film_rows] when fstep != null
    int film_rows;
    int film_stride;       // film_rows, or 0 for the hoisted table (one row per step, shared by the batch)
    const StepDev* fstep;  // device-resident current step (row selector of the hoisted table) or null
    float* partial;        // split-K slab shared by all convolutions of the forward
    size_t partial_cap;
    // GroupNorm statistics already produced by the epilogue of the convolution that wrote a tensor (keyed by its address)
    struct FusedStat { const float2* slots; int nslots; const double2* part; };   // epilogue slots, or per-plane fp64 records (split-K combine)
    std::unordered_map<const float*, FusedStat> fused;
    // The most recent 3x3 convolution if it ran split-K and its slabs are not combined yet: either the next layer's fused
    // low-resolution prologue (gn_act_small) finishes it, or resolve() does -- before anything else reads the tensor or
    // reuses the slab buffer.
    PendingConv pending;
    bool grad = false;    // grad mode: GroupNorm tables kept (also by the fused prologues), tape recorded
    bool fuse_small;
    int emit_skip;        // conv5 emits conv1's operand planes in ResBlocks with a 1x1 skip projection: 0 never, 1 when the projection has
                          // ONE 128-channel output block (each input row is then read exactly once; with two blocks only one of them emits and
                          // the other re-reads -- measured slower on the ImageNet-256 topology, profiles/r03), 2 always

    Status resolve() {
        if (!pending.partial) return Status{};
        ProfScope ps(&e->prof, PC_CONV3);
        DPIR_TRY(launch_conv6_resolve(s, pending));
        if (pending.stat_plane) fused[pending.out] = FusedStat{nullptr, 0, pending.stat_plane};
        else fused.erase(pending.out);
        pending = PendingConv{};
        return Status{};
    }
    bool is_pending(const float* p) const { return pending.partial && p && p == pending.out; }

    // GroupNorm (+FiLM) + SiLU + 3x3 convolution.  Low-resolution layers on the f16 path take the fused prologue; everything
    // else the separate statistics / gn_prm / act_split (or fp32 in-kernel prologue) route.
    Status gn_conv(const GnW& g, const std::string& tag, int film_off, const ConvW& cw, const Act& in, int mode,
                   const float* res, int res_mode, float* out, int Ho, int Wo, float4** prm_out = nullptr, float2** stats_out = nullptr,
                   const Conv6Emit* emit = nullptr) {
        const int C = in.C();
        const bool f16path = cw.w16 && cw.ks == 3 && conv6_supported(Ho, Wo);
        if (fuse_small && f16path && C == g.c && C % 16 == 0 && gn_act_small_supported(C, in.H, in.W, mode)) {
            const bool x1 = e->precision == 2;
            int eh = mode == 1 ? Ho / 2 : (mode == 2 ? Ho * 2 : Ho), ew = mode == 1 ? Wo / 2 : (mode == 2 ? Wo * 2 : Wo);
            if (eh != in.H || ew != in.W) return invalid("conv: source resolution does not match mode");
            if (pending.partial && !is_pending(in.a)) DPIR_TRY(resolve());
            if (is_pending(res)) return invalid("gn_conv: residual is an unfinished convolution");
            const int C8 = 2 * ((C + 15) / 16);
            const size_t plane = (size_t)B * C8 * Ho * Wo * 16;
            char* s16 = nullptr;
            DPIR_TRY(ws.getT("act#s16", 2 * plane, &s16));
            GnActArgs ga;
            ga.src = CatSrc{in.a, in.ca, in.b, in.cb};
            ga.pend = pending;
            ga.gamma = g.gamma; ga.beta = g.beta;
            ga.film = film_off >= 0 ? film : nullptr; ga.film_stride = film_stride; ga.film_off = film_off < 0 ? 0 : film_off;
            ga.fstep = fstep; ga.frows = film_rows;
            ga.silu = true; ga.mode = mode; ga.B = B; ga.Hs = in.H; ga.Ws = in.W;
            ga.hi = s16; ga.lo = x1 ? nullptr : s16 + plane; ga.range_ctr = e->range_ctr;
            if (grad) {     // the backward pass reads the same tables gn_prm_kernel would have written
                DPIR_TRY(ws.getT(tag + "#prm", (size_t)B * C, &ga.prm_out));
                DPIR_TRY(ws.getT(tag + "#gst", (size_t)B * 32, &ga.stats_out));
                if (prm_out) *prm_out = ga.prm_out;
                if (stats_out) *stats_out = ga.stats_out;
            }
            {
                ProfScope ps(&e->prof, PC_ELEM);
                DPIR_TRY(launch_gn_act_small(s, ga));
            }
            if (pending.partial) { fused.erase(pending.out); pending = PendingConv{}; }   // finished (and stored) by the fused prologue
            return conv6_on_planes(cw, s16, plane, res, res_mode, out, Ho, Wo, emit);
        }
        DPIR_TRY(resolve());
        float4* prm = nullptr;
        DPIR_TRY(gn(g, in, tag, film_off, true, &prm, stats_out));
        if (prm_out) *prm_out = prm;
        return conv(cw, in, mode, prm, res, res_mode, out, Ho, Wo, emit);
    }

    Status conv6_on_planes(const ConvW& cw, char* s16, size_t plane, const float* res, int res_mode, float* out, int Ho, int Wo,
                           const Conv6Emit* emit = nullptr) {
        const bool x1 = e->precision == 2;
        Conv6Args a6;
        a6.x1 = x1;
        a6.xhi = s16; a6.xlo = s16 + plane; a6.w16 = cw.w16; a6.w16_scale = cw.w16_scale;
        a6.bias = cw.bias; a6.out = out; a6.res = res; a6.res_mode = res_mode;
        a6.B = B; a6.Cin = cw.cin; a6.Cout = cw.cout; a6.H = Ho; a6.W = Wo;
        a6.partial = partial; a6.partial_capacity = partial_cap;
        const int slots = conv6_stat_slots(Ho, Wo);
        float2* st = nullptr; double2* sp = nullptr;
        if (slots > 0 && cw.cout % 32 == 0) {
            const std::string key = std::to_string(reinterpret_cast<uintptr_t>(out));
            DPIR_TRY(ws.getT("st#" + key, (size_t)B * cw.cout * slots, &st));
            DPIR_TRY(ws.getT("sp#" + key, (size_t)B * cw.cout, &sp));
        }
        a6.stat = st; a6.stat_plane = sp;
        a6.emit = emit;
        if (emit) { a6.stat = nullptr; a6.stat_plane = nullptr; a6.partial = nullptr; a6.partial_capacity = 0; }
        int kind = 0;
        PendingConv pc;
        ProfScope ps(&e->prof, PC_CONV3);
        DPIR_TRY(launch_conv6(s, a6, &kind, fuse_small ? &pc : nullptr));
        if (kind == 1) fused[out] = FusedStat{st, slots, nullptr};
        else if (kind == 2) fused[out] = FusedStat{nullptr, 0, sp};
        else fused.erase(out);
        if (kind == 3) pending = pc;
        return Status{};
    }

    Status conv(const ConvW& cw, const Act& in, int mode, const float4* prm, const float* res, int res_mode, float* out, int Ho, int Wo,
                const Conv6Emit* emit = nullptr) {
        const bool x1 = e->precision == 2;        // f16x1: single-product mode, hi halves only
        // an unfinished split-K output is finished before anything but the fused prologue reads it (or reuses the slab buffer)
        const bool use5 = cw.w16 && cw.ks == 1 && mode == 0 && (!res || res_mode == 0) && conv5_supported(B, cw.cout, Ho, Wo, prm != nullptr) &&
                          !(prm && in.C() % 16);   // no slab buffer
        if (pending.partial && (is_pending(in.a) || is_pending(in.b) || is_pending(res) || !use5)) DPIR_TRY(resolve());
        // the output layer (128 -> 6): GroupNorm / SiLU / split happen in the convolution's own LDS fill (conv8.hip); grad mode keeps the
        // planes route (the backward pass reads act#s16)
        static const bool use_conv8 = !(getenv("DPIR_CONV8") && atoi(getenv("DPIR_CONV8")) == 0);
        if (use_conv8 && cw.w8 && prm && !grad && !emit && !res && mode == 0 && !in.b && in.H == Ho && in.W == Wo &&
            conv8_supported(B, in.C(), cw.cout, Ho, Wo)) {
            Conv8Args a8;
            a8.x = in.a; a8.prm = prm; a8.w = cw.w8; a8.w_scale = cw.w8_scale; a8.bias = cw.bias; a8.out = out;
            a8.B = B; a8.C = in.C(); a8.Cout = cw.cout; a8.H = Ho; a8.W = Wo; a8.range_ctr = e->range_ctr; a8.x1 = x1;
            fused.erase(out);
            ProfScope ps(&e->prof, PC_CONV3);
            return launch_conv8(s, a8);
        }
        if (cw.w16 && cw.ks == 3 && conv6_supported(Ho, Wo)) {
            // operand-split f16 path: one elementwise pre-pass (GroupNorm/FiLM/SiLU/resample/concat/split),
            // then conv6 (pure LDS-DMA + MFMA, two workgroups per CU); GroupNorm statistics of the output come out of its
            // epilogue or of its split-K combine
            int eh = mode == 1 ? Ho / 2 : (mode == 2 ? Ho * 2 : Ho), ew = mode == 1 ? Wo / 2 : (mode == 2 ? Wo * 2 : Wo);
            if (eh != in.H || ew != in.W) return invalid("conv: source resolution does not match mode");
            const int C = in.C(), C8 = 2 * ((C + 15) / 16);
            const size_t plane = (size_t)B * C8 * Ho * Wo * 16;
            if (plane >= ((size_t)1 << 32)) return invalid("conv: split activation plane exceeds the 4 GiB buffer-descriptor range; reduce the batch");
            char* s16 = nullptr;
            DPIR_TRY(ws.getT("act#s16", 2 * plane, &s16));
            {
                ProfScope ps(&e->prof, PC_ELEM);
                DPIR_TRY(launch_act_split(s, CatSrc{in.a, in.ca, in.b, in.cb}, prm, mode, B, Ho, Wo, s16, x1 ? nullptr : s16 + plane, e->range_ctr));
            }
            return conv6_on_planes(cw, s16, plane, res, res_mode, out, Ho, Wo, emit);
        }
        if (emit) return invalid("conv: fused emission was requested for a launch that is not on the f16 3x3 path");
        if (use5) {
            Conv5Args a5;
            a5.src = CatSrc{in.a, in.ca, in.b, in.cb}; a5.prm = prm; a5.w16 = cw.w16; a5.w16_scale = cw.w16_scale;
            a5.bias = cw.bias; a5.out = out; a5.res = res; a5.B = B; a5.Cout = cw.cout; a5.H = Ho; a5.W = Wo;
            a5.range_ctr = e->range_ctr; a5.x1 = x1;
            fused.erase(out);
            ProfScope ps(&e->prof, PC_CONV1);
            return launch_conv5(s, a5);
        }
        fused.erase(out);
        ConvArgs a;
        a.src.a = in.a; a.src.ca = in.ca; a.src.b = in.b; a.src.cb = in.cb; a.src.Hs = in.H; a.src.Ws = in.W;
        a.src.mode = mode; a.src.prm = prm;
        a.w = cw.w; a.bias = cw.bias; a.out = out; a.res = res; a.res_mode = res_mode;
        a.B = B; a.Cin = cw.cin; a.Cout = cw.cout; a.CoutP = cw.coutp; a.H = Ho; a.W = Wo; a.ks = cw.ks;
        a.partial = partial; a.partial_capacity = partial_cap;
        ProfScope ps(&e->prof, cw.ks == 3 ? PC_CONV3 : PC_CONV1);
        return launch_conv(s, a);
    }
    Status gn(const GnW& g, const Act& in, const std::string& tag, int film_off, bool silu, float4** prm_out, float2** stats_out = nullptr) {
        float4* prm = nullptr;
        float2* gst = nullptr;
        if (grad) DPIR_TRY(ws.getT(tag + "#gst", (size_t)B * 32, &gst));
        if (is_pending(in.a) || is_pending(in.b)) DPIR_TRY(resolve());
        DPIR_TRY(ws.getT(tag + "#prm", (size_t)B * g.c, &prm));
        GnStatSrc src[2];
        const float* tp[2] = {in.a, in.b};
        const int tc[2] = {in.ca, in.cb};
        for (int k = 0; k < 2; ++k) {
            src[k].c = tc[k];
            if (!tp[k] || tc[k] == 0) { src[k].c = 0; continue; }
            auto it = fused.find(tp[k]);
            if (it != fused.end()) { src[k].slots = it->second.slots; src[k].nslots = it->second.nslots; src[k].part = it->second.part; continue; }
            double2* part = nullptr;     // this tensor was not written by a statistics-fusing kernel: one streaming pass
            DPIR_TRY(ws.getT(tag + "#stats" + std::to_string(k), (size_t)B * tc[k], &part));
            ProfScope ps(&e->prof, PC_GN);
            DPIR_TRY(launch_gn_stats(s, CatSrc{tp[k], tc[k], nullptr, 0}, B, in.H * in.W, part));
            src[k].part = part;
        }
        ProfScope ps(&e->prof, PC_ELEM);
        DPIR_TRY(launch_gn_prm(s, src[0], src[1], in.H * in.W, g.gamma, g.beta, film_off >= 0 ? film : nullptr, film_stride, film_off < 0 ? 0 : film_off, B, g.c, silu, prm, fstep, film_rows, gst));
        *prm_out = prm;
        if (stats_out) *stats_out = gst;
        return Status{};
    }
    void tap(const std::string& name, const float* p, size_t numel) {
        if (e->collect_taps) e->taps[name] = TapInfo{p, numel};
    }

    // The hop conv1 -> GroupNorm + FiLM + SiLU -> conv2 without h1 (Conv6Emit): conv1's epilogue writes conv2's operand planes.
    // Accumulators / arrival counters of all fused layers of a forward live in one arena, zeroed once per forward.
    bool fuse_h1 = true;
    long long* fuse_arena = nullptr; size_t fuse_cap = 0, fuse_off = 0;       // in 8-byte words
    bool h1_fusable(const ResW& r, int Ho, int Wo) const {
        return fuse_h1 && !grad && e->precision != 0 && r.conv1.w16 && r.conv2.w16 && r.conv1.ks == 3 && r.conv2.ks == 3 && r.gn2.c == r.cout &&
               conv7_emit_supported(B, r.cout, Ho, Wo) && fuse_arena && fuse_off + (size_t)B * 64 + (size_t)B * (r.cout / 128) <= fuse_cap &&
               (size_t)B * (2 * ((r.cout + 15) / 16)) * Ho * Wo * 16 < ((size_t)1 << 32);
    }
    Status make_emit(const ResW& r, int Ho, int Wo, Conv6Emit* em, char** s16b, size_t* plane2) {
        const bool x1 = e->precision == 2;
        const int C8 = 2 * ((r.cout + 15) / 16);
        *plane2 = (size_t)B * C8 * Ho * Wo * 16;
        DPIR_TRY(ws.getT("act#s16b", 2 * *plane2, s16b));
        em->hi = *s16b; em->lo = x1 ? nullptr : *s16b + *plane2; em->C8 = C8;
        em->gamma = r.gn2.gamma; em->beta = r.gn2.beta;
        em->film = r.film_off >= 0 ? film : nullptr; em->film_stride = film_stride; em->film_off = r.film_off < 0 ? 0 : r.film_off;
        em->frows = film_rows; em->fstep = fstep;
        em->acc = fuse_arena + fuse_off;
        fuse_off += (size_t)B * 64;                     // [B][32 groups][2]
        em->cnt = reinterpret_cast<unsigned*>(fuse_arena + fuse_off);
        fuse_off += ((size_t)B * (r.cout / 128) + 1) / 2;
        em->range_ctr = e->range_ctr;
        // test hooks for the time-out path (tests/test_gpu_benched_batches.py): a short spin limit and an arrival count that cannot be reached
        static const int spin_env = getenv("DPIR_FUSE_SPIN_LIMIT") ? atoi(getenv("DPIR_FUSE_SPIN_LIMIT")) : 0;
        static const int extra_env = getenv("DPIR_FUSE_EXPECT_EXTRA") ? atoi(getenv("DPIR_FUSE_EXPECT_EXTRA")) : 0;
        static const int sleep_env = getenv("DPIR_FUSE_SLEEP") ? atoi(getenv("DPIR_FUSE_SLEEP")) : 5;
        em->sleep_sel = sleep_env;
        if (spin_env > 0) em->spin_limit = spin_env;
        em->expect_extra = extra_env;
        return Status{};
    }

    Status resblock(const ResW& r, const Act& in, Act* out) {
        if (in.C() != r.cin) return invalid("resblock " + r.name + ": input channels mismatch");
        int Ho = r.mode == 1 ? in.H * 2 : (r.mode == 2 ? in.H / 2 : in.H);
        int Wo = r.mode == 1 ? in.W * 2 : (r.mode == 2 ? in.W / 2 : in.W);
        if (r.mode == 2 && ((in.H | in.W) & 1)) return invalid("resblock " + r.name + ": odd size cannot be average-pooled");
        size_t on = (size_t)B * r.cout * Ho * Wo;
        float* h1 = nullptr;
        DPIR_TRY(ws.getT(r.name + "#h1", on, &h1));
        Act h1a; h1a.a = h1; h1a.ca = r.cout; h1a.H = Ho; h1a.W = Wo;
        const float* res = nullptr; int res_mode = 0;
        // ResBlock with a 1x1 skip projection at a resolution the fused low-resolution prologue does not take: ONE pass over the
        // (concat) input feeds both the skip projection and in_layers -- conv5 emits conv1's split operand planes
        const int Cin = in.C();
        const size_t eplane = (size_t)B * (2 * ((Cin + 15) / 16)) * Ho * Wo * 16;
        if ((emit_skip == 2 || (emit_skip == 1 && r.cout <= 128)) && r.has_skip && r.mode == 0 && r.conv1.w16 && r.skip.w16 && conv6_supported(Ho, Wo) && conv5_supported(B, r.cout, Ho, Wo) &&
            Cin % 16 == 0 && Cin <= kConv5EmitMaxC && (Ho * Wo) % 256 == 0 && eplane < ((size_t)1 << 32) && !(fuse_small && gn_act_small_supported(Cin, in.H, in.W, 0))) {
            const bool x1 = e->precision == 2;
            float4* prm1 = nullptr;
            TapeRes tr{};
            DPIR_TRY(resolve());
            DPIR_TRY(gn(r.gn1, in, r.name + "#gn1", -1, true, &prm1, &tr.st1));
            tr.prm1 = prm1;
            char* s16 = nullptr;
            DPIR_TRY(ws.getT("act#s16", 2 * eplane, &s16));
            float* sk = nullptr;
            DPIR_TRY(ws.getT(r.name + "#skip", on, &sk));
            Conv5Args a5;
            a5.src = CatSrc{in.a, in.ca, in.b, in.cb}; a5.prm = nullptr; a5.w16 = r.skip.w16; a5.w16_scale = r.skip.w16_scale;
            a5.bias = r.skip.bias; a5.out = sk; a5.res = nullptr; a5.B = B; a5.Cout = r.cout; a5.H = Ho; a5.W = Wo;
            a5.range_ctr = e->range_ctr; a5.x1 = x1;
            a5.emit_prm = prm1; a5.emit_hi = s16; a5.emit_lo = x1 ? nullptr : s16 + eplane;
            fused.erase(sk);
            {
                ProfScope ps(&e->prof, PC_CONV1);
                DPIR_TRY(launch_conv5(s, a5));
            }
            float* o = nullptr;
            DPIR_TRY(ws.getT(r.name + "#out", on, &o));
            if (h1_fusable(r, Ho, Wo)) {
                Conv6Emit em; char* s16b = nullptr; size_t plane2 = 0;
                DPIR_TRY(make_emit(r, Ho, Wo, &em, &s16b, &plane2));
                DPIR_TRY(conv6_on_planes(r.conv1, s16, eplane, nullptr, 0, nullptr, Ho, Wo, &em));
                DPIR_TRY(conv6_on_planes(r.conv2, s16b, plane2, sk, 0, o, Ho, Wo));
            } else {
                DPIR_TRY(conv6_on_planes(r.conv1, s16, eplane, nullptr, 0, h1, Ho, Wo));
                tap(r.name + "#h1", h1, on);
                DPIR_TRY(gn_conv(r.gn2, r.name + "#gn2", r.film_off, r.conv2, h1a, 0, sk, 0, o, Ho, Wo, &tr.prm2, &tr.st2));
            }
            tap(r.name, o, on);
            if (grad) {
                tr.idx = (int)(&r - e->net.res.data());
                tr.in = CatSrc{in.a, in.ca, in.b, in.cb}; tr.inH = in.H; tr.inW = in.W; tr.Ho = Ho; tr.Wo = Wo; tr.h1 = h1; tr.sk = sk; tr.out = o;
                e->tape.nodes.push_back(TapeNode{1, (int)e->tape.res.size()});
                e->tape.res.push_back(tr);
            }
            out->a = o; out->ca = r.cout; out->b = nullptr; out->cb = 0; out->H = Ho; out->W = Wo;
            return Status{};
        }
        TapeRes tr{};
        // (Measured dead end, round 4: forking the 1x1 skip projection onto a side stream -- a parallel branch of the captured step graph --
        // so that it runs under conv1: 19.47 / 19.51 ms per forward with it against 19.28 / 19.30 without, 8.28-8.30 images/s either way;
        // profiles/r04/dead_end_side_stream_skip_*.log.)
        float* sk = nullptr;
        if (r.has_skip) DPIR_TRY(ws.getT(r.name + "#skip", on, &sk));
        const bool fuse = h1_fusable(r, Ho, Wo);
        Conv6Emit em; char* s16b = nullptr; size_t plane2 = 0;
        if (fuse) DPIR_TRY(make_emit(r, Ho, Wo, &em, &s16b, &plane2));
        DPIR_TRY(gn_conv(r.gn1, r.name + "#gn1", -1, r.conv1, in, r.mode, nullptr, 0, fuse ? nullptr : h1, Ho, Wo, &tr.prm1, &tr.st1, fuse ? &em : nullptr));
        if (!fuse) tap(r.name + "#h1", h1, on);
        if (r.has_skip) {
            DPIR_TRY(conv(r.skip, in, 0, nullptr, nullptr, 0, sk, Ho, Wo));
            res = sk;
            tr.sk = sk;
        } else {
            if (in.b) return invalid("resblock " + r.name + ": identity skip on a concatenated input");
            res = in.a; res_mode = r.mode;
        }
        float* o = nullptr;
        DPIR_TRY(ws.getT(r.name + "#out", on, &o));
        if (fuse) DPIR_TRY(conv6_on_planes(r.conv2, s16b, plane2, res, res_mode, o, Ho, Wo));
        else DPIR_TRY(gn_conv(r.gn2, r.name + "#gn2", r.film_off, r.conv2, h1a, 0, res, res_mode, o, Ho, Wo, &tr.prm2, &tr.st2));
        tap(r.name, o, on);
        if (grad) {
            tr.idx = (int)(&r - e->net.res.data());
            tr.in = CatSrc{in.a, in.ca, in.b, in.cb}; tr.inH = in.H; tr.inW = in.W; tr.Ho = Ho; tr.Wo = Wo; tr.h1 = h1; tr.out = o;
            e->tape.nodes.push_back(TapeNode{1, (int)e->tape.res.size()});
            e->tape.res.push_back(tr);
        }
        out->a = o; out->ca = r.cout; out->b = nullptr; out->cb = 0; out->H = Ho; out->W = Wo;
        return Status{};
    }

    Status attention(const AttnW& aw, const Act& in, Act* out) {
        if (in.b || in.ca != aw.c) return invalid("attention " + aw.name + ": bad input");
        int T = in.H * in.W;
        float4* prm = nullptr;
        float2* gst = nullptr;
        DPIR_TRY(resolve());
        DPIR_TRY(gn(aw.norm, in, aw.name + "#norm", -1, false, &prm, &gst));
        float *qkv = nullptr, *att = nullptr, *o = nullptr;
        DPIR_TRY(ws.getT(aw.name + "#qkv", (size_t)B * 3 * aw.c * T, &qkv));
        DPIR_TRY(ws.getT(aw.name + "#att", (size_t)B * aw.c * T, &att));
        DPIR_TRY(ws.getT(aw.name + "#out", (size_t)B * aw.c * T, &o));
        DPIR_TRY(conv(aw.qkv, in, 0, prm, nullptr, 0, qkv, in.H, in.W));
        {
            ProfScope ps(&e->prof, PC_ATTN);
            DPIR_TRY(launch_attention(s, qkv, att, B, aw.c, T, 64));
        }
        Act aa; aa.a = att; aa.ca = aw.c; aa.H = in.H; aa.W = in.W;
        DPIR_TRY(conv(aw.proj, aa, 0, nullptr, in.a, 0, o, in.H, in.W));
        tap(aw.name + "#qkv", qkv, (size_t)B * 3 * aw.c * T);
        tap(aw.name + "#att", att, (size_t)B * aw.c * T);
        tap(aw.name, o, (size_t)B * aw.c * T);
        if (grad) {
            e->tape.nodes.push_back(TapeNode{2, (int)e->tape.attn.size()});
            e->tape.attn.push_back(TapeAttn{(int)(&aw - e->net.attn.data()), in.a, in.H, in.W, qkv, att, o, prm, gst});
        }
        out->a = o; out->ca = aw.c; out->b = nullptr; out->cb = 0; out->H = in.H; out->W = in.W;
        return Status{};
    }

    Status run_block(const UNet& net, const Block& blk, Act in, Act* out) {
        Act cur = in;
        for (const Layer& l : blk) {
            Act nxt;
            if (l.kind == 1) DPIR_TRY(resblock(net.res[l.idx], cur, &nxt));
            else if (l.kind == 2) DPIR_TRY(attention(net.attn[l.idx], cur, &nxt));
            else return invalid("unexpected layer kind");
            cur = nxt;
        }
        *out = cur;
        return Status{};
    }
};
}  // namespace

// All FiLM projections of a schedule at once (unet.py:199-205, 471-475): the timestep is uniform over the
// batch, so emb -> emb_layers of every ResBlock depends on the step only.  table[i, :] = rows of step i (t_dev[i]).
Status unet_film_table(dpir_engine* e, const int* t_dev, int n_steps, float* table) {
    UNet& net = e->net;
    if (!net.loaded) return Status{DPIR_ERR_STATE, "dpir_load_unet has not been called"};
    if (net.desc.num_classes > 0) return invalid("FiLM hoisting needs a class-unconditional model");
    const int mc = net.desc.model_channels, ted = 4 * mc;
    float *tmp = nullptr, *semb = nullptr;
    DPIR_TRY(e->ws.getT("embS#tmp", (size_t)n_steps * (mc + ted), &tmp));
    DPIR_TRY(e->ws.getT("embS#semb", (size_t)n_steps * ted, &semb));
    ProfScope ps(&e->prof, PC_ELEM);
    DPIR_TRY(launch_time_embed(e->stream, t_dev, nullptr, net.freqs, net.te_w0, net.te_b0, net.te_w2, net.te_b2, nullptr, n_steps, mc, tmp, semb));
    DPIR_TRY(launch_rows_gemv(e->stream, net.film_w, net.film_b, semb, n_steps, net.film_rows, ted, table));
    return Status{};
}

Status unet_forward(dpir_engine* e, const float* x, const int* t_dev, const int* y_dev, float* out, int B, int H, int W,
                    const float* film_table, const StepDev* film_step, bool uniform_t) {
    UNet& net = e->net;
    if (!net.loaded) return Status{DPIR_ERR_STATE, "dpir_load_unet has not been called"};
    if ((net.desc.num_classes > 0) != (y_dev != nullptr))
        return invalid("must specify y if and only if the model is class-conditional");   // unet.py:643-645
    if (B <= 0 || H <= 0 || W <= 0) return invalid("bad batch / image size");
    Workspace& ws = e->ws;
    hipStream_t s = e->stream;
    ProfScope whole(&e->prof, PC_UNET);
    const int mc = net.desc.model_channels, ted = 4 * mc;
    float *tmp = nullptr, *semb = nullptr, *film = nullptr;
    const bool hoisted = film_table != nullptr && film_step != nullptr;
    static const bool one_row_env = !(getenv("DPIR_UNIFORM_T") && atoi(getenv("DPIR_UNIFORM_T")) == 0);      // A/B switch
    const bool one_row = one_row_env && !hoisted && uniform_t && net.desc.num_classes == 0 && y_dev == nullptr;
    if (hoisted && net.desc.num_classes > 0) return invalid("hoisted FiLM table with a class-conditional model");
    if (!hoisted) {
        DPIR_TRY(ws.getT("emb#tmp", (size_t)B * (mc + ted), &tmp));
        DPIR_TRY(ws.getT("emb#semb", (size_t)B * ted, &semb));
        DPIR_TRY(ws.getT("emb#film", (size_t)B * net.film_rows, &film));
        ProfScope ps(&e->prof, PC_ELEM);
        const int Be = one_row ? 1 : B;          // uniform timestep, no labels: row 0 serves every image (film_stride 0 below)
        DPIR_TRY(launch_time_embed(s, t_dev, y_dev, net.freqs, net.te_w0, net.te_b0, net.te_w2, net.te_b2, net.label_emb, Be, mc, tmp, semb));
        DPIR_TRY(launch_rows_gemv(s, net.film_w, net.film_b, semb, Be, net.film_rows, ted, film));
    }
    // split-K slab: 16 slices of the largest low-resolution output (layers with < 384 workgroups)
    float* partial = nullptr;
    size_t partial_cap = (size_t)16 * 1024 * 1024;   // 64 MiB
    DPIR_TRY(ws.getT("conv#partial", partial_cap, &partial));
    Fwd f{e, s, ws, B, hoisted ? film_table : film, net.film_rows, (hoisted || one_row) ? 0 : net.film_rows, hoisted ? film_step : nullptr, partial, partial_cap};
    {
        static const bool fuse_env = !(getenv("DPIR_FUSE_SMALL") && atoi(getenv("DPIR_FUSE_SMALL")) == 0);   // A/B switch (tools/, tests)
        static const int emit_env = getenv("DPIR_EMIT_SKIP") ? atoi(getenv("DPIR_EMIT_SKIP")) : 1;
        f.grad = e->grad_enabled;
        // gradient mode keeps the fused prologues (round 4): gn_act_small leaves the {mean, a, b, act} / {mean, rstd} tables the backward reads
        f.fuse_small = fuse_env;
        f.emit_skip = emit_env;
        static const bool fuse_h1_env = !(getenv("DPIR_FUSE_H1") && atoi(getenv("DPIR_FUSE_H1")) == 0);
        f.fuse_h1 = fuse_h1_env && !e->fuse_h1_off && !f.grad && e->precision != 0;
        if (f.fuse_h1) {       // accumulators + arrival counters of the fused conv1 -> conv2 hops: one arena, zeroed once per forward
            f.fuse_cap = (size_t)B * 4096;         // 8-byte words: room for ~60 fused hops of (64 + Cout / 128) words per image
            DPIR_TRY(ws.getT("fuse#arena", f.fuse_cap, &f.fuse_arena));
            DPIR_HIP(hipMemsetAsync(f.fuse_arena, 0, f.fuse_cap * sizeof(long long), s));
        }
        ++e->fwd_serial;
        if (f.grad) { e->tape.clear(); e->tape.serial = e->fwd_serial; e->tape.B = B; e->tape.H = H; e->tape.W = W; }
    }
    if (e->collect_taps) e->taps.clear();

    std::vector<Act> hs;
    Act h;
    {
        float* o = nullptr;
        size_t on = (size_t)B * net.conv_in.cout * H * W;
        DPIR_TRY(ws.getT("input_blocks.0.0#out", on, &o));
        Act xin; xin.a = x; xin.ca = 3; xin.H = H; xin.W = W;
        DPIR_TRY(f.conv(net.conv_in, xin, 0, nullptr, nullptr, 0, o, H, W));
        f.tap("input_blocks.0.0", o, on);
        h.a = o; h.ca = net.conv_in.cout; h.H = H; h.W = W;
        hs.push_back(h);
        if (f.grad) e->tape.conv_in_out = o;
    }
    for (size_t i = 1; i < net.in_blocks.size(); ++i) {
        Act o;
        DPIR_TRY(f.run_block(net, net.in_blocks[i], h, &o));
        h = o;
        hs.push_back(h);
    }
    {
        Act o;
        DPIR_TRY(f.run_block(net, net.mid, h, &o));
        h = o;
    }
    for (size_t i = 0; i < net.out_blocks.size(); ++i) {
        Act skip = hs.back(); hs.pop_back();
        if (skip.H != h.H || skip.W != h.W) return invalid("skip connection size mismatch (image size not divisible by the UNet stride)");
        Act cat; cat.a = h.a; cat.ca = h.ca; cat.b = skip.a; cat.cb = skip.ca; cat.H = h.H; cat.W = h.W;   // th.cat([h, hs.pop()], 1)
        Act o;
        DPIR_TRY(f.run_block(net, net.out_blocks[i], cat, &o));
        h = o;
    }
    float4* fprm = nullptr; float2* fst = nullptr;
    DPIR_TRY(f.gn_conv(net.out_gn, "out#gn", -1, net.out_conv, h, 0, nullptr, 0, out, H, W, &fprm, &fst));
    DPIR_TRY(f.resolve());
    if (f.grad) { e->tape.final_h = h.a; e->tape.final_prm = fprm; e->tape.final_st = fst; e->tape.valid = true; }
    f.tap("out", out, (size_t)B * net.desc.out_channels * H * W);
    return Status{};
}

// cls: -1 total, PC_CONV3 (3x3 convs), PC_CONV1 (1x1 convs incl. qkv / proj_out), PC_ATTN (attention matmuls),
// PC_ELEM (linear layers)
double unet_flops(const UNet& net, int H, int W, int cls) {
    if (!net.loaded) return 0.0;
    const double mc = net.desc.model_channels, ted = 4 * mc;
    double f3 = 0, f1 = 0, fa = 0, fl = 2 * (mc * ted + ted * ted);
    double h = H, w = W;
    auto res = [&](const ResW& r) {
        if (r.mode == 1) { h *= 2; w *= 2; } else if (r.mode == 2) { h /= 2; w /= 2; }
        f3 += 2.0 * 9 * r.cin * r.cout * h * w + 2.0 * 9 * r.cout * r.cout * h * w;
        fl += 2.0 * ted * 2 * r.cout;
        if (r.has_skip) f1 += 2.0 * r.cin * r.cout * h * w;
    };
    auto att = [&](const AttnW& a) {
        double T = h * w, c = a.c;
        f1 += 2 * c * 3 * c * T + 2 * c * c * T;
        fa += 2 * 2 * T * T * c;
    };
    auto blk = [&](const Block& b) {
        for (const Layer& l : b) {
            if (l.kind == 0) f3 += 2.0 * 9 * 3 * net.conv_in.cout * h * w;
            else if (l.kind == 1) res(net.res[l.idx]);
            else att(net.attn[l.idx]);
        }
    };
    for (auto& b : net.in_blocks) blk(b);
    blk(net.mid);
    for (auto& b : net.out_blocks) blk(b);
    f3 += 2.0 * 9 * net.out_conv.cin * net.out_conv.cout * H * W;
    if (cls == PC_CONV3) return f3;
    if (cls == PC_CONV1) return f1;
    if (cls == PC_ATTN) return fa;
    if (cls == PC_ELEM) return fl;
    return f3 + f1 + fa + fl;
}

}  // namespace dpir
