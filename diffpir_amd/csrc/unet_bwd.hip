// Input gradient of the UNet: the vector-Jacobian product that torch.autograd builds for the reference's DPS modes
// (utils/utils_model.py:390-394, main_ddpir.py:370-373, 434-438).  Weights are frozen (main_ddpir.py:238-239), so
// only activation gradients flow: the tape of the last forward (engine.h Tape, recorded by unet.hip in grad mode) is walked in
// reverse --
//   ResBlock (unet.py:236-256):   dout -> [skip: identity / resampled identity / 1x1 dgrad] + conv2 dgrad -> GN2+FiLM+SiLU backward
//                                 -> conv1 dgrad -> (resampling adjoint) GN1+SiLU backward -> gradient of the (concat) input
//   AttentionBlock (:299-305):    dout -> residual + proj_out dgrad -> QKV attention backward -> qkv dgrad -> GroupNorm backward
// A convolution's dgrad is a FORWARD convolution kernel on the transposed, spatially flipped weight pack built at load time: in the
// f16 precisions conv6 / conv5 (operand-split MFMA; dY is scaled by a run-time power of two before the split because gradients span
// many orders of magnitude, the epilogue undoes it), in f32 mode -- and for shapes those kernels do not tile -- conv2 / conv.
// GroupNorm / SiLU / attention / resampling adjoints are in grad.hip (fp32, fp64 reductions).  A tensor with several consumers
// (block inputs, skip connections) has one gradient buffer: the first contribution writes, later ones accumulate, in tape order --
// deterministic.
#include "engine.h"
#include "grad.h"
#include <unordered_map>
#include <stdlib.h>

namespace dpir {
namespace {

struct Bwd {
    dpir_engine* e; hipStream_t s; Workspace& ws; int B;
    float* partial; size_t partial_cap; float* zeros;
    std::unordered_map<const float*, float*> gbuf;     // activation -> its gradient buffer
    std::unordered_map<const float*, bool> written;
    const float* scaled = nullptr; bool scaled_f16 = false;      // the tensor the shared scale buffers currently describe

    Status grad_of(const float* act, size_t numel, float** out) {
        auto it = gbuf.find(act);
        if (it != gbuf.end()) { *out = it->second; return Status{}; }
        float* g = nullptr;
        DPIR_TRY(ws.getT("grad#" + std::to_string(reinterpret_cast<uintptr_t>(act)), numel, &g));
        gbuf[act] = g; written[act] = false;
        *out = g;
        return Status{};
    }
    bool take_acc(const float* act) { bool w = written[act]; written[act] = true; return w; }     // false: first writer assigns

    // dX [B, cin, H, W] = dgrad of cw applied to dY [B, cout, H, W]
    // reuse_scale: dY is the tensor the PREVIOUS dgrad call scaled (a ResBlock's dout feeds the skip projection's and conv2's dgrad back to back):
    // its absmax / power-of-two scale / table are still in the shared buffers
    Status dgrad(const ConvW& cw, const float* dY, float* dX, int H, int W, bool reuse_scale = false) {
        if (!cw.wT) return Status{DPIR_ERR_STATE, "gradient mode was not enabled before dpir_load_unet (dpir_enable_grad)"};
        static const bool f16_env = !(getenv("DPIR_DGRAD_F32") && atoi(getenv("DPIR_DGRAD_F32")) != 0);      // A/B switch (tools/, tests)
        const bool x1 = e->precision == 2;
        const bool use6 = f16_env && cw.w16T && cw.ks == 3 && conv6_supported(H, W);
        const bool use5 = f16_env && cw.w16T && cw.ks == 1 && conv5_supported(B, cw.cin, H, W);
        if (use6 || use5) {
            // f16 operand-split dgrad on the forward's MFMA kernels.  Gradients span many orders of magnitude: dY is scaled by a run-time
            // power of two (max|dY| s in [512, 1024)) before the split and the epilogue multiplies by 1 / s (grad.hip launch_grad_scale).
            const int C = cw.cout;
            float *part = nullptr, *scal = nullptr; float4* prm = nullptr;
            DPIR_TRY(ws.getT("bwd#absmax", (size_t)512, &part));
            DPIR_TRY(ws.getT("bwd#scal", (size_t)4, &scal));
            DPIR_TRY(ws.getT("bwd#sprm", (size_t)B * 2048, &prm));
            if (C > 2048) return invalid("dgrad: more than 2048 channels");
            if (!(reuse_scale && scaled == dY && scaled_f16)) {
                ProfScope ps(&e->prof, PC_ELEM);
                DPIR_TRY(launch_grad_scale(s, dY, (size_t)B * C * H * W, part, scal, prm, B * C));
            }
            scaled = dY; scaled_f16 = true;
            if (use5) {
                Conv5Args a5;
                a5.src = CatSrc{dY, C, nullptr, 0}; a5.prm = prm; a5.w16 = cw.w16T; a5.w16_scale = cw.w16T_scale;
                a5.bias = zeros; a5.out = dX; a5.res = nullptr; a5.B = B; a5.Cout = cw.cin; a5.H = H; a5.W = W;
                a5.range_ctr = e->range_ctr; a5.x1 = x1; a5.out_scale_dev = scal + 1;
                ProfScope ps(&e->prof, PC_CONV1);
                return launch_conv5(s, a5);
            }
            const int C8 = 2 * ((C + 15) / 16);
            const size_t plane = (size_t)B * C8 * H * W * 16;
            if (plane >= ((size_t)1 << 32)) return invalid("dgrad: split plane exceeds the 4 GiB buffer-descriptor range; reduce the batch");
            char* s16 = nullptr;
            DPIR_TRY(ws.getT("act#s16", 2 * plane, &s16));
            {
                ProfScope ps(&e->prof, PC_ELEM);
                DPIR_TRY(launch_act_split(s, CatSrc{dY, C, nullptr, 0}, prm, 0, B, H, W, s16, x1 ? nullptr : s16 + plane, e->range_ctr));
            }
            Conv6Args a6;
            a6.x1 = x1; a6.xhi = s16; a6.xlo = s16 + plane; a6.w16 = cw.w16T; a6.w16_scale = cw.w16T_scale;
            a6.bias = zeros; a6.out = dX; a6.res = nullptr; a6.res_mode = 0;
            a6.B = B; a6.Cin = C; a6.Cout = cw.cin; a6.H = H; a6.W = W;
            a6.partial = partial; a6.partial_capacity = partial_cap; a6.out_scale_dev = scal + 1;
            ProfScope ps(&e->prof, PC_CONV3);
            return launch_conv6(s, a6);
        }
        scaled = nullptr; scaled_f16 = false;
        ConvArgs a;
        a.src.a = dY; a.src.ca = cw.cout; a.src.Hs = H; a.src.Ws = W; a.src.mode = 0; a.src.prm = nullptr;
        a.w = cw.wT; a.bias = zeros; a.out = dX; a.res = nullptr;
        a.B = B; a.Cin = cw.cout; a.Cout = cw.cin; a.CoutP = cw.coutpT; a.H = H; a.W = W; a.ks = cw.ks;
        a.partial = partial; a.partial_capacity = partial_cap;
        ProfScope ps(&e->prof, cw.ks == 3 ? PC_CONV3 : PC_CONV1);
        return launch_conv(s, a);
    }
    Status gn_bwd(const CatSrc& x, int Hs, int Ws, const float4* prm, const float2* st, const float* dA, int mode, float* ga, bool acc_a, float* gb,
                  bool acc_b, const float* extra = nullptr) {
        GnBwdArgs g;
        g.extra = extra;
        g.x = x; g.prm = prm; g.stats = st; g.dA = dA; g.mode = mode; g.Hs = Hs; g.Ws = Ws;
        DPIR_TRY(ws.getT("bwd#sums", (size_t)B * 32 * 64, &g.sums));
        g.ga = ga; g.gb = gb; g.acc_a = acc_a ? 1 : 0; g.acc_b = acc_b ? 1 : 0;
        ProfScope ps(&e->prof, PC_ELEM);
        return launch_gn_bwd(s, g, B);
    }
    void tap(const std::string& name, const float* p, size_t n) { if (e->collect_taps) e->taps["grad:" + name] = TapInfo{p, n}; }

    Status resblock(const TapeRes& t) {
        const ResW& r = e->net.res[t.idx];
        const size_t on = (size_t)B * r.cout * t.Ho * t.Wo;
        const size_t in_px = (size_t)t.inH * t.inW;
        float* dout = nullptr;
        DPIR_TRY(grad_of(t.out, on, &dout));
        if (!written[t.out]) return Status{DPIR_ERR_STATE, "backward: no gradient reached " + r.name};
        tap(r.name, dout, on);
        float *ga = nullptr, *gb = nullptr;
        DPIR_TRY(grad_of(t.in.a, (size_t)B * t.in.ca * in_px, &ga));
        if (t.in.b) DPIR_TRY(grad_of(t.in.b, (size_t)B * t.in.cb * in_px, &gb));
        // skip branch (unet.py:256 `self.skip_connection(x) + h`)
        const float* fold_identity = nullptr;
        if (t.sk) {
            float* tmp = nullptr;
            DPIR_TRY(ws.getT("bwd#skiptmp", (size_t)B * r.cin * in_px, &tmp));
            DPIR_TRY(dgrad(r.skip, dout, tmp, t.Ho, t.Wo));
            ProfScope ps(&e->prof, PC_ELEM);
            DPIR_TRY(launch_accum_adj(s, tmp, r.cin, 0, ga, t.in.ca, 0, B, t.inH, t.inW, take_acc(t.in.a)));
            if (t.in.b) DPIR_TRY(launch_accum_adj(s, tmp, r.cin, t.in.ca, gb, t.in.cb, 0, B, t.inH, t.inW, take_acc(t.in.b)));
        } else if (r.mode != 0 || t.in.b || r.cout != t.in.ca) {
            ProfScope ps(&e->prof, PC_ELEM);        // the up / down-sampled identity of the resampling blocks (x_upd)
            DPIR_TRY(launch_accum_adj(s, dout, r.cout, 0, ga, t.in.ca, r.mode, B, t.inH, t.inW, take_acc(t.in.a)));
        } else {
            fold_identity = dout;                    // plain identity: added inside the in_layers GroupNorm backward below (GnBwdArgs::extra), not as its own pass
        }
        // out_layers: GroupNorm + FiLM + SiLU + conv2
        float *dA = nullptr, *gh1 = nullptr;
        DPIR_TRY(ws.getT("bwd#dA", (size_t)B * std::max(r.cin, r.cout) * t.Ho * t.Wo, &dA));
        DPIR_TRY(ws.getT("bwd#gh1", on, &gh1));
        DPIR_TRY(dgrad(r.conv2, dout, dA, t.Ho, t.Wo, t.sk != nullptr));
        DPIR_TRY(gn_bwd(CatSrc{t.h1, r.cout, nullptr, 0}, t.Ho, t.Wo, t.prm2, t.st2, dA, 0, gh1, false, nullptr, false));
        tap(r.name + "#h1", gh1, on);
        // in_layers: GroupNorm + SiLU [+ resampling] + conv1
        DPIR_TRY(dgrad(r.conv1, gh1, dA, t.Ho, t.Wo));
        const bool acc_a = take_acc(t.in.a), acc_b = t.in.b ? take_acc(t.in.b) : false;
        DPIR_TRY(gn_bwd(t.in, t.inH, t.inW, t.prm1, t.st1, dA, r.mode, ga, acc_a, gb, acc_b, fold_identity));
        return Status{};
    }

    Status attention(const TapeAttn& t) {
        const AttnW& a = e->net.attn[t.idx];
        const int T = t.H * t.W;
        const size_t n = (size_t)B * a.c * T;
        float *dout = nullptr, *gin = nullptr;
        DPIR_TRY(grad_of(t.out, n, &dout));
        if (!written[t.out]) return Status{DPIR_ERR_STATE, "backward: no gradient reached " + a.name};
        tap(a.name, dout, n);
        DPIR_TRY(grad_of(t.in, n, &gin));
        {
            ProfScope ps(&e->prof, PC_ELEM);        // (x + h) residual of AttentionBlock._forward
            DPIR_TRY(launch_accum_adj(s, dout, a.c, 0, gin, a.c, 0, B, t.H, t.W, take_acc(t.in)));
        }
        float *dAtt = nullptr, *dqkv = nullptr, *P = nullptr, *dP = nullptr, *dXn = nullptr;
        const size_t pp = (size_t)B * (a.c / 64) * T * T;
        DPIR_TRY(ws.getT("bwd#dA", n, &dAtt));
        DPIR_TRY(ws.getT("bwd#dqkv", 3 * n, &dqkv));
        DPIR_TRY(ws.getT("bwd#P", pp, &P));
        DPIR_TRY(ws.getT("bwd#dP", pp, &dP));
        DPIR_TRY(ws.getT("bwd#gh1", n, &dXn));
        DPIR_TRY(dgrad(a.proj, dout, dAtt, t.H, t.W));
        {
            ProfScope ps(&e->prof, PC_ATTN);
            DPIR_TRY(launch_attention_bwd(s, t.qkv, dAtt, dqkv, P, dP, B, a.c, T));
        }
        tap(a.name + "#qkv", dqkv, 3 * n);
        DPIR_TRY(dgrad(a.qkv, dqkv, dXn, t.H, t.W));
        DPIR_TRY(gn_bwd(CatSrc{t.in, a.c, nullptr, 0}, t.H, t.W, t.prm, t.st, dXn, 0, gin, take_acc(t.in), nullptr, false));
        return Status{};
    }
};

}  // namespace

Status unet_backward(dpir_engine* e, const float* gout, float* dx) {
    Tape& tp = e->tape;
    if (!e->grad_enabled) return Status{DPIR_ERR_STATE, "gradient mode is off: call dpir_enable_grad before dpir_load_unet"};
    if (!tp.valid) return Status{DPIR_ERR_STATE, "unet_backward: no forward pass has been recorded"};
    UNet& net = e->net;
    const int B = tp.B, H = tp.H, W = tp.W;
    float* partial = nullptr;
    const size_t partial_cap = (size_t)16 * 1024 * 1024;
    DPIR_TRY(e->ws.getT("conv#partial", partial_cap, &partial));
    float* zeros = nullptr;
    bool fresh = e->ws.bufs.find("grad#zeros") == e->ws.bufs.end();
    DPIR_TRY(e->ws.getT("grad#zeros", (size_t)4096, &zeros));
    if (fresh) DPIR_HIP(hipMemsetAsync(zeros, 0, 4096 * sizeof(float), e->stream));
    Bwd b{e, e->stream, e->ws, B, partial, partial_cap, zeros};

    // out: GroupNorm + SiLU + conv (unet.py:611-616)
    const int ch = net.out_conv.cin;
    const size_t hn = (size_t)B * ch * H * W;
    float *dA = nullptr, *gh = nullptr;
    DPIR_TRY(e->ws.getT("bwd#dA", hn, &dA));
    DPIR_TRY(b.grad_of(tp.final_h, hn, &gh));
    DPIR_TRY(b.dgrad(net.out_conv, gout, dA, H, W));
    DPIR_TRY(b.gn_bwd(CatSrc{tp.final_h, ch, nullptr, 0}, H, W, tp.final_prm, tp.final_st, dA, 0, gh, b.take_acc(tp.final_h), nullptr, false));
    for (int i = (int)tp.nodes.size() - 1; i >= 0; --i) {
        const TapeNode& nd = tp.nodes[i];
        if (nd.kind == 1) DPIR_TRY(b.resblock(tp.res[nd.idx]));
        else DPIR_TRY(b.attention(tp.attn[nd.idx]));
    }
    // input_blocks.0: the 3 -> C convolution
    float* g0 = nullptr;
    DPIR_TRY(b.grad_of(tp.conv_in_out, (size_t)B * net.conv_in.cout * H * W, &g0));
    if (!b.written[tp.conv_in_out]) return Status{DPIR_ERR_STATE, "backward: no gradient reached input_blocks.0"};
    b.tap("input_blocks.0.0", g0, (size_t)B * net.conv_in.cout * H * W);
    DPIR_TRY(b.dgrad(net.conv_in, g0, dx, H, W));
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
