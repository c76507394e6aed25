// ndt1.hip — host-side orchestration of the NDT1-CTC forward/backward on one stream.
//
// Replaces the Python layer loop of the reference (models/ndt1.py:408-450 NeuralEncoder.forward,
// :523-589 NDT1.forward ctc branch) and its autograd graph. All launches go to the caller's
// stream; nothing here synchronises or allocates (the plan owns one tiny device buffer for the
// smoothing taps, created in nbci_ndt1_plan_create).
//
// Memory: parameters live in ONE flat f32 buffer (plus a bf16 shadow in bf16 mode) laid out
// [embed | layer 0 | ... | layer L-1 | head] so each backward segment's gradients are one
// contiguous range = one RCCL bucket. Activations saved for backward live in the caller's
// workspace (size from nbci_ndt1_workspace_bytes).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels.h"
#include "plan_common.h"

namespace nbci {

struct LayerOff {
    int64_t ln1w, ln1b, qw, kw, vw, qb, kb, vb, ow, ob, ln2w, ln2b, upw, upb, dnw, dnb;
};

struct Plan {
    nbci_ndt1_config c;
    std::vector<PInfo> params;
    std::vector<LayerOff> L;
    int64_t embw, embb, stkw, stkb, pos, onw, onb, decw, decb;
    int64_t facw = -1, facb = -1;   // factors projection (head segment), -1 = absent
    int64_t dayemb = -1, blkemb = -1;   // prefix-token tables (embedder segment)
    int64_t day_stride = 0;         // adapt: elements between consecutive days' embed layers (weight and bias alike)
    int64_t total;
    std::vector<std::pair<int64_t, int64_t>> seg;  // [begin,end) per segment
    float* d_taps;
    int ntaps;
    bool fused_attn;  // use attention.hip when the shape allows (NBCI_FUSED_ATTN=0 disables)
    bool flash_attn;  // otherwise the masked streaming kernels of attn_flash.hip (bf16, head 32 / 64 / 96 / 128; NBCI_FLASH_ATTN=0 disables)
    int flash_min;    // ... from this many tokens on (NBCI_FLASH_MIN_TOKENS): below it the batched-GEMM + softmax path is as fast (measured)
    // replicated accumulators for the 1-D parameters' gradients (biases, LayerNorm): compact index space
    std::vector<int> flat_of;               // compact index -> flat gradient offset (-1 = padding)
    std::vector<std::pair<int, int>> cseg;  // compact [begin,end) per segment
    int* d_flat_of;
    int compact_total;
    // aux-stream backward (nbci_ndt1_io.aux_stream): fork / join events, created on first use on the plan's device.
    // ev_main: recorded on the main stream where the aux stream may start something; ev_wg[p]: recorded on the aux stream after
    // the weight gradients of a layer of parity p, waited for by the main stream before it rewrites that parity's scratch buffers.
    mutable hipEvent_t ev_main = nullptr, ev_wg[2] = {nullptr, nullptr};
    int compact_of(int64_t flat_off) const {
        for (size_t i = 0; i < cmap.size(); ++i) if (cmap[i].first == flat_off) return cmap[i].second;
        return -1;
    }
    std::vector<std::pair<int64_t, int>> cmap;  // (flat offset of a 1-D param, compact offset)
};

static int64_t add_param(Plan& p, int64_t& cur, const std::string& name, int rows, int cols, int seg) {
    cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
    const int64_t off = cur;
    const int64_t n = (int64_t)rows * (cols > 0 ? cols : 1);
    p.params.push_back({name, off, n, rows, cols, seg});
    cur += n;
    return off;
}

static void build_layout(Plan& p) {
    const auto& c = p.c;
    int64_t cur = 0;
    const int H = c.hidden, I = c.inter, D = c.input_dim;
    if (c.adapt_days > 0) {   // nn.ModuleList of per-day Linears (ndt1.py:124-129): <d>.weight, <d>.bias, uniform stride
        for (int d = 0; d < c.adapt_days; ++d) {
            const int64_t wo = add_param(p, cur, "encoder.embedder.embed_spikes." + std::to_string(d) + ".weight", D, c.n_channels, 0);
            const int64_t bo = add_param(p, cur, "encoder.embedder.embed_spikes." + std::to_string(d) + ".bias", D, 0, 0);
            if (d == 0) { p.embw = wo; p.embb = bo; }
            if (d == 1) p.day_stride = wo - p.embw;
        }
        if (c.adapt_days == 1) p.day_stride = (int64_t)D * c.n_channels + D;
    } else {
        p.embw = add_param(p, cur, "encoder.embedder.embed_spikes.weight", D, c.n_channels, 0);
        p.embb = add_param(p, cur, "encoder.embedder.embed_spikes.bias", D, 0, 0);
    }
    p.stkw = add_param(p, cur, "encoder.embedder.stack_projection.weight", H, D * c.stack_size, 0);
    p.stkb = add_param(p, cur, "encoder.embedder.stack_projection.bias", H, 0, 0);
    p.pos = c.pos ? add_param(p, cur, "encoder.embedder.embed_pos.weight", c.max_F, H, 0) : -1;
    if (c.block_token_blocks > 0) p.blkemb = add_param(p, cur, "encoder.embedder.block_embedding.weight", c.block_token_blocks, H, 0);
    if (c.day_token_days > 0) p.dayemb = add_param(p, cur, "encoder.embedder.day_embedding.weight", c.day_token_days, H, 0);
    cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
    p.seg.push_back({0, cur});
    for (int l = 0; l < c.n_layers; ++l) {
        const int64_t begin = cur;
        const std::string pre = "encoder.layers." + std::to_string(l) + ".";
        LayerOff o;
        o.ln1w = add_param(p, cur, pre + "ln1.weight", H, 0, l + 1);
        o.ln1b = add_param(p, cur, pre + "ln1.bias", H, 0, l + 1);
        // query/key/value are contiguous so the three projections run as ONE [3H][H] GEMM
        o.qw = add_param(p, cur, pre + "attn.query.weight", H, H, l + 1);
        o.kw = add_param(p, cur, pre + "attn.key.weight", H, H, l + 1);
        o.vw = add_param(p, cur, pre + "attn.value.weight", H, H, l + 1);
        o.qb = add_param(p, cur, pre + "attn.query.bias", H, 0, l + 1);
        o.kb = add_param(p, cur, pre + "attn.key.bias", H, 0, l + 1);
        o.vb = add_param(p, cur, pre + "attn.value.bias", H, 0, l + 1);
        o.ow = add_param(p, cur, pre + "attn.out_proj.weight", H, H, l + 1);
        o.ob = add_param(p, cur, pre + "attn.out_proj.bias", H, 0, l + 1);
        o.ln2w = add_param(p, cur, pre + "ln2.weight", H, 0, l + 1);
        o.ln2b = add_param(p, cur, pre + "ln2.bias", H, 0, l + 1);
        o.upw = add_param(p, cur, pre + "mlp.up_proj.weight", I, H, l + 1);
        o.upb = add_param(p, cur, pre + "mlp.up_proj.bias", I, 0, l + 1);
        o.dnw = add_param(p, cur, pre + "mlp.down_proj.weight", H, I, l + 1);
        o.dnb = add_param(p, cur, pre + "mlp.down_proj.bias", H, 0, l + 1);
        cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
        p.L.push_back(o);
        p.seg.push_back({begin, cur});
    }
    const int64_t begin = cur;
    const int hs = c.n_layers + 1;
    p.onw = add_param(p, cur, "encoder.out_norm.weight", H, 0, hs);
    p.onb = add_param(p, cur, "encoder.out_norm.bias", H, 0, hs);
    if (c.factors_size > 0) {   // NeuralFactorsProjection.proj[0] (ndt1.py:362-365)
        p.facw = add_param(p, cur, "encoder.out_proj.proj.0.weight", c.factors_size, H, hs);
        if (c.factors_bias) p.facb = add_param(p, cur, "encoder.out_proj.proj.0.bias", c.factors_size, 0, hs);
    }
    p.decw = add_param(p, cur, "decoder.0.weight", c.vocab, c.factors_size > 0 ? c.factors_size : H, hs);
    p.decb = add_param(p, cur, "decoder.0.bias", c.vocab, 0, hs);
    cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
    p.seg.push_back({begin, cur});
    p.total = cur;
    // compact index space of all 1-D parameters, in parameter order (so a segment's entries are contiguous)
    int cc = 0;
    p.cseg.assign(p.seg.size(), {0, 0});
    int cur_seg = -1;
    for (const PInfo& pi : p.params) {
        if (pi.seg != cur_seg) {
            if (cur_seg >= 0) p.cseg[cur_seg].second = cc;
            cur_seg = pi.seg;
            p.cseg[cur_seg].first = cc;
        }
        if (pi.cols != 0) continue;
        p.cmap.push_back({pi.off, cc});
        for (int i = 0; i < pi.rows; ++i) p.flat_of.push_back((int)(pi.off + i));
        cc += pi.rows;
        while (cc % 4) { p.flat_of.push_back(-1); ++cc; }
    }
    if (cur_seg >= 0) p.cseg[cur_seg].second = cc;
    p.compact_total = cc;
}

// ---- workspace carve -----------------------------------------------------------------------
struct LayerWS {
    size_t x_in, mean1, rstd1, h1, qkv, P, Pd, ad, lse, x_mid, mean2, rstd2, h2, u, g;
};
struct WS {
    size_t xs, y, tmask, tts, tlens;
    std::vector<LayerWS> L;
    size_t x_last, mean_o, rstd_o, xo, logits, alpha, dlogits, argmax;
    size_t fo, fgate, dfo;             // factors projection: output, act'(pre-activation), gradient (M, factors_size)
    size_t wsel, rsel, wpart, bpart;   // adapt: per-sample day weights (B,D,N), row -> day table (B*T), per-sample weight / bias gradients
    size_t scores;                     // f32 (B,nh,Tp,ldS): forward scores, backward dPd
    size_t dx, dtmp, dA[2], dA2[2], dB[2], dB2, dqkv[2], dS, dwin, dpre, rep;   // [layer parity]: what a layer's weight gradients read (see ndt1_backward, aux stream)
    size_t xtok;                       // (prefix tokens) stack-projection output of the spike tokens before the prefix rows are put in front
    size_t dAp;                        // (phase-GEMM embedder backward) dx0 with every sample's tokens zero-padded: (B, P, H)
    int phase_ok, Q, P, npad;          // Q = T / stride output groups = rows per sample block of dAp (P = Q), npad = size/stride - 1 zero rows in front
    size_t dAp_rows, y_slack;          // rows of dAp (npad + B * Q); bytes of zeroed slack behind y
    size_t bytes;
    int Tp, M, ldS, ldP, vpad;
    int npre, Tt, Mk;                  // learned prefix tokens per sequence (day / block), Tt = npre + Tp tokens in the transformer (M = B*Tt rows), Mk = B*Tp
};

static int carve(const Plan& p, int B, int T, int S, WS& w) {
    const auto& c = p.c;
    NBCI_REQUIRE(T >= c.stack_size, NBCI_ESHAPE, "ndt1: sequence shorter than the stacking window");
    const int Tp = 1 + (T - c.stack_size) / c.stack_stride;
    NBCI_REQUIRE(Tp <= c.max_F && Tp <= 1024, NBCI_ESHAPE, "ndt1: more tokens than max_F / 1024");
    const size_t es = c.dtype == NBCI_BF16 ? 2 : 4;
    const size_t rs = c.residual_dtype == NBCI_BF16 ? 2 : 4;   // the residual stream x and its gradient stream dx
    const int npre = (c.day_token_days > 0 ? 1 : 0) + (c.block_token_blocks > 0 ? 1 : 0), Tt = Tp + npre;
    const size_t M = (size_t)B * Tt, Mk = (size_t)B * Tp, H = c.hidden, I = c.inter, D = c.input_dim;
    w.Tp = Tp; w.M = (int)M; w.npre = npre; w.Tt = Tt; w.Mk = (int)Mk;
    w.ldS = (Tt + 3) / 4 * 4;
    w.ldP = (Tt + 7) / 8 * 8;
    w.vpad = (c.vocab + 7) / 8 * 8;
    size_t cur = 0;
    // embedder backward without the (B,T',size*D) window-gradient tensor: possible when the windows tile the bins evenly
    w.phase_ok = (c.dtype == NBCI_BF16 && c.stack_size % c.stack_stride == 0 && T % c.stack_stride == 0 && c.embed_act != NBCI_ACT_GELU &&
                  (c.stack_size / c.stack_stride) * (int)H % 64 == 0 && D % 8 == 0 && npre == 0) ? 1 : 0;
    { static const bool off = measure_env("NBCI_PHASE_DGRAD", 1) == 0; if (off) w.phase_ok = 0; }
    w.Q = T / c.stack_stride; w.npad = c.stack_size / c.stack_stride - 1; w.P = w.Q;
    w.xs = bump(cur, (size_t)B * T * c.n_channels * es);
    // (phase layout: the stack-projection weight gradient reads window rows j = T' .. Q - 1 of every sample against ZERO rows of dx0; the
    // last sample's run (size - stride) bins past the end of y: that slack is part of y and zeroed in the forward)
    w.y_slack = w.phase_ok ? (size_t)(c.stack_size - c.stack_stride) * D * es : 0;
    w.y = bump(cur, (size_t)B * T * D * es + w.y_slack);
    if (c.adapt_days > 0) {
        w.wsel = bump(cur, (size_t)B * D * c.n_channels * es);
        w.rsel = bump(cur, (size_t)B * T * 8);
        w.wpart = bump(cur, (size_t)B * D * c.n_channels * 4);
        w.bpart = bump(cur, (size_t)B * D * 4);
    }
    w.xtok = npre ? bump(cur, Mk * H * 4) : 0;
    w.tmask = bump(cur, M * 4);
    w.tts = bump(cur, M * 8);
    w.tlens = bump(cur, (size_t)B * 4);
    w.L.resize(c.n_layers);
    const size_t nP = (size_t)B * c.n_heads * Tt * w.ldP;
    for (auto& l : w.L) {
        l.x_in = bump(cur, M * H * rs);
        l.mean1 = bump(cur, M * 4); l.rstd1 = bump(cur, M * 4);
        l.h1 = bump(cur, M * H * es);
        l.qkv = bump(cur, M * 3 * H * es);
        l.P = bump(cur, nP * es);
        l.Pd = bump(cur, nP * es);
        l.ad = bump(cur, M * H * es);
        l.lse = bump(cur, (size_t)B * c.n_heads * Tt * 4);
        l.x_mid = bump(cur, M * H * rs);
        l.mean2 = bump(cur, M * 4); l.rstd2 = bump(cur, M * 4);
        l.h2 = bump(cur, M * H * es);
        l.u = bump(cur, M * I * es);
        l.g = bump(cur, M * I * es);
    }
    w.x_last = bump(cur, M * H * rs);
    w.mean_o = bump(cur, M * 4); w.rstd_o = bump(cur, M * 4);
    w.xo = bump(cur, M * H * es);
    if (c.factors_size > 0) {
        w.fo = bump(cur, M * (size_t)c.factors_size * es);
        w.fgate = bump(cur, M * (size_t)c.factors_size * es);
        w.dfo = bump(cur, M * (size_t)c.factors_size * es);
    }
    w.logits = bump(cur, M * w.vpad * 4);
    w.alpha = bump(cur, ctc_alpha_floats(B, Tp, S > 0 ? S : 1) * 4);
    w.dlogits = bump(cur, M * w.vpad * es);
    w.argmax = bump(cur, M * 4);
    w.scores = bump(cur, (size_t)B * c.n_heads * Tt * w.ldS * 4);
    w.dx = bump(cur, M * H * rs);
    w.dtmp = bump(cur, M * H * 4);
    for (int p2 = 0; p2 < 2; ++p2) {
        w.dA[p2] = bump(cur, M * H * es);
        w.dA2[p2] = bump(cur, M * H * es);
        w.dB[p2] = bump(cur, M * std::max(H, I) * es);
        w.dqkv[p2] = bump(cur, M * 3 * H * es);
    }
    w.dB2 = bump(cur, M * H * es);
    w.dS = bump(cur, nP * es);
    // phase layout of dx0: npad zero rows, then per sample Q rows = its T' token rows + npad zero rows (Q - T' = npad when the windows tile
    // the bins evenly). Every sample's tokens are preceded by npad zero rows (what the phase GEMM's windows reach back into) and ALL rows
    // are one stride apart: the phase GEMM's A and both operands of the stack-projection weight gradient are plain (non-view) operands.
    w.dAp_rows = (size_t)w.npad + (size_t)B * w.Q;
    w.dAp = w.phase_ok ? bump(cur, w.dAp_rows * H * es) : 0;
    w.dwin = w.phase_ok ? 0 : bump(cur, M * (size_t)c.stack_size * D * es);
    w.dpre = bump(cur, (size_t)B * T * D * es);
    w.rep = bump(cur, (size_t)NREP * p.compact_total * 4);
    w.bytes = (cur + 255) / 256 * 256;
    return NBCI_OK;
}

struct Ctx {
    const Plan& p;
    const float* pf;      // f32 params
    const void* pw;       // GEMM-weight source (f32 params or bf16 shadow)
    size_t es;
    int dt;
    char* ws;
    WS w;
    hipStream_t s;
    const void* W(int64_t off) const { return (const char*)pw + off * (int64_t)es; }
};

static int validate_io(const Plan& p, const nbci_ndt1_io* io) {
    NBCI_REQUIRE(io, NBCI_EINVAL, "ndt1: null io");
    NBCI_REQUIRE(io->B > 0 && io->T > 0, NBCI_ESHAPE, "ndt1: B and T must be positive");
    NBCI_REQUIRE(io->spikes && io->spikes_mask && io->spikes_timestamp && io->spikes_lengths, NBCI_EINVAL,
                 "ndt1: spikes, spikes_mask, spikes_timestamp, spikes_lengths are required");
    NBCI_REQUIRE(io->workspace, NBCI_EWORKSPACE, "ndt1: null workspace");
    NBCI_REQUIRE(((uintptr_t)io->workspace) % 256 == 0, NBCI_EALIGN, "ndt1: workspace must be 256-byte aligned");
    NBCI_REQUIRE(!(p.c.use_rope && (!io->rope_cos || !io->rope_sin)), NBCI_EINVAL, "ndt1: rope tables required");
    return NBCI_OK;
}

int ndt1_forward(const Plan& p, const float* params, const void* params_lp, const nbci_ndt1_io* io, hipStream_t s) {
    TRY(validate_io(p, io));
    const auto& c = p.c;
    NBCI_REQUIRE(params, NBCI_EINVAL, "ndt1: null params");
    NBCI_REQUIRE(c.dtype == NBCI_F32 || params_lp, NBCI_EINVAL, "ndt1: bf16 mode needs the bf16 parameter shadow");
    Ctx x{p, params, c.dtype == NBCI_BF16 ? params_lp : (const void*)params, (size_t)(c.dtype == NBCI_BF16 ? 2 : 4), c.dtype,
          (char*)io->workspace, {}, s};
    const int B = io->B, T = io->T, S = io->S;
    TRY(carve(p, B, T, S, x.w));
    NBCI_REQUIRE((size_t)io->workspace_bytes >= x.w.bytes, NBCI_EWORKSPACE, "ndt1: workspace too small");
    NBCI_REQUIRE(io->preds, NBCI_EINVAL, "ndt1: preds output is required");
    const WS& w = x.w;
    // Tk spike tokens per sample; Tp = npre + Tk tokens in the transformer once the learned prefix tokens (day / block) are in front
    const int Tk = w.Tp, Tp = w.Tt, npre = w.npre, M = w.M, Mk = w.Mk, H = c.hidden, I = c.inter, D = c.input_dim, nh = c.n_heads, hd = H / nh;
    const int dt = c.dtype;
    const int xdt = c.residual_dtype;   // storage of the residual stream between kernels (every kernel widens it and computes in f32)
    const size_t es = x.es, rs = xdt == NBCI_BF16 ? 2 : 4;
    const bool train = io->train != 0;
    const float p_emb = train ? c.embed_dropout : 0.f, p_lay = train ? c.dropout : 0.f;
    char* ws = x.ws;
    NBCI_REQUIRE(c.day_token_days == 0 || io->day_idx, NBCI_EINVAL, "ndt1: embedder.day_token needs day_idx");
    NBCI_REQUIRE(c.block_token_blocks == 0 || io->block_idx, NBCI_EINVAL, "ndt1: embedder.block_token needs block_idx");

    if (io->want_grad)  // replicated small-gradient accumulators start each step at zero
        NBCI_CHECK_HIP(hipMemsetAsync(ws + w.rep, 0, (size_t)NREP * p.compact_total * 4, s));
    // 0. token bookkeeping + smoothing/noise (ndt1.py:92-107,181-183,207-208)
    TRY(token_prep_launch(io->spikes_mask, io->spikes_timestamp, io->spikes_lengths, B, T, Tk, c.stack_size, c.stack_stride,
                          (int32_t*)(ws + w.tmask), (int64_t*)(ws + w.tts), (int32_t*)(ws + w.tlens), s, npre));
    const bool noise = train && c.noise;
    TRYP("smooth_noise_kernel", 0, (double)B * T * c.n_channels * (4 + es), s,
         smooth_noise_launch(io->spikes, ws + w.xs, dt, B, T, c.n_channels, p.d_taps, p.ntaps, noise ? c.white_noise_sd : 0.f,
                             noise ? c.constant_offset_sd : 0.f, io->seed, s));
    if (w.y_slack) NBCI_CHECK_HIP(hipMemsetAsync(ws + w.y + (size_t)B * T * D * es, 0, w.y_slack, s));   // (see carve)
    // 1. embed Linear + activation (ndt1.py:173-176)
    if (c.adapt_days > 0) {
        // day-specific layers: each sample's day weights gathered side by side -> ONE batched GEMM (batch = sample); the day's bias
        // arrives through the residual gather (row (b,t) -> day[b]; added before the activation)
        NBCI_REQUIRE(io->day_idx, NBCI_EINVAL, "ndt1: embedder.adapt needs day_idx");
        const int wn = D * c.n_channels;
        TRY(adapt_gather_launch(x.W(p.embw), p.day_stride, io->day_idx, ws + w.wsel, (int64_t*)(ws + w.rsel), dt, B, wn, T, c.adapt_days, s));
        nbci_gemm_desc d = gd(T, D, c.n_channels, dt, op(ws + w.xs, es, 0, c.n_channels, 1, 0, 0, (int64_t)T * c.n_channels),
                              op(ws + w.wsel, es, 0, c.n_channels, 1, 0, 0, wn), ws + w.y, D, dt);
        d.batch = B; d.zdiv = 1; d.czs1 = (int64_t)T * D;
        d.residual = params + p.embb; d.ldr = p.day_stride; d.residual_rows = (const int64_t*)(ws + w.rsel); d.residual_first = 1;
        d.act = c.embed_act;
        TRY(gemm_launch_timed(d, s));
    } else {
        nbci_gemm_desc d = gd(B * T, D, c.n_channels, dt, op(ws + w.xs, es, 0, c.n_channels, 1),
                              op(x.W(p.embw), es, 0, c.n_channels, 1), ws + w.y, D, dt);
        d.bias = params + p.embb; d.act = c.embed_act;
        TRY(gemm_launch_timed(d, s));
    }
    // 2. Unfold + stack_projection as a GEMM over the overlapping-window view, + pos-emb gather,
    //    + embed dropout (ndt1.py:138-140,180,188-189,203)
    void* x_cur = ws + (c.n_layers ? w.L[0].x_in : w.x_last);
    {
        const int KS = c.stack_size * D;
        // with prefix tokens the spike tokens go to a side buffer first: the prefix rows are put in front and the embedder dropout is
        // drawn over the whole (B, npre + T', H) block afterwards
        nbci_gemm_desc d = gd(Mk, H, KS, dt, op(ws + w.y, es, 0, (int64_t)c.stack_stride * D, 1, Tk, (int64_t)T * D),
                              op(x.W(p.stkw), es, 0, KS, 1), npre ? (void*)(ws + w.xtok) : x_cur, H, npre ? NBCI_F32 : xdt);
        d.bias = params + p.stkb;
        if (c.pos) {
            d.residual = params + p.pos; d.ldr = H; d.residual_rows = (const int64_t*)(ws + w.tts); d.residual_first = 1;
        }
        if (!npre) { d.drop_p = p_emb; d.seed = io->seed; d.site = 3; }
        TRY(gemm_launch_timed(d, s));
        if (npre) {   // [day, block, tokens...] (ndt1.py:192-201: the block token is prepended first, then the day token)
            const float* tab0 = params + (c.day_token_days > 0 ? p.dayemb : p.blkemb);
            const int64_t* idx0 = c.day_token_days > 0 ? io->day_idx : io->block_idx;
            TRY(prefix_assemble_launch((const float*)(ws + w.xtok), tab0, idx0, npre == 2 ? params + p.blkemb : nullptr,
                                       npre == 2 ? io->block_idx : nullptr, x_cur, B, Tk, npre, H, p_emb, io->seed, 3, s, xdt));
        }
    }
    const float scale = 1.0f / sqrtf((float)hd);
    for (int l = 0; l < c.n_layers; ++l) {
        const LayerWS& lw = w.L[l];
        const LayerOff& lo = p.L[l];
        void* x_in = ws + lw.x_in;
        void* x_mid = ws + lw.x_mid;
        void* x_out = ws + (l + 1 < c.n_layers ? w.L[l + 1].x_in : w.x_last);
        // ---- attention block (ndt1.py:266-292,325)
        TRYP("ln_fwd_kernel", 0, (double)M * H * (rs + es), s,
             layernorm_fwd_launch(x_in, xdt, params + lo.ln1w, params + lo.ln1b, ws + lw.h1, dt, (float*)(ws + lw.mean1),
                                  (float*)(ws + lw.rstd1), M, H, s));
        {
            nbci_gemm_desc d = gd(M, 3 * H, H, dt, op(ws + lw.h1, es, 0, H, 1), op(x.W(lo.qw), es, 0, H, 1), ws + lw.qkv,
                                  3 * H, dt);
            d.bias = params + lo.qb;
            TRY(gemm_launch_timed(d, s));
        }
        if (c.use_rope)
            TRY(rope_launch(ws + lw.qkv, dt, (const int64_t*)(ws + w.tts), io->rope_cos, io->rope_sin, M, H, nh, 0, s));
        if (p.fused_attn && attn_fused_eligible(dt, Tp, H, nh)) {
            TRYP("attn_fwd_kernel", 4.0 * Tp * Tp * hd * B * nh, (double)M * 4 * H * es, s,   // bytes: q, k, v in; merged output out
                 attn_fwd_launch(ws + lw.qkv, (const int32_t*)(ws + w.tmask), ws + lw.ad, (float*)(ws + lw.lse), B, nh, Tp, H, c.context_forward,
                                 c.context_backward, p_lay, io->seed, 16 + 4 * l, 17 + 4 * l, s));
        } else if (p.flash_attn && Tp >= p.flash_min && fattn_eligible(dt, Tp, H, nh)) {   // longer than the one-workgroup kernel holds (T' > 160), or another head size
            TRYP("fa_fwd_kernel", 4.0 * Tp * Tp * hd * B * nh, (double)M * 4 * H * es, s,
                 fattn_masked_fwd_launch(ws + lw.qkv, (const int32_t*)(ws + w.tmask), ws + lw.ad, (float*)(ws + lw.lse), B, nh, Tp, H,
                                         c.context_forward, c.context_backward, p_lay, io->seed, 16 + 4 * l, 17 + 4 * l, s));
        } else {
            {   // scores = q k^T / sqrt(hd), batched over (b, head)
                nbci_gemm_desc d = gd(Tp, Tp, hd, dt, op(ws + lw.qkv, es, 0, 3 * H, 1, 0, 0, (int64_t)Tp * 3 * H, hd),
                                      op(ws + lw.qkv, es, H, 3 * H, 1, 0, 0, (int64_t)Tp * 3 * H, hd), ws + w.scores, w.ldS,
                                      NBCI_F32);
                d.batch = B * nh; d.zdiv = nh; d.czs1 = (int64_t)nh * Tp * w.ldS; d.czs2 = (int64_t)Tp * w.ldS; d.alpha = scale;
                TRY(gemm_launch_timed(d, s));
            }
            TRY(softmax_fwd_launch((const float*)(ws + w.scores), ws + lw.P, ws + (p_lay > 0.f ? lw.Pd : lw.P), dt,
                                   (const int32_t*)(ws + w.tmask), B, nh, Tp, w.ldS, w.ldP, c.context_forward, c.context_backward,
                                   p_lay, io->seed, 16 + 4 * l, s));
            {   // a = dropout(merge_heads(Pd v)) written straight into the merged (M, H) layout
                const size_t pd = p_lay > 0.f ? lw.Pd : lw.P;
                nbci_gemm_desc d = gd(Tp, hd, Tp, dt, op(ws + pd, es, 0, w.ldP, 1, 0, 0, (int64_t)nh * Tp * w.ldP, (int64_t)Tp * w.ldP),
                                      op(ws + lw.qkv, es, 2 * H, 3 * H, 0, 0, 0, (int64_t)Tp * 3 * H, hd), ws + lw.ad, H, dt);
                d.batch = B * nh; d.zdiv = nh; d.czs1 = (int64_t)Tp * H; d.czs2 = hd;
                d.drop_p = p_lay; d.seed = io->seed; d.site = 17 + 4 * l;
                TRY(gemm_launch_timed(d, s));
            }
        }
        {   // x_mid = x_in + out_proj(a)
            nbci_gemm_desc d = gd(M, H, H, dt, op(ws + lw.ad, es, 0, H, 1), op(x.W(lo.ow), es, 0, H, 1), x_mid, H, xdt);
            d.bias = params + lo.ob; d.residual = x_in; d.ldr = H; d.residual_dtype = xdt;
            TRY(gemm_launch_timed(d, s));
        }
        // ---- MLP block (ndt1.py:224-227,328)
        TRYP("ln_fwd_kernel", 0, (double)M * H * (rs + es), s,
             layernorm_fwd_launch(x_mid, xdt, params + lo.ln2w, params + lo.ln2b, ws + lw.h2, dt, (float*)(ws + lw.mean2),
                                  (float*)(ws + lw.rstd2), M, H, s));
        {
            nbci_gemm_desc d = gd(M, I, H, dt, op(ws + lw.h2, es, 0, H, 1), op(x.W(lo.upw), es, 0, H, 1), ws + lw.g, I, dt);
            d.bias = params + lo.upb; d.act = c.mlp_act; d.C2 = ws + lw.u; d.c2_grad = 1;   // lw.u holds act'(u)
            TRY(gemm_launch_timed(d, s));
        }
        {
            nbci_gemm_desc d = gd(M, H, I, dt, op(ws + lw.g, es, 0, I, 1), op(x.W(lo.dnw), es, 0, I, 1), x_out, H, xdt);
            d.bias = params + lo.dnb; d.drop_p = p_lay; d.seed = io->seed; d.site = 18 + 4 * l;
            d.residual = x_mid; d.ldr = H; d.residual_dtype = xdt;
            TRY(gemm_launch_timed(d, s));
        }
    }
    // ---- out_norm + decoder + log-softmax (+ CTC) (ndt1.py:442,494-499,545,581)
    TRYP("ln_fwd_kernel", 0, (double)M * H * (rs + es), s,
         layernorm_fwd_launch(ws + w.x_last, xdt, params + p.onw, params + p.onb, ws + w.xo, dt,
                              (float*)(ws + w.mean_o), (float*)(ws + w.rstd_o), M, H, s));
    const int FS = c.factors_size;
    const void* enc_out = ws + w.xo;   // what the decoder (and a coupler) reads: out_norm(x), or its factors projection
    const int Kd = FS > 0 ? FS : H;
    if (FS > 0) {   // act(Linear(hidden -> factors)); act'(pre-activation) kept for the backward (ndt1.py:372-373)
        nbci_gemm_desc d = gd(M, FS, H, dt, op(ws + w.xo, es, 0, H, 1), op(x.W(p.facw), es, 0, H, 1), ws + w.fo, FS, dt);
        if (p.facb >= 0) d.bias = params + p.facb;
        d.act = c.factors_act; d.C2 = ws + w.fgate; d.c2_grad = 1;
        TRY(gemm_launch_timed(d, s));
        enc_out = ws + w.fo;
    }
    {   // decoder on the spike tokens only: the prefix tokens are dropped after out_norm (ndt1.py:444-448) -> one GEMM per sample block
        nbci_gemm_desc d = npre ? gd(Tk, c.vocab, Kd, dt, op(enc_out, es, (int64_t)npre * Kd, Kd, 1, 0, 0, (int64_t)Tp * Kd),
                                     op(x.W(p.decw), es, 0, Kd, 1), ws + w.logits, w.vpad, NBCI_F32)
                                : gd(M, c.vocab, Kd, dt, op(enc_out, es, 0, Kd, 1), op(x.W(p.decw), es, 0, Kd, 1), ws + w.logits,
                                     w.vpad, NBCI_F32);
        if (npre) { d.batch = B; d.zdiv = 1; d.czs1 = (int64_t)Tk * w.vpad; }
        d.bias = params + p.decb;
        TRY(gemm_launch_timed(d, s));
    }
    int32_t* amax = io->argmax ? io->argmax : (int32_t*)(ws + w.argmax);
    TRYP("logsoftmax_kernel", 0, (double)Mk * (w.vpad + c.vocab) * 4, s, logsoftmax_launch((const float*)(ws + w.logits), w.vpad, io->preds, amax, Mk, c.vocab, s));
    if (io->token_mask_out)   // (B,T'): the spike tokens' validity (the reference's mask also carries the prefix ones; its x does not)
        NBCI_CHECK_HIP(hipMemcpy2DAsync(io->token_mask_out, (size_t)Tk * 4, ws + w.tmask + (size_t)npre * 4, (size_t)Tp * 4, (size_t)Tk * 4, B,
                                        hipMemcpyDeviceToDevice, s));
    if (io->hidden_out)  // optional copy-out of the encoder output (B,T',H) for BCI-style couplers
        NBCI_CHECK_HIP(hipMemcpy2DAsync(io->hidden_out, (size_t)Tk * Kd * es, (const char*)enc_out + (size_t)npre * Kd * es, (size_t)Tp * Kd * es,
                                        (size_t)Tk * Kd * es, B, hipMemcpyDeviceToDevice, s));
    if (io->targets) {
        NBCI_REQUIRE(io->targets_lengths && io->loss && S > 0, NBCI_EINVAL, "ndt1: targets need targets_lengths, loss and S > 0");
        TRYP("ctc_kernel", 0, 0, s,   // (serial in T': latency-bound, no roofline)
             ctc_launch(io->preds, io->targets, (const int32_t*)(ws + w.tlens), io->targets_lengths, B, Tk, c.vocab, S, c.blank_id,
                        c.zero_infinity, io->loss, (float*)(ws + w.alpha), io->want_grad ? ws + w.dlogits : nullptr, dt, w.vpad,
                        io->grad_scale, s));
    }
    return NBCI_OK;
}

// Backward over segments seg_hi .. seg_lo (descending): head = L+1, layers L..1, embed = 0.
// Gradients are ACCUMULATED into `grads` (same flat layout as the parameters).
int ndt1_backward(const Plan& p, const float* params, const void* params_lp, const nbci_ndt1_io* io, float* grads, int seg_hi,
                  int seg_lo, hipStream_t s) {
    TRY(validate_io(p, io));
    const auto& c = p.c;
    NBCI_REQUIRE(params && grads, NBCI_EINVAL, "ndt1: null params/grads");
    NBCI_REQUIRE(c.dtype == NBCI_F32 || params_lp, NBCI_EINVAL, "ndt1: bf16 mode needs the bf16 parameter shadow");
    NBCI_REQUIRE(seg_hi <= c.n_layers + 1 && seg_lo >= 0 && seg_lo <= seg_hi, NBCI_EINVAL, "ndt1: bad segment range");
    Ctx x{p, params, c.dtype == NBCI_BF16 ? params_lp : (const void*)params, (size_t)(c.dtype == NBCI_BF16 ? 2 : 4), c.dtype,
          (char*)io->workspace, {}, s};
    const int B = io->B, T = io->T, S = io->S;
    TRY(carve(p, B, T, S, x.w));
    NBCI_REQUIRE((size_t)io->workspace_bytes >= x.w.bytes, NBCI_EWORKSPACE, "ndt1: workspace too small");
    const WS& w = x.w;
    const int Tk = w.Tp, Tp = w.Tt, npre = w.npre, M = w.M, Mk = w.Mk;   // (see ndt1_forward)
    const int H = c.hidden, I = c.inter, D = c.input_dim, nh = c.n_heads, hd = H / nh, V = c.vocab;
    const int dt = c.dtype;
    const int xdt = c.residual_dtype;   // storage of the residual stream AND of its gradient stream dx (bf16: rounded once per store, summed in f32)
    const bool rb = xdt == NBCI_BF16;
    const size_t es = x.es, rs = rb ? 2 : 4;
    const bool train = io->train != 0;
    const float p_emb = train ? c.embed_dropout : 0.f, p_lay = train ? c.dropout : 0.f;
    char* ws = x.ws;
    float* dx = (float*)(ws + w.dx);   // (bf16 elements when rb)
    float* dtmp = (float*)(ws + w.dtmp);
    const float scale = 1.0f / sqrtf((float)hd);
    float* rep = (float*)(ws + w.rep);
    const RepCfg rc{p.compact_total, NREP};
    // io->aux_stream (bf16 mode): the layers' grouped weight gradients, the stack-projection weight gradient and the fold of the
    // replicated small-vector gradients are queued on a SECOND stream, beside the data-gradient chain on `s`. With few rows per
    // launch (small batches: B = 8 runs 72-144 workgroups on 256 CUs, every launch bound by its own latency) the two streams
    // overlap; the caller decides (llm_bci_amd/trainer.py). Ordering: ev_main forks (main -> aux) wherever the aux work's inputs
    // are complete; the scratch a layer's weight gradients read is double-buffered by layer parity and the main stream waits for
    // ev_wg[parity] before the LayerNorm backward that rewrites it a layer later. The gradients of the segments of this call are
    // then complete ON THE AUX STREAM: the caller orders its optimizer / all-reduce after it and joins it before the next forward.
    hipStream_t aux = (io->aux_stream && dt == NBCI_BF16) ? (hipStream_t)io->aux_stream : s;
    const bool two = aux != s;
    hipStream_t wgs = aux;
    if (two && !p.ev_main) {
        // (no system-scope fence on the fork event: its only waiter is the aux stream of the SAME device - kernel boundaries already publish at agent
        //  scope - and the system fence's cache write-back / invalidate sits between two kernels of the chain every time the event is recorded)
        NBCI_CHECK_HIP(hipEventCreateWithFlags(&p.ev_main, hipEventDisableTiming | hipEventDisableSystemFence));
        NBCI_CHECK_HIP(hipEventCreateWithFlags(&p.ev_wg[0], hipEventDisableTiming | hipEventDisableSystemFence));
        NBCI_CHECK_HIP(hipEventCreateWithFlags(&p.ev_wg[1], hipEventDisableTiming | hipEventDisableSystemFence));
    }
    auto fork = [&]() -> int {   // everything queued on the main stream so far happens-before what is queued on the aux stream from here on
        if (!two) return NBCI_OK;
        NBCI_CHECK_HIP(hipEventRecord(p.ev_main, s));
        NBCI_CHECK_HIP(hipStreamWaitEvent(aux, p.ev_main, 0));
        return NBCI_OK;
    };
    auto RG = [&](int64_t flat_off) -> float* { return rep + p.compact_of(flat_off); };  // replica-0 slot of a 1-D param's grad
    // The f32 gradient stream dx is consumed by GEMMs in the operand dtype: the LayerNorm backward that
    // finalises dx also writes that (dropout-masked) copy into ws.dA and sums its columns (bias grad).
    const bool need_cast = !(dt == NBCI_F32 && p_lay == 0.f && p_emb == 0.f) || npre > 0;   // (prefix tokens: the cast also keeps their rows out of the bias sums)
    auto cast_for = [&](int layer_below) -> LnCast {   // consumer = MLP backward of `layer_below`, or the embedder if < 0
        if (!need_cast) return LnCast{nullptr, 0, 0u, 1.f, 0u, nullptr};
        const float pp = layer_below >= 0 ? p_lay : p_emb;
        const uint32_t site = layer_below >= 0 ? 18 + 4 * layer_below : 3;
        if (layer_below < 0 && w.phase_ok)   // the embedder's phase GEMMs read zero-padded sample blocks
            return LnCast{ws + w.dAp, 1, drop_threshold(pp), pp > 0.f ? 1.f / (1.f - pp) : 1.f, drop_key(io->seed, site), RG(p.stkb),
                          Tk, w.P, w.npad, 0};
        if (layer_below < 0 && npre > 0)     // same layout, but the prefix-token rows stay out of the stack-projection bias gradient
            return LnCast{ws + w.dA[1], dt == NBCI_BF16, drop_threshold(pp), pp > 0.f ? 1.f / (1.f - pp) : 1.f, drop_key(io->seed, site),
                          RG(p.stkb), Tp, Tp, 0, npre};
        return LnCast{ws + w.dA[layer_below & 1], dt == NBCI_BF16, drop_threshold(pp), pp > 0.f ? 1.f / (1.f - pp) : 1.f, drop_key(io->seed, site),
                      RG(layer_below >= 0 ? p.L[layer_below].dnb : p.stkb), 0, 0, 0, 0};
    };

    for (int seg = seg_hi; seg >= seg_lo; --seg) {
        if (seg == c.n_layers + 1) {
            // ---- head: decoder Linear + out_norm (or an external gradient of the encoder output)
            const void* d_xo = dtmp;   // (bf16 in bf16 mode: written by the decoder data-gradient GEMM)
            int d_xo_lp = (dt == NBCI_BF16) ? 1 : 0;
            const int FS = c.factors_size, Kd = FS > 0 ? FS : H;
            const void* enc_out = FS > 0 ? ws + w.fo : ws + w.xo;
            NBCI_REQUIRE(!(io->d_hidden && FS > 0 && npre > 0), NBCI_EINVAL, "ndt1: external encoder gradient with factors AND prefix tokens is not supported");
            if (io->d_hidden && FS == 0) {
                if (npre) {   // (B,T',H) gradient of the stripped output -> the (B, npre + T', H) layout, prefix rows zero
                    NBCI_CHECK_HIP(hipMemsetAsync(dtmp, 0, (size_t)M * H * 4, s));
                    NBCI_CHECK_HIP(hipMemcpy2DAsync((char*)dtmp + (size_t)npre * H * 4, (size_t)Tp * H * 4, io->d_hidden, (size_t)Tk * H * 4,
                                                    (size_t)Tk * H * 4, B, hipMemcpyDeviceToDevice, s));
                } else {
                    d_xo = io->d_hidden;
                }
                d_xo_lp = 0;
            } else {
                if (!io->d_hidden) {   // decoder Linear: bias / weight gradients, then d(encoder output) = dlogits W_d
                    const void* dl = ws + w.dlogits;
                    TRY(colsum_launch(dl, dt, w.vpad, Mk, V, RG(p.decb), s, rc));
                    TRY(wgrad(s, dt, V, Kd, Mk, op(dl, es, 0, w.vpad, 0),
                              npre ? op(enc_out, es, (int64_t)npre * Kd, Kd, 0, Tk, (int64_t)Tp * Kd) : op(enc_out, es, 0, Kd, 0), grads + p.decw, Kd));
                    char* tgt = FS > 0 ? ws + w.dfo : (char*)dtmp;
                    nbci_gemm_desc d;
                    if (npre) {   // spike-token rows only (one GEMM per sample block); the prefix rows of the target stay zero
                        NBCI_CHECK_HIP(hipMemsetAsync(tgt, 0, (size_t)M * Kd * es, s));
                        d = gd(Tk, Kd, V, dt, op(dl, es, 0, w.vpad, 1, 0, 0, (int64_t)Tk * w.vpad), op(x.W(p.decw), es, 0, Kd, 0),
                               tgt + (size_t)npre * Kd * es, Kd, dt);
                        d.batch = B; d.zdiv = 1; d.czs1 = (int64_t)Tp * Kd;
                    } else {
                        d = gd(M, Kd, V, dt, op(dl, es, 0, w.vpad, 1), op(x.W(p.decw), es, 0, Kd, 0), tgt, Kd, dt);
                    }
                    if (FS > 0) {   // through the factors activation: * act'(pre-activation), + the factors bias gradient
                        d.gate = ws + w.fgate + (size_t)npre * FS * es; d.ldg = FS; d.gate_act = -1; d.gate_follows_c = 1;
                        if (p.facb >= 0) { d.colsum = RG(p.facb); d.colsum_rep_stride = rc.stride; d.colsum_nrep = rc.n; }
                    }
                    TRY(gemm_launch_timed(d, s));
                } else {               // external gradient of the factors output (coupler): same activation gate, as a cast
                    TRY(gate_cast_launch(io->d_hidden, ws + w.fgate, ws + w.dfo, dt, (long long)M * FS, s));
                    if (p.facb >= 0) TRY(colsum_launch(ws + w.dfo, dt, FS, M, FS, RG(p.facb), s, rc));
                }
                if (FS > 0) {
                    TRY(wgrad(s, dt, FS, H, M, op(ws + w.dfo, es, 0, FS, 0), op(ws + w.xo, es, 0, H, 0), grads + p.facw, H));
                    nbci_gemm_desc d = gd(M, H, FS, dt, op(ws + w.dfo, es, 0, FS, 1), op(x.W(p.facw), es, 0, H, 0), dtmp, H, dt);
                    TRY(gemm_launch_timed(d, s));
                }
            }
            if (c.n_layers == 0 && w.phase_ok) NBCI_CHECK_HIP(hipMemsetAsync(ws + w.dAp, 0, w.dAp_rows * H * es, s));   // zero pad rows
            TRYP("ln_bwd_kernel", 0, (double)M * H * ((d_xo_lp ? 2 : 4) + 2 * rs + es), s,   // dy, x, dx out, cast out
                 layernorm_bwd_launch(d_xo, d_xo_lp, ws + w.x_last, params + p.onw, (const float*)(ws + w.mean_o), (const float*)(ws + w.rstd_o),
                                     LnStreams{rb, nullptr, dx, rb}, RG(p.onw), RG(p.onb), M, H, s, rc, cast_for(c.n_layers - 1)));
        } else if (seg >= 1) {
            const int l = seg - 1;
            const LayerWS& lw = w.L[l];
            const LayerOff& lo = p.L[l];
            // scratch read by this layer's weight gradients, by layer parity: with an aux stream they run beside the next layer's chain
            const size_t o_dA = w.dA[l & 1], o_dA2 = w.dA2[l & 1], o_dB = w.dB[l & 1], o_dqkv = w.dqkv[l & 1];
            WgradQueue wq; wq.dtype = dt; wq.s = wgs;
            // ---- MLP backward: x_out = x_mid + dropout(down(act(up(ln2(x_mid)))))
            const void* dm;  // d(down output) in the GEMM operand dtype (written by the previous LayerNorm backward)
            if (!need_cast) {
                dm = dx;
                TRY(colsum_launch(dm, dt, H, M, H, RG(lo.dnb), s, rc));
            } else {
                dm = ws + o_dA;
            }
            TRY(wq.push(H, I, M, op(dm, es, 0, H, 0), op(ws + lw.g, es, 0, I, 0), grads + lo.dnw, I));
            {   // du = (dm W_down) * act'(u)
                nbci_gemm_desc d = gd(M, I, H, dt, op(dm, es, 0, H, 1), op(x.W(lo.dnw), es, 0, I, 0), ws + o_dB, I, dt);
                d.gate = ws + lw.u; d.ldg = I; d.gate_act = -1;   // lw.u already holds act'(u) (stored by the forward up_proj GEMM)
                d.colsum = RG(lo.upb); d.colsum_rep_stride = rc.stride; d.colsum_nrep = rc.n;  // up_proj bias grad = column sums of du
                TRY(gemm_launch_timed(d, s));
            }
            TRY(wq.push(I, H, M, op(ws + o_dB, es, 0, I, 0), op(ws + lw.h2, es, 0, H, 0), grads + lo.upw, H));
            {
                nbci_gemm_desc d = gd(M, H, I, dt, op(ws + o_dB, es, 0, I, 1), op(x.W(lo.upw), es, 0, H, 0), dtmp, H, dt);
                TRY(gemm_launch_timed(d, s));
            }
            // bf16 gradient stream: the updated stream IS the operand the out_proj gradients read -> it is written to dA2 (this layer's
            // parity) instead of a separate cast copy, and the attention block's LayerNorm backward below reads it from there
            TRYP("ln_bwd_kernel", 0, (double)M * H * (es + 3 * rs + (rb ? 0 : es)), s,   // dy (operand dtype), x, dx in / out, cast out
                 layernorm_bwd_launch(dtmp, dt == NBCI_BF16 ? 1 : 0, ws + lw.x_mid, params + lo.ln2w, (const float*)(ws + lw.mean2),
                                     (const float*)(ws + lw.rstd2), LnStreams{rb, dx, rb ? (void*)(ws + o_dA2) : (void*)dx, rb}, RG(lo.ln2w),
                                     RG(lo.ln2b), M, H, s, rc,
                                     dt == NBCI_F32 ? LnCast{nullptr, 0, 0u, 1.f, 0u, nullptr}
                                                    : LnCast{rb ? nullptr : ws + o_dA2, 1, 0u, 1.f, 0u, RG(lo.ob)}));
            // ---- attention backward: x_mid = x_in + out_proj(dropout(merge(Pd v)))
            const void* dxc;
            if (dt == NBCI_F32) {
                dxc = dx;
                TRY(colsum_launch(dx, NBCI_F32, H, M, H, RG(lo.ob), s, rc));
            } else {
                dxc = ws + o_dA2;  // bf16 copy + out_proj bias grad came out of the LayerNorm backward above
            }
            TRY(wq.push(H, H, M, op(dxc, es, 0, H, 0), op(ws + lw.ad, es, 0, H, 0), grads + lo.ow, H));
            {   // da = (dx W_o) * keep(attn_out)  -> dB (M, H)
                nbci_gemm_desc d = gd(M, H, H, dt, op(dxc, es, 0, H, 1), op(x.W(lo.ow), es, 0, H, 0), ws + w.dB2, H, dt);
                d.drop_p = p_lay; d.seed = io->seed; d.site = 17 + 4 * l;
                TRY(gemm_launch_timed(d, s));
            }
            const size_t pd = p_lay > 0.f ? lw.Pd : lw.P;
            const int64_t pz1 = (int64_t)nh * Tp * w.ldP, pz2 = (int64_t)Tp * w.ldP;
            const int64_t qz1 = (int64_t)Tp * 3 * H, az1 = (int64_t)Tp * H;
            static const bool attn_bias = measure_env("NBCI_ATTN_BIASGRAD", 1) != 0;   // q/k/v bias sums inside the attention backward (DPP row sums + LDS atomics: +6 us on the two kernels, -11 us colsum launch per layer)
            bool bias_in_attn = false;
            if (p.fused_attn && attn_fused_eligible(dt, Tp, H, nh)) {
                bias_in_attn = attn_bias && !c.use_rope;
                TRYP("attn_bwd_kernel", 10.0 * Tp * Tp * hd * B * nh, (double)M * 8 * H * es, s,   // bytes: q, k, v, out, d out in; dq, dk, dv out
                     attn_bwd_launch(ws + lw.qkv, (const int32_t*)(ws + w.tmask), ws + lw.ad, (const float*)(ws + lw.lse), ws + w.dB2, ws + w.dS, ws + lw.Pd, w.ldP, ws + o_dqkv,
                                     bias_in_attn ? RG(lo.qb) : nullptr, B, nh, Tp, H, c.context_forward, c.context_backward, p_lay,
                                     io->seed, 16 + 4 * l, s, rc));
            } else if (p.flash_attn && Tp >= p.flash_min && fattn_eligible(dt, Tp, H, nh)) {   // (Dsum lives in the score buffer, unused on this path)
                TRY(fattn_masked_bwd_launch(ws + lw.qkv, (const int32_t*)(ws + w.tmask), ws + lw.ad, ws + w.dB2, (const float*)(ws + lw.lse),
                                            (float*)(ws + w.scores), ws + o_dqkv, B, nh, Tp, H, c.context_forward, c.context_backward, p_lay,
                                            io->seed, 16 + 4 * l, s));
            } else {
                {   // dPd = da v^T   (f32, reuses the score buffer)
                    nbci_gemm_desc d = gd(Tp, Tp, hd, dt, op(ws + w.dB2, es, 0, H, 1, 0, 0, az1, hd),
                                          op(ws + lw.qkv, es, 2 * H, 3 * H, 1, 0, 0, qz1, hd), ws + w.scores, w.ldS, NBCI_F32);
                    d.batch = B * nh; d.zdiv = nh; d.czs1 = (int64_t)nh * Tp * w.ldS; d.czs2 = (int64_t)Tp * w.ldS;
                    TRY(gemm_launch_timed(d, s));
                }
                {   // dv = Pd^T da -> dqkv[:, 2H + h*hd ..]
                    nbci_gemm_desc d = gd(Tp, hd, Tp, dt, op(ws + pd, es, 0, w.ldP, 0, 0, 0, pz1, pz2),
                                          op(ws + w.dB2, es, 0, H, 0, 0, 0, az1, hd), (char*)(ws + o_dqkv) + (size_t)2 * H * es, 3 * H, dt);
                    d.batch = B * nh; d.zdiv = nh; d.czs1 = qz1; d.czs2 = hd;
                    if (!c.use_rope) d.colsum = RG(lo.vb); d.colsum_rep_stride = rc.stride; d.colsum_nrep = rc.n;
                    TRY(gemm_launch_timed(d, s));
                }
                TRY(softmax_bwd_launch((const float*)(ws + w.scores), ws + lw.P, ws + w.dS, dt, B, nh, Tp, w.ldS, w.ldP, p_lay,
                                       io->seed, 16 + 4 * l, s));
                {   // dq = dS k * scale
                    nbci_gemm_desc d = gd(Tp, hd, Tp, dt, op(ws + w.dS, es, 0, w.ldP, 1, 0, 0, pz1, pz2),
                                          op(ws + lw.qkv, es, H, 3 * H, 0, 0, 0, qz1, hd), ws + o_dqkv, 3 * H, dt);
                    d.batch = B * nh; d.zdiv = nh; d.czs1 = qz1; d.czs2 = hd; d.alpha = scale;
                    if (!c.use_rope) d.colsum = RG(lo.qb); d.colsum_rep_stride = rc.stride; d.colsum_nrep = rc.n;
                    TRY(gemm_launch_timed(d, s));
                }
                {   // dk = dS^T q * scale
                    nbci_gemm_desc d = gd(Tp, hd, Tp, dt, op(ws + w.dS, es, 0, w.ldP, 0, 0, 0, pz1, pz2),
                                          op(ws + lw.qkv, es, 0, 3 * H, 0, 0, 0, qz1, hd), (char*)(ws + o_dqkv) + (size_t)H * es, 3 * H, dt);
                    d.batch = B * nh; d.zdiv = nh; d.czs1 = qz1; d.czs2 = hd; d.alpha = scale;
                    if (!c.use_rope) d.colsum = RG(lo.kb); d.colsum_rep_stride = rc.stride; d.colsum_nrep = rc.n;
                    TRY(gemm_launch_timed(d, s));
                }
            }
            const bool fused_bwd = (p.fused_attn && attn_fused_eligible(dt, Tp, H, nh)) || (p.flash_attn && Tp >= p.flash_min && fattn_eligible(dt, Tp, H, nh));
            if (c.use_rope) TRY(rope_launch(ws + o_dqkv, dt, (const int64_t*)(ws + w.tts), io->rope_cos, io->rope_sin, M, H, nh, 1, s));
            if ((c.use_rope || fused_bwd) && !bias_in_attn)  // q/k/v bias grads = column sums of dqkv (after the inverse rotation)
                TRY(colsum_launch(ws + o_dqkv, dt, 3 * H, M, 3 * H, RG(lo.qb), s, rc));
            TRY(wq.push(3 * H, H, M, op(ws + o_dqkv, es, 0, 3 * H, 0), op(ws + lw.h1, es, 0, H, 0), grads + lo.qw, H));
            TRY(fork());
            TRY(wq.flush());   // all four operand pairs exist now (aux stream: beside the rest of this layer and the next one)
            if (two) NBCI_CHECK_HIP(hipEventRecord(p.ev_wg[l & 1], aux));
            {
                nbci_gemm_desc d = gd(M, H, 3 * H, dt, op(ws + o_dqkv, es, 0, 3 * H, 1), op(x.W(lo.qw), es, 0, H, 0), dtmp, H, dt);
                TRY(gemm_launch_timed(d, s));
            }
            if (l == 0 && w.phase_ok) NBCI_CHECK_HIP(hipMemsetAsync(ws + w.dAp, 0, w.dAp_rows * H * es, s));   // zero pad rows
            if (two) NBCI_CHECK_HIP(hipStreamWaitEvent(s, p.ev_wg[(l + 1) & 1], 0));   // the cast below rewrites dA[(l - 1) & 1], which layer l + 1's weight gradients read
            TRYP("ln_bwd_kernel", 0, (double)M * H * (es + 3 * rs + es), s,   // dy (operand dtype), x, dx in / out, cast out
                 layernorm_bwd_launch(dtmp, dt == NBCI_BF16 ? 1 : 0, ws + lw.x_in, params + lo.ln1w, (const float*)(ws + lw.mean1),
                                     (const float*)(ws + lw.rstd1), LnStreams{rb, rb ? (const void*)(ws + o_dA2) : (const void*)dx, dx, rb},
                                     RG(lo.ln1w), RG(lo.ln1b), M, H, s, rc, cast_for(l - 1)));
        } else {
            // ---- embedder backward (ndt1.py:160-203)
            const int KS = c.stack_size * D;
            const void* dx0;
            if (!need_cast) {
                dx0 = dx;
                if (io->embed_part != 2) TRY(colsum_launch(dx0, dt, H, M, H, RG(p.stkb), s, rc));
            } else {
                dx0 = ws + w.dA[1];   // (the embedder is "layer -1": parity 1)
            }
            // io->embed_part splits this segment for the data-parallel trainer: 1 = everything that finishes the stack-projection /
            // position / token-table gradients (35 of the segment's 38 MB: their all-reduce then runs beside part 2), 2 = the rest
            // (d pre-activation + the embed_spikes gradients), 0 = both.
            const int part = io->embed_part;
            NBCI_REQUIRE(part >= 0 && part <= 2, NBCI_EINVAL, "ndt1: embed_part must be 0, 1 or 2");
            if (part != 2) {
            if (c.pos) TRYP("posgrad_kernel", 0, (double)Mk * H * rs, s, posgrad_launch(dx, (const int64_t*)(ws + w.tts), grads + p.pos, Mk, H, p_emb, io->seed, 3, s, Tk, npre, xdt));
            if (npre) {   // the prefix tokens' tables: [day, block] order as assembled in the forward
                int k = 0;
                if (c.day_token_days > 0) TRY(prefix_grad_launch(dx, io->day_idx, grads + p.dayemb, B, Tp, k++, H, p_emb, io->seed, 3, s, xdt));
                if (c.block_token_blocks > 0) TRY(prefix_grad_launch(dx, io->block_idx, grads + p.blkemb, B, Tp, k++, H, p_emb, io->seed, 3, s, xdt));
            }
            }
            // the spike-token rows of dx0: all of it, or (prefix tokens) a view that skips the first npre rows of every sample block
            const nbci_operand dx0_km = npre ? op(dx0, es, (int64_t)npre * H, H, 1, Tk, (int64_t)Tp * H) : op(dx0, es, 0, H, 1);
            const nbci_operand dx0_rm = npre ? op(dx0, es, (int64_t)npre * H, H, 0, Tk, (int64_t)Tp * H) : op(dx0, es, 0, H, 0);
            if (w.phase_ok) {
                // dx0 sits in the phase layout (carve): token j of sample b at row npad + b*Q + j, zero rows in between.
                const int st = c.stack_stride, nwin = c.stack_size / st;
                if (part != 2) TRY(fork());
                if (part != 2)   // K = all B*Q rows (the zero rows add nothing): row r of dx0 against the window starting at bin st*r of y, both
                                 // operands plain (the window rows overlap: rpb = -1) - no view stepping in the K loop (187 -> 161 us)
                    TRY(wgrad(wgs, dt, H, KS, B * w.Q, op(ws + w.dAp, es, (int64_t)w.npad * H, H, 0),
                              op(ws + w.y, es, 0, (int64_t)st * D, 0, -1), grads + p.stkw, KS));
                if (part != 1) {
                // d pre-activation WITHOUT the (M, size*D) window-gradient tensor and its col2im pass. The st bins t = st*q + ph of
                // group q collect  sum_{i < nwin} dx0[q - i] . W_s[:, D*(st*i + ph) .. + D]; with n = D*ph + c and i' = nwin-1-i that
                // is ONE GEMM: rows (b, q), N = st*D, a contraction over k = (i', h) of the nwin consecutive padded rows q .. q+nwin-1
                // (an overlapping-row view, K = nwin*H contiguous) against the weight slices taken in reverse order (a row-major-in-k
                // view with a negative group stride). Output row (b, q) IS bins st*q .. st*q+st-1 of d pre-activation; the
                // activation gradient (from the stored output y, same layout) is the gate.
                nbci_gemm_desc d = gd(B * w.Q, st * D, nwin * H, dt, op(ws + w.dAp, es, 0, H, 1),   // (rows b*Q + q .. + nwin - 1: every row one H apart)
                                      op(x.W(p.stkw), es, (int64_t)D * st * (nwin - 1), KS, 0, H, -(int64_t)D * st), ws + w.dpre,
                                      (int64_t)st * D, dt);
                d.gate = ws + w.y; d.ldg = (int64_t)st * D; d.gate_act = 64 + c.embed_act;
                TRY(gemm_launch_timed(d, s));
                }
            } else {
            if (part != 2) TRY(fork());
            if (part != 2)
                TRY(wgrad(wgs, dt, H, KS, Mk, dx0_rm,
                          op(ws + w.y, es, 0, (int64_t)c.stack_stride * D, 0, Tk, (int64_t)T * D), grads + p.stkw, KS));
            if (part != 1) {   // dwin = dx0 W_s  (B*T', S*D)
                nbci_gemm_desc d = gd(Mk, KS, H, dt, dx0_km, op(x.W(p.stkw), es, 0, KS, 0), ws + w.dwin, KS, dt);
                TRY(gemm_launch_timed(d, s));
                TRY(col2im_actgrad_launch(ws + w.dwin, ws + w.y, ws + w.dpre, dt, B, T, Tk, D, c.stack_size, c.stack_stride,
                                          c.embed_act, s));
            }
            }
            if (part == 1) {
                // (the embed_spikes gradients belong to part 2)
            } else if (c.adapt_days > 0) {   // per-sample partial gradients (batched), then a deterministic scatter into the days' rows
                NBCI_REQUIRE(io->day_idx, NBCI_EINVAL, "ndt1: embedder.adapt needs day_idx");
                const int wn = D * c.n_channels;
                nbci_gemm_desc d = gd(D, c.n_channels, T, dt, op(ws + w.dpre, es, 0, D, 0, 0, 0, (int64_t)T * D),
                                      op(ws + w.xs, es, 0, c.n_channels, 0, 0, 0, (int64_t)T * c.n_channels), ws + w.wpart, c.n_channels, NBCI_F32);
                d.batch = B; d.zdiv = 1; d.czs1 = wn;
                TRY(gemm_launch_timed(d, s));
                TRY(adapt_grads_launch(ws + w.dpre, dt, (const float*)(ws + w.wpart), (float*)(ws + w.bpart), io->day_idx, grads + p.embw,
                                       grads + p.embb, p.day_stride, B, T, D, wn, c.adapt_days, s));
            } else {
            TRY(colsum_launch(ws + w.dpre, dt, D, B * T, D, RG(p.embb), s, rc));
            TRY(wgrad(s, dt, D, c.n_channels, B * T, op(ws + w.dpre, es, 0, D, 0), op(ws + w.xs, es, 0, c.n_channels, 0),
                      grads + p.embw, c.n_channels));
            }
        }
    }
    // the replicated small-vector gradients of every segment of this call, folded in ONE launch (their compact ranges are adjacent)
    // A part of segment 0 folds (and so writes) only ITS side of the split: the other side may be in an all-reduce by then.
    int clo = p.cseg[seg_lo].first, chi = p.cseg[seg_hi].second;
    if (seg_lo == 0 && io->embed_part == 1) clo = p.compact_of(p.stkb);
    if (seg_lo == 0 && io->embed_part == 2) chi = p.compact_of(p.stkb);
    // The join of this call's main-stream work onto the aux stream is unconditional: position / prefix / adapt gradients and the main
    // chain's reads of the segment's bf16 weights must happen-before whatever the caller queues on aux next (per-segment AdamW + zero_grad,
    // an all-reduce) even for a segment without 1-D parameters, i.e. with nothing to fold.
    TRY(fork());
    if (chi > clo) {
        TRYP("fold_replicas_kernel", 0, (double)(chi - clo) * (rc.n + 1) * 4, aux, fold_replicas_launch(rep, rc.stride, rc.n, p.d_flat_of, clo, chi, grads, aux));
    }
    return NBCI_OK;
}

}  // namespace nbci

// ---------------------------------------------------------------------------------------------
using namespace nbci;

extern "C" {

int nbci_ndt1_plan_create(const nbci_ndt1_config* cfg, nbci_ndt1_plan* out) {
    if (!cfg || !out) return fail(NBCI_EINVAL, "plan_create: null argument");
    const nbci_ndt1_config& c = *cfg;
    NBCI_REQUIRE(c.hidden > 0 && c.n_heads > 0 && c.hidden % c.n_heads == 0, NBCI_ESHAPE,
                 "Hidden dim is not multiple of head size");  // ndt1.py:242
    NBCI_REQUIRE(c.hidden % 8 == 0 && c.inter % 8 == 0 && c.input_dim % 8 == 0, NBCI_ESHAPE,
                 "hidden, inter_size and input_dim must be multiples of 8");
    NBCI_REQUIRE((c.hidden / c.n_heads) % 8 == 0, NBCI_ESHAPE, "head size must be a multiple of 8");
    NBCI_REQUIRE(c.n_channels > 0 && c.stack_size > 0 && c.stack_stride > 0 && c.vocab > 0 && c.n_layers >= 0, NBCI_ESHAPE,
                 "bad NDT1 shape parameters");
    NBCI_REQUIRE(c.dtype == NBCI_F32 || c.dtype == NBCI_BF16, NBCI_EINVAL, "dtype must be f32 or bf16");
    NBCI_REQUIRE(c.residual_dtype == NBCI_F32 || (c.residual_dtype == NBCI_BF16 && c.dtype == NBCI_BF16), NBCI_EINVAL,
                 "residual_dtype must be f32, or bf16 together with dtype bf16");
    NBCI_REQUIRE(c.adapt_days >= 0 && c.adapt_days <= 4096, NBCI_EINVAL, "adapt_days out of range");
    NBCI_REQUIRE(c.day_token_days >= 0 && c.block_token_blocks >= 0, NBCI_EINVAL, "token table sizes must be >= 0");
    NBCI_REQUIRE(!(c.use_rope && (c.day_token_days > 0 || c.block_token_blocks > 0)), NBCI_EINVAL,
                 "rope with day / block tokens: the reference hands T' timestamps to T'+n tokens (ndt1.py:181,441) and fails; not supported");
    NBCI_REQUIRE(c.factors_size >= 0 && c.factors_size % 8 == 0, NBCI_ESHAPE, "factors size must be a multiple of 8 (0 = no factors projection)");
    NBCI_REQUIRE(c.blank_id >= 0 && c.blank_id < c.vocab, NBCI_EINVAL, "blank_id out of range");
    NBCI_REQUIRE(!(c.use_rope && ((c.hidden / c.n_heads) % 2)), NBCI_ESHAPE, "rope needs an even head size");
    Plan* p = new Plan();
    p->c = c;
    build_layout(*p);
    p->d_taps = nullptr;
    p->ntaps = 0;
    {
        const char* e = getenv("NBCI_FUSED_ATTN");
        p->fused_attn = !(e && e[0] == '0');
        const char* e2 = getenv("NBCI_FLASH_ATTN");
        p->flash_attn = !(e2 && e2[0] == '0');
        const char* e3 = getenv("NBCI_FLASH_MIN_TOKENS");
        p->flash_min = e3 ? atoi(e3) : 1;
    }
    if (c.smooth_sd > 0.f) {
        // scipy.signal.gaussian(1 + 6*sd, sd) normalised, built in float64 (ndt1.py:87-88)
        const int n = 1 + (int)(6 * c.smooth_sd);
        if (n > 64) { delete p; return fail(NBCI_ESHAPE, "smooth_sd too large (max 64 taps)"); }
        std::vector<double> w(n);
        double sum = 0;
        for (int i = 0; i < n; ++i) { const double k = i - (n - 1) / 2.0; w[i] = std::exp(-0.5 * (k / c.smooth_sd) * (k / c.smooth_sd)); sum += w[i]; }
        std::vector<float> wf(n);
        for (int i = 0; i < n; ++i) wf[i] = (float)(w[i] / sum);
        hipError_t e = hipMalloc(&p->d_taps, 64 * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(p->d_taps, wf.data(), n * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) { delete p; return fail(NBCI_EHIP, std::string("plan_create: ") + hipGetErrorString(e)); }
        p->ntaps = n;
    }
    p->d_flat_of = nullptr;
    {
        hipError_t e = hipMalloc(&p->d_flat_of, std::max<size_t>(4, p->flat_of.size() * sizeof(int)));
        if (e == hipSuccess && !p->flat_of.empty())
            e = hipMemcpy(p->d_flat_of, p->flat_of.data(), p->flat_of.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (p->d_taps) (void)hipFree(p->d_taps); delete p; return fail(NBCI_EHIP, std::string("plan_create: ") + hipGetErrorString(e)); }
    }
    *out = (nbci_ndt1_plan)p;
    return NBCI_OK;
}

void nbci_ndt1_plan_destroy(nbci_ndt1_plan plan) {
    Plan* p = (Plan*)plan;
    if (!p) return;
    if (p->d_taps) (void)hipFree(p->d_taps);
    if (p->d_flat_of) (void)hipFree(p->d_flat_of);
    if (p->ev_main) (void)hipEventDestroy(p->ev_main);
    for (hipEvent_t e : p->ev_wg) if (e) (void)hipEventDestroy(e);
    delete p;
}

int64_t nbci_ndt1_param_count(nbci_ndt1_plan plan) { return plan ? ((Plan*)plan)->total : -1; }
int32_t nbci_ndt1_num_params(nbci_ndt1_plan plan) { return plan ? (int32_t)((Plan*)plan)->params.size() : -1; }
int32_t nbci_ndt1_num_segments(nbci_ndt1_plan plan) { return plan ? (int32_t)((Plan*)plan)->seg.size() : -1; }

int nbci_ndt1_param_info(nbci_ndt1_plan plan, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel,
                         int32_t* rows, int32_t* cols, int32_t* segment) {
    Plan* p = (Plan*)plan;
    if (!p || index < 0 || index >= (int)p->params.size()) return fail(NBCI_EINVAL, "param_info: bad plan/index");
    const PInfo& i = p->params[index];
    if (name && name_cap > 0) { strncpy(name, i.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = i.off;
    if (numel) *numel = i.numel;
    if (rows) *rows = i.rows;
    if (cols) *cols = i.cols;
    if (segment) *segment = i.seg;
    return NBCI_OK;
}

int nbci_ndt1_segment_range(nbci_ndt1_plan plan, int32_t seg, int64_t* begin, int64_t* end) {
    Plan* p = (Plan*)plan;
    if (!p || seg < 0 || seg >= (int)p->seg.size()) return fail(NBCI_EINVAL, "segment_range: bad plan/segment");
    *begin = p->seg[seg].first; *end = p->seg[seg].second;
    return NBCI_OK;
}

int64_t nbci_ndt1_workspace_bytes(nbci_ndt1_plan plan, int32_t B, int32_t T, int32_t S) {
    Plan* p = (Plan*)plan;
    if (!p) { fail(NBCI_EINVAL, "workspace_bytes: null plan"); return -1; }
    WS w;
    if (carve(*p, B, T, S, w) != NBCI_OK) return -1;
    return (int64_t)w.bytes;
}

int32_t nbci_ndt1_tokens(nbci_ndt1_plan plan, int32_t T) {
    Plan* p = (Plan*)plan;
    if (!p || T < p->c.stack_size) return -1;
    return 1 + (T - p->c.stack_size) / p->c.stack_stride;
}

int nbci_ndt1_forward(nbci_ndt1_plan plan, const float* params, const void* params_lp, const nbci_ndt1_io* io,
                      nbci_stream_t stream) {
    if (!plan) return fail(NBCI_EINVAL, "forward: null plan");
    return ndt1_forward(*(Plan*)plan, params, params_lp, io, (hipStream_t)stream);
}

int nbci_ndt1_backward(nbci_ndt1_plan plan, const float* params, const void* params_lp, const nbci_ndt1_io* io, float* grads,
                       int32_t seg_hi, int32_t seg_lo, nbci_stream_t stream) {
    if (!plan) return fail(NBCI_EINVAL, "backward: null plan");
    return ndt1_backward(*(Plan*)plan, params, params_lp, io, grads, seg_hi, seg_lo, (hipStream_t)stream);
}

}  // extern "C"
