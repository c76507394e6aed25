// patchtst.hip — host-side orchestration of the PatchTST forward/backward (ctc and mlm heads) on one stream.
//
// Replaces PatchTSTForSpikingActivity.forward (models/patchtst.py:214-255) + PredictHead / PretrainHead and the HF
// PatchTSTModel encoder underneath (patchtst.py:8,176), and their autograd graph. Rows are (batch, channel, patch):
// M = B*C*P rows of d_model features; attention runs over the P patches of one (batch, channel) pair.
// Same conventions as ndt1.hip / itransformer.hip: flat f32 parameters (+ bf16 shadow), caller-owned workspace,
// gradient segments [embedding | layer 0..L-1 | head]. The residual stream stays f32; BatchNorm writes the GEMM operand
// dtype. BatchNorm running statistics live in the caller's `aux` buffer and are updated by train-mode forwards.
#include <cmath>
#include <cstdlib>

#include "plan_common.h"

namespace nbci {

struct PtLayerOff {
    int64_t n1w, n1b, qw, kw, vw, qb, kb, vb, ow, ob, n3w, n3b, f0w, f0b, f3w, f3b;
};

struct PtPlan {
    nbci_ptst_config c;
    std::vector<PInfo> params;
    std::vector<PtLayerOff> L;
    int64_t embw, embb, d0w, d0b, d2w, d2b;
    int64_t total;
    int P, start, nout;
    std::vector<std::pair<int64_t, int64_t>> seg;
    std::vector<int> flat_of;
    std::vector<std::pair<int, int>> cseg;
    std::vector<std::pair<int64_t, int>> cmap;
    int* d_flat_of;
    int compact_total;
    int compact_of(int64_t flat_off) const {
        for (size_t i = 0; i < cmap.size(); ++i) if (cmap[i].first == flat_off) return cmap[i].second;
        return -1;
    }
};

static int64_t pt_add(PtPlan& p, int64_t& cur, const std::string& name, int rows, int cols, int seg) {
    cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
    const int64_t off = cur;
    p.params.push_back({name, off, (int64_t)rows * (cols > 0 ? cols : 1), rows, cols, seg});
    cur += (int64_t)rows * (cols > 0 ? cols : 1);
    return off;
}

static void pt_layout(PtPlan& p) {
    const auto& c = p.c;
    const int D = c.d_model, F = c.ffn_dim, pl = c.patch_length;
    int64_t cur = 0;
    p.embw = pt_add(p, cur, "encoder.encoder.embedder.input_embedding.weight", D, pl, 0);
    p.embb = pt_add(p, cur, "encoder.encoder.embedder.input_embedding.bias", D, 0, 0);
    cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
    p.seg.push_back({0, cur});
    for (int l = 0; l < c.num_hidden_layers; ++l) {
        const int64_t begin = cur;
        const std::string pre = "encoder.encoder.layers." + std::to_string(l) + ".";
        PtLayerOff o;
        o.n1w = pt_add(p, cur, pre + "norm_sublayer1.batchnorm.weight", D, 0, l + 1);
        o.n1b = pt_add(p, cur, pre + "norm_sublayer1.batchnorm.bias", D, 0, l + 1);
        // q/k/v contiguous: the three projections run as ONE [3D][D] GEMM
        o.qw = pt_add(p, cur, pre + "self_attn.q_proj.weight", D, D, l + 1);
        o.kw = pt_add(p, cur, pre + "self_attn.k_proj.weight", D, D, l + 1);
        o.vw = pt_add(p, cur, pre + "self_attn.v_proj.weight", D, D, l + 1);
        o.qb = pt_add(p, cur, pre + "self_attn.q_proj.bias", D, 0, l + 1);
        o.kb = pt_add(p, cur, pre + "self_attn.k_proj.bias", D, 0, l + 1);
        o.vb = pt_add(p, cur, pre + "self_attn.v_proj.bias", D, 0, l + 1);
        o.ow = pt_add(p, cur, pre + "self_attn.out_proj.weight", D, D, l + 1);
        o.ob = pt_add(p, cur, pre + "self_attn.out_proj.bias", D, 0, l + 1);
        o.n3w = pt_add(p, cur, pre + "norm_sublayer3.batchnorm.weight", D, 0, l + 1);
        o.n3b = pt_add(p, cur, pre + "norm_sublayer3.batchnorm.bias", D, 0, l + 1);
        o.f0w = pt_add(p, cur, pre + "ff.0.weight", F, D, l + 1);
        o.f0b = pt_add(p, cur, pre + "ff.0.bias", F, 0, l + 1);
        o.f3w = pt_add(p, cur, pre + "ff.3.weight", D, F, l + 1);
        o.f3b = pt_add(p, cur, pre + "ff.3.bias", D, 0, l + 1);
        cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
        p.L.push_back(o);
        p.seg.push_back({begin, cur});
    }
    const int64_t begin = cur;
    const int hs = c.num_hidden_layers + 1;
    p.nout = c.method == NBCI_PTST_CTC ? c.vocab : pl;
    p.d2w = p.d2b = -1;
    if (c.mlp_decoder) {
        p.d0w = pt_add(p, cur, "decoder.projection.0.weight", D, D, hs);
        p.d0b = pt_add(p, cur, "decoder.projection.0.bias", D, 0, hs);
        p.d2w = pt_add(p, cur, "decoder.projection.2.weight", p.nout, D, hs);
        p.d2b = pt_add(p, cur, "decoder.projection.2.bias", p.nout, 0, hs);
    } else {
        p.d0w = pt_add(p, cur, "decoder.projection.weight", p.nout, D, hs);
        p.d0b = pt_add(p, cur, "decoder.projection.bias", p.nout, 0, hs);
    }
    cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
    p.seg.push_back({begin, cur});
    p.total = cur;
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

struct PtLayerWS {
    size_t x_in, mean1, rstd1, y1, qkv, P, Pd, ad, lse, x_mid, mean3, rstd3, y3, u, g;
};
struct PtWS {
    size_t mask, mask2, xm, patch;
    std::vector<PtLayerWS> L;
    size_t x_last, pooled, d1, logits, alpha, dlogits, argmax, tlens, pred, dpred, scores, bnpart, bnsums;
    size_t dx, dtmp, cA, cA2, dU, dAtt, dqkv, dS, dpool, poolpart, dsum, rep;
    size_t y1q, y1s, wq8, wq8s;   // fp8_qkv: MX e4m3 copy of the BatchNorm output (M x D bytes + M x D/32 scale bytes), quantised q/k/v weights
    size_t bytes;
    long long M;
    int Mh, ldS, ldP, vpad, ldp;
    bool small_attn, flash;
};

static int pt_carve(const PtPlan& p, int B, int S, PtWS& w) {
    const auto& c = p.c;
    NBCI_REQUIRE(B > 0, NBCI_ESHAPE, "patchtst: B must be positive");
    const size_t es = c.dtype == NBCI_BF16 ? 2 : 4;
    const size_t rs = c.residual_dtype == NBCI_BF16 ? 2 : 4;   // the residual stream (saved BatchNorm inputs) and its gradient stream
    const size_t C = c.num_input_channels, P = p.P, D = c.d_model, F = c.ffn_dim, pl = c.patch_length, nh = c.num_attention_heads;
    const size_t M = (size_t)B * C * P;
    NBCI_REQUIRE(M * F < (1ull << 32) && (size_t)B * C * nh * P * P < (1ull << 32), NBCI_ESHAPE,
                 "patchtst: batch too large for the 32-bit dropout counters (split the batch)");
    NBCI_REQUIRE(P <= 2048, NBCI_ESHAPE, "patchtst: at most 2048 patches");
    w.M = (long long)M;
    w.Mh = (int)(c.method == NBCI_PTST_CTC ? (size_t)B * P : M);
    w.ldS = ((int)P + 3) / 4 * 4;
    w.ldP = ((int)P + 7) / 8 * 8;
    w.vpad = (c.vocab + 7) / 8 * 8;
    w.ldp = ((int)pl + 7) / 8 * 8;
    size_t cur = 0;
    w.mask = bump(cur, (size_t)B * C * P);
    w.mask2 = bump(cur, (size_t)B * C * P);
    w.xm = bump(cur, M * pl * 4);
    w.patch = bump(cur, M * pl * 4);
    w.L.resize(c.num_hidden_layers);
    w.flash = fattn_eligible(c.dtype, (int)P, (int)D, (int)nh);
    w.small_attn = w.flash || sattn_eligible(c.dtype, (int)P, (int)D, (int)nh);   // no score / probability tensors on that path
    const size_t nP = w.small_attn ? 0 : (size_t)B * C * nh * P * w.ldP;
    const size_t nstat = sattn_stat_floats(B * (int)C, (int)nh, (int)P);
    for (auto& l : w.L) {
        l.x_in = bump(cur, M * D * rs);
        l.mean1 = bump(cur, D * 4); l.rstd1 = bump(cur, D * 4);
        l.y1 = bump(cur, M * D * es);
        l.qkv = bump(cur, M * 3 * D * es);
        l.P = bump(cur, nP * es);
        l.Pd = bump(cur, nP * es);
        l.ad = bump(cur, M * D * es);
        l.lse = bump(cur, nstat * 4);
        l.x_mid = bump(cur, M * D * rs);
        l.mean3 = bump(cur, D * 4); l.rstd3 = bump(cur, D * 4);
        l.y3 = bump(cur, M * D * es);
        l.u = bump(cur, M * F * es);
        l.g = bump(cur, M * F * es);
    }
    w.x_last = bump(cur, M * D * rs);
    const size_t Mh = w.Mh;
    w.pooled = bump(cur, Mh * D * es);
    w.d1 = bump(cur, Mh * D * es);
    w.logits = bump(cur, (size_t)B * P * w.vpad * 4);
    w.alpha = bump(cur, ctc_alpha_floats(B, (int)P, S > 0 ? S : 1) * 4);
    w.dlogits = bump(cur, (size_t)B * P * w.vpad * es);
    w.argmax = bump(cur, (size_t)B * P * 4);
    w.tlens = bump(cur, (size_t)B * 4);
    w.pred = bump(cur, M * w.ldp * 4);
    w.dpred = bump(cur, M * w.ldp * es);
    w.scores = bump(cur, w.small_attn ? 0 : (size_t)B * C * nh * P * w.ldS * 4);
    w.dsum = bump(cur, nstat * 4);
    w.bnpart = bump(cur, bn_partial_floats((long long)M, (int)D) * 4);
    w.bnsums = bump(cur, 3 * D * 4);   // per-column coefficients of the BatchNorm backward's second pass
    w.dx = bump(cur, M * D * rs);
    w.dtmp = bump(cur, M * D * 4);   // (f32-sized: also the f32 d(embedding output) of the patch embedding's backward)
    w.cA = bump(cur, M * D * es);
    w.cA2 = bump(cur, M * D * es);
    w.dU = bump(cur, M * F * es);
    w.dAtt = bump(cur, M * D * es);
    w.dqkv = bump(cur, M * 3 * D * es);
    w.dS = bump(cur, nP * es);
    w.dpool = bump(cur, Mh * D * 4);
    w.poolpart = bump(cur, c.method == NBCI_PTST_CTC ? ptst_pool_partial_floats(B, (int)P, (int)D) * 4 : 0);
    w.rep = bump(cur, (size_t)NREP * p.compact_total * 4);
    if (c.fp8_qkv) {
        NBCI_REQUIRE(c.dtype == NBCI_BF16 && D % 128 == 0, NBCI_ESHAPE, "patchtst: fp8_qkv needs the bf16 path and d_model in multiples of 128");
        w.y1q = bump(cur, M * D); w.y1s = bump(cur, M * D / 32);
        w.wq8 = bump(cur, 3 * D * D); w.wq8s = bump(cur, 3 * D * D / 32);
    }
    w.bytes = (cur + 255) / 256 * 256;
    return NBCI_OK;
}

static int pt_validate(const PtPlan& p, const nbci_ptst_io* io) {
    NBCI_REQUIRE(io, NBCI_EINVAL, "patchtst: null io");
    NBCI_REQUIRE(io->spikes && io->spikes_mask && io->aux && io->nbt, NBCI_EINVAL, "patchtst: spikes, spikes_mask, aux and nbt are required");
    NBCI_REQUIRE(io->workspace, NBCI_EWORKSPACE, "patchtst: null workspace");
    NBCI_REQUIRE(((uintptr_t)io->workspace) % 256 == 0 && ((uintptr_t)io->aux) % 16 == 0, NBCI_EALIGN, "patchtst: workspace / aux misaligned");
    if (p.c.method == NBCI_PTST_CTC) NBCI_REQUIRE(io->spikes_lengths, NBCI_EINVAL, "patchtst ctc: spikes_lengths required");
    return NBCI_OK;
}

__global__ void pt_nbt_kernel(long long* nbt, int n) {
    if ((int)threadIdx.x < n) nbt[threadIdx.x] += 1;
}

struct PtAux {   // views into io->aux
    float* pos;
    float* base;
    int D;
    float* rm1(int l) const { return base + (size_t)l * 4 * D; }
    float* rv1(int l) const { return base + (size_t)l * 4 * D + D; }
    float* rm3(int l) const { return base + (size_t)l * 4 * D + 2 * D; }
    float* rv3(int l) const { return base + (size_t)l * 4 * D + 3 * D; }
};

int ptst_forward(const PtPlan& p, const float* params, const void* params_lp, const nbci_ptst_io* io, hipStream_t s) {
    TRY(pt_validate(p, io));
    const auto& c = p.c;
    NBCI_REQUIRE(params, NBCI_EINVAL, "patchtst: null params");
    NBCI_REQUIRE(c.dtype == NBCI_F32 || params_lp, NBCI_EINVAL, "patchtst: bf16 mode needs the bf16 parameter shadow");
    NBCI_REQUIRE(io->preds && io->loss && io->n_examples, NBCI_EINVAL, "patchtst: preds, loss, n_examples outputs are required");
    const int B = io->B, S = io->S;
    PtWS w;
    TRY(pt_carve(p, B, S, w));
    NBCI_REQUIRE((size_t)io->workspace_bytes >= w.bytes, NBCI_EWORKSPACE, "patchtst: workspace too small");
    const int C = c.num_input_channels, T = c.context_length, P = p.P, D = c.d_model, F = c.ffn_dim, pl = c.patch_length,
              nh = c.num_attention_heads, hd = D / nh, L = c.num_hidden_layers;
    const long long M = w.M;
    const int Mi = (int)M;
    const int dt = c.dtype;
    const int xdt = c.residual_dtype;   // storage of the residual stream between kernels (every kernel widens it and computes in f32)
    const bool rb = xdt == NBCI_BF16;
    const size_t es = dt == NBCI_BF16 ? 2 : 4;
    const void* pw = dt == NBCI_BF16 ? params_lp : (const void*)params;
    auto W = [&](int64_t off) -> const void* { return (const char*)pw + off * (int64_t)es; };
    const bool train = io->train != 0;
    const float pa = train ? c.attention_dropout : 0.f, pp = train ? c.path_dropout : 0.f, pf = train ? c.ff_dropout : 0.f,
                ppos = train ? c.positional_dropout : 0.f;
    char* ws = (char*)io->workspace;
    const PtAux aux{io->aux, io->aux + (size_t)P * D, D};

    if (io->want_grad) NBCI_CHECK_HIP(hipMemsetAsync(ws + w.rep, 0, (size_t)NREP * p.compact_total * 4, s));
    // 0. mask + patchify + embedding (PatchTSTModel.forward: scaler(NOP) -> patchifier -> masking -> embedder -> positions)
    const uint8_t* mask = nullptr;
    if (c.do_mask_input) {
        if (io->ext_mask) mask = io->ext_mask;
        else {
            TRY(ptst_mask_launch((uint8_t*)(ws + w.mask), B, C, P, c.random_mask_ratio, c.channel_consistent_masking, io->seed, 6, s));
            mask = (const uint8_t*)(ws + w.mask);
        }
    }
    float* patch = io->patch_input ? io->patch_input : (float*)(ws + w.patch);
    TRY(ptst_patchify_launch(io->spikes, patch, (float*)(ws + w.xm), mask, B, T, C, P, pl, c.patch_stride, p.start, c.mask_value, s));
    void* x_cur = ws + (L ? w.L[0].x_in : w.x_last);
    TRY(ptst_embed_launch((const float*)(ws + w.xm), params + p.embw, params + p.embb, aux.pos, x_cur, M, P, pl, D, ppos, io->seed, 4, s, xdt));
    const float scale = 1.0f / sqrtf((float)hd);
    for (int l = 0; l < L; ++l) {
        const PtLayerWS& lw = w.L[l];
        const PtLayerOff& lo = p.L[l];
        void* x_in = ws + lw.x_in;
        void* x_mid = ws + lw.x_mid;
        void* x_out = ws + (l + 1 < L ? w.L[l + 1].x_in : w.x_last);
        TRY(batchnorm_fwd_launch(x_in, params + lo.n1w, params + lo.n1b, aux.rm1(l), aux.rv1(l), train, c.norm_eps, ws + lw.y1, dt,
                                 (float*)(ws + lw.mean1), (float*)(ws + lw.rstd1), (float*)(ws + w.bnpart), M, D, s,
                                 c.fp8_qkv ? ws + w.y1q : nullptr, c.fp8_qkv ? ws + w.y1s : nullptr, xdt));
        if (c.fp8_qkv) {   // q / k / v on the block-scaled fp8 matrix instruction (fp8.hip); y1 stays in bf16 for the weight gradient
            TRY(mx_quantize_launch(params + lo.qw, NBCI_F32, D, ws + w.wq8, ws + w.wq8s, 3 * D, D, s));   // from the f32 master weights
            TRY(gemm_fp8_launch(ws + w.y1q, ws + w.y1s, ws + w.wq8, ws + w.wq8s, params + lo.qb, ws + lw.qkv, dt, M, 3 * D, D, 3 * D, s));
        } else {
            nbci_gemm_desc d = gd(Mi, 3 * D, D, dt, op(ws + lw.y1, es, 0, D, 1), op(W(lo.qw), es, 0, D, 1), ws + lw.qkv, 3 * D, dt);
            d.bias = params + lo.qb;
            TRY(gemm_launch_timed(d, s));
        }
        if (w.small_attn) {
            if (w.flash) TRY(fattn_fwd_launch(ws + lw.qkv, ws + lw.ad, (float*)(ws + lw.lse), B * C, nh, P, D, pa, io->seed, 16 + 4 * l, s));
            else TRY(sattn_fwd_launch(ws + lw.qkv, ws + lw.ad, (float*)(ws + lw.lse), dt, B * C, nh, P, D, pa, io->seed, 16 + 4 * l, s));
        } else {
        {   // scores = q k^T / sqrt(hd), batched over (b, c, head)
            nbci_gemm_desc d = gd(P, P, hd, dt, op(ws + lw.qkv, es, 0, 3 * D, 1, 0, 0, (int64_t)P * 3 * D, hd),
                                  op(ws + lw.qkv, es, D, 3 * D, 1, 0, 0, (int64_t)P * 3 * D, hd), ws + w.scores, w.ldS, NBCI_F32);
            d.batch = B * C * nh; d.zdiv = nh; d.czs1 = (int64_t)nh * P * w.ldS; d.czs2 = (int64_t)P * w.ldS; d.alpha = scale;
            TRY(gemm_launch_timed(d, s));
        }
        TRY(softmax_fwd_launch((const float*)(ws + w.scores), ws + lw.P, ws + (pa > 0.f ? lw.Pd : lw.P), dt, nullptr, B * C, nh, P, w.ldS, w.ldP,
                               -2, -2, pa, io->seed, 16 + 4 * l, s));
        {
            const size_t pd = pa > 0.f ? lw.Pd : lw.P;
            nbci_gemm_desc d = gd(P, hd, P, dt, op(ws + pd, es, 0, w.ldP, 1, 0, 0, (int64_t)nh * P * w.ldP, (int64_t)P * w.ldP),
                                  op(ws + lw.qkv, es, 2 * D, 3 * D, 0, 0, 0, (int64_t)P * 3 * D, hd), ws + lw.ad, D, dt);
            d.batch = B * C * nh; d.zdiv = nh; d.czs1 = (int64_t)P * D; d.czs2 = hd;
            TRY(gemm_launch_timed(d, s));
        }
        }
        {   // x_mid = x_in + path_dropout(out_proj(a))
            nbci_gemm_desc d = gd(Mi, D, D, dt, op(ws + lw.ad, es, 0, D, 1), op(W(lo.ow), es, 0, D, 1), x_mid, D, xdt);
            d.bias = params + lo.ob; d.drop_p = pp; d.seed = io->seed; d.site = 17 + 4 * l; d.residual = x_in; d.ldr = D; d.residual_dtype = xdt;
            TRY(gemm_launch_timed(d, s));
        }
        TRY(batchnorm_fwd_launch(x_mid, params + lo.n3w, params + lo.n3b, aux.rm3(l), aux.rv3(l), train, c.norm_eps, ws + lw.y3, dt,
                                 (float*)(ws + lw.mean3), (float*)(ws + lw.rstd3), (float*)(ws + w.bnpart), M, D, s, nullptr, nullptr, xdt));
        {   // g = ff_dropout(act(ff.0(y3))); lw.u keeps act'(u) for the backward gate
            nbci_gemm_desc d = gd(Mi, F, D, dt, op(ws + lw.y3, es, 0, D, 1), op(W(lo.f0w), es, 0, D, 1), ws + lw.g, F, dt);
            d.bias = params + lo.f0b; d.act = c.act; d.C2 = ws + lw.u; d.c2_grad = 1; d.drop_p = pf; d.seed = io->seed; d.site = 18 + 4 * l;
            TRY(gemm_launch_timed(d, s));
        }
        {
            nbci_gemm_desc d = gd(Mi, D, F, dt, op(ws + lw.g, es, 0, F, 1), op(W(lo.f3w), es, 0, F, 1), x_out, D, xdt);
            d.bias = params + lo.f3b; d.drop_p = pp; d.seed = io->seed; d.site = 19 + 4 * l; d.residual = x_mid; d.ldr = D; d.residual_dtype = xdt;
            TRY(gemm_launch_timed(d, s));
        }
    }
    if (train && L > 0) hipLaunchKernelGGL(pt_nbt_kernel, dim3(1), dim3(256), 0, s, (long long*)io->nbt, 2 * L);
    const void* h = ws + w.x_last;
    if (io->hidden_out) {   // (B,C,P,D) f32 whatever the stream's storage
        if (rb) TRY(dropcast2d_launch(h, io->hidden_out, NBCI_F32, Mi, D, 0.f, 0, 0, nullptr, s, RepCfg{0, 1}, NBCI_BF16));
        else NBCI_CHECK_HIP(hipMemcpyAsync(io->hidden_out, h, (size_t)M * D * 4, hipMemcpyDeviceToDevice, s));
    }
    if (c.method == NBCI_PTST_CTC) {
        const int Mh = w.Mh;
        TRY(ptst_pool_fwd_launch(h, ws + w.pooled, dt, B, C, P, D, s, xdt, (float*)(ws + w.poolpart)));
        const void* src = ws + w.pooled;
        int64_t ow = p.d0w, ob = p.d0b;
        if (c.mlp_decoder) {
            nbci_gemm_desc d = gd(Mh, D, D, dt, op(src, es, 0, D, 1), op(W(p.d0w), es, 0, D, 1), ws + w.d1, D, dt);
            d.bias = params + p.d0b; d.act = c.dec_act; d.C2 = ws + w.cA; d.c2_grad = 1;   // cA (>= Mh rows) keeps act'(u) until the head backward
            TRY(gemm_launch_timed(d, s));
            src = ws + w.d1; ow = p.d2w; ob = p.d2b;
        }
        {
            nbci_gemm_desc d = gd(Mh, c.vocab, D, dt, op(src, es, 0, D, 1), op(W(ow), es, 0, D, 1), ws + w.logits, w.vpad, NBCI_F32);
            d.bias = params + ob;
            TRY(gemm_launch_timed(d, s));
        }
        int32_t* amax = io->argmax ? io->argmax : (int32_t*)(ws + w.argmax);
        TRY(logsoftmax_launch((const float*)(ws + w.logits), w.vpad, io->preds, amax, Mh, c.vocab, s));
        TRY(ptst_lens_launch(io->spikes_lengths, (int32_t*)(ws + w.tlens), B, pl, c.patch_stride, s));
        if (io->targets) {
            NBCI_REQUIRE(io->targets_lengths && S > 0, NBCI_EINVAL, "patchtst ctc: targets need targets_lengths and S > 0");
            TRY(ctc_launch(io->preds, io->targets, (const int32_t*)(ws + w.tlens), io->targets_lengths, B, P, c.vocab, S, c.blank_id,
                           c.zero_infinity, io->loss, (float*)(ws + w.alpha), io->want_grad ? ws + w.dlogits : nullptr, dt, w.vpad,
                           io->grad_scale, s));
        }
    } else {
        NBCI_REQUIRE(mask, NBCI_EINVAL, "Can't pretrain with inactive masking");   // patchtst.py:193
        // PretrainHead operands must be in the GEMM dtype: BatchNorm-free cast of the last hidden state (a bf16 stream IS in that dtype)
        if (!rb) TRY(cast_launch((const float*)h, ws + w.pooled, dt, M * D, s));
        const void* src = rb ? h : (const void*)(ws + w.pooled);
        int64_t ow = p.d0w, ob = p.d0b;
        if (c.mlp_decoder) {
            nbci_gemm_desc d = gd(Mi, D, D, dt, op(src, es, 0, D, 1), op(W(p.d0w), es, 0, D, 1), ws + w.d1, D, dt);
            d.bias = params + p.d0b; d.act = c.dec_act; d.C2 = ws + w.cA; d.c2_grad = 1;
            TRY(gemm_launch_timed(d, s));
            src = ws + w.d1; ow = p.d2w; ob = p.d2b;
        }
        {
            nbci_gemm_desc d = gd(Mi, pl, D, dt, op(src, es, 0, D, 1), op(W(ow), es, 0, D, 1), ws + w.pred, w.ldp, NBCI_F32);
            d.bias = params + ob;
            TRY(gemm_launch_timed(d, s));
        }
        uint8_t* mo = io->mask_out ? io->mask_out : (uint8_t*)(ws + w.mask2);   // never aliases the model mask being read
        TRY(ptst_mlm_loss_launch((const float*)(ws + w.pred), w.ldp, patch, mask, io->spikes_mask, io->preds, mo,
                                 io->want_grad ? ws + w.dpred : nullptr, dt, io->loss, io->n_examples, B, T, C, P, pl, c.patch_stride, c.loss,
                                 io->grad_scale, s));
    }
    return NBCI_OK;
}

int ptst_backward(const PtPlan& p, const float* params, const void* params_lp, const nbci_ptst_io* io, float* grads, int seg_hi, int seg_lo,
                  hipStream_t s) {
    TRY(pt_validate(p, io));
    const auto& c = p.c;
    NBCI_REQUIRE(params && grads, NBCI_EINVAL, "patchtst: null params/grads");
    NBCI_REQUIRE(c.dtype == NBCI_F32 || params_lp, NBCI_EINVAL, "patchtst: bf16 mode needs the bf16 parameter shadow");
    NBCI_REQUIRE(seg_hi <= c.num_hidden_layers + 1 && seg_lo >= 0 && seg_lo <= seg_hi, NBCI_EINVAL, "patchtst: bad segment range");
    const int B = io->B, S = io->S;
    PtWS w;
    TRY(pt_carve(p, B, S, w));
    NBCI_REQUIRE((size_t)io->workspace_bytes >= w.bytes, NBCI_EWORKSPACE, "patchtst: workspace too small");
    const int C = c.num_input_channels, P = p.P, D = c.d_model, F = c.ffn_dim, pl = c.patch_length, nh = c.num_attention_heads, hd = D / nh,
              L = c.num_hidden_layers;
    const long long M = w.M;
    const int Mi = (int)M;
    const int dt = c.dtype;
    const size_t es = dt == NBCI_BF16 ? 2 : 4;
    const void* pw = dt == NBCI_BF16 ? params_lp : (const void*)params;
    auto W = [&](int64_t off) -> const void* { return (const char*)pw + off * (int64_t)es; };
    const bool train = io->train != 0;
    const float pa = train ? c.attention_dropout : 0.f, pp = train ? c.path_dropout : 0.f, pf = train ? c.ff_dropout : 0.f,
                ppos = train ? c.positional_dropout : 0.f;
    char* ws = (char*)io->workspace;
    const int xdt = c.residual_dtype;   // storage of the residual stream AND of its gradient stream dx
    const bool rb = xdt == NBCI_BF16;
    const int gdt = rb ? dt : NBCI_F32;   // what the data-gradient GEMMs hand BatchNorm's backward (dtmp): bf16 with bf16 streams
    float* dx = (float*)(ws + w.dx);      // (bf16 elements when rb)
    float* dtmp = (float*)(ws + w.dtmp);
    const float scale = 1.0f / sqrtf((float)hd);
    float* rep = (float*)(ws + w.rep);
    const RepCfg rc{p.compact_total, NREP};
    auto RG = [&](int64_t flat_off) -> float* { return rep + p.compact_of(flat_off); };

    for (int seg = seg_hi; seg >= seg_lo; --seg) {
        if (seg == L + 1) {
            const int Mh = w.Mh;
            const bool ctc = c.method == NBCI_PTST_CTC;
            const void* dl = ctc ? ws + w.dlogits : ws + w.dpred;
            const int ldl = ctc ? w.vpad : w.ldp, nout = p.nout;
            const void* head_in = (!ctc && rb) ? ws + w.x_last : ws + w.pooled;   // (mlm with a bf16 stream: the head read the stream itself)
            const void* src = c.mlp_decoder ? ws + w.d1 : head_in;
            const int64_t ow = c.mlp_decoder ? p.d2w : p.d0w, ob = c.mlp_decoder ? p.d2b : p.d0b;
            float* dsrc = ctc ? (float*)(ws + w.dpool) : dx;   // mlm: the head's input gradient IS the stream gradient
            const int dsdt = ctc ? NBCI_F32 : xdt;             // (the pooled gradient stays f32: B*P rows)
            TRY(colsum_launch(dl, dt, ldl, Mh, nout, RG(ob), s, rc));
            TRY(wgrad(s, dt, nout, D, Mh, op(dl, es, 0, ldl, 0), op(src, es, 0, D, 0), grads + ow, D));
            if (c.mlp_decoder) {
                {   // d u = (dl W_2) * act'(u)   (act' stored by the forward in cA), decoder.projection.0 bias grad = column sums
                    nbci_gemm_desc d = gd(Mh, D, nout, dt, op(dl, es, 0, ldl, 1), op(W(p.d2w), es, 0, D, 0), ws + w.cA2, D, dt);
                    d.gate = ws + w.cA; d.ldg = D; d.gate_act = -1;
                    d.colsum = RG(p.d0b); d.colsum_rep_stride = rc.stride; d.colsum_nrep = rc.n;
                    TRY(gemm_launch_timed(d, s));
                }
                TRY(wgrad(s, dt, D, D, Mh, op(ws + w.cA2, es, 0, D, 0), op(head_in, es, 0, D, 0), grads + p.d0w, D));
                nbci_gemm_desc d = gd(Mh, D, D, dt, op(ws + w.cA2, es, 0, D, 1), op(W(p.d0w), es, 0, D, 0), dsrc, D, dsdt);
                TRY(gemm_launch_timed(d, s));
            } else {
                nbci_gemm_desc d = gd(Mh, D, nout, dt, op(dl, es, 0, ldl, 1), op(W(p.d0w), es, 0, D, 0), dsrc, D, dsdt);
                TRY(gemm_launch_timed(d, s));
            }
            if (ctc) TRY(ptst_pool_bwd_launch(dsrc, dx, B, C, P, D, s, xdt));
        } else if (seg >= 1) {
            const int l = seg - 1;
            const PtLayerWS& lw = w.L[l];
            const PtLayerOff& lo = p.L[l];
            WgradQueue wq; wq.dtype = dt; wq.s = s;
            // ---- x_out = x_mid + path_drop(ff.3(ff_drop(act(ff.0(BN3(x_mid))))))
            TRY(dropcast2d_launch(dx, ws + w.cA, dt, Mi, D, pp, io->seed, 19 + 4 * l, RG(lo.f3b), s, rc, xdt));
            TRY(wq.push(D, F, Mi, op(ws + w.cA, es, 0, D, 0), op(ws + lw.g, es, 0, F, 0), grads + lo.f3w, F));
            {
                nbci_gemm_desc d = gd(Mi, F, D, dt, op(ws + w.cA, es, 0, D, 1), op(W(lo.f3w), es, 0, F, 0), ws + w.dU, F, dt);
                d.gate = ws + lw.u; d.ldg = F; d.gate_act = -1; d.drop_p = pf; d.seed = io->seed; d.site = 18 + 4 * l;
                d.colsum = RG(lo.f0b); d.colsum_rep_stride = rc.stride; d.colsum_nrep = rc.n;
                TRY(gemm_launch_timed(d, s));
            }
            TRY(wq.push(F, D, Mi, op(ws + w.dU, es, 0, F, 0), op(ws + lw.y3, es, 0, D, 0), grads + lo.f0w, D));
            {
                nbci_gemm_desc d = gd(Mi, D, F, dt, op(ws + w.dU, es, 0, F, 1), op(W(lo.f0w), es, 0, D, 0), dtmp, D, gdt);
                TRY(gemm_launch_timed(d, s));
            }
            TRY(batchnorm_bwd_launch(dtmp, ws + lw.x_mid, (const float*)(ws + lw.mean3), (const float*)(ws + lw.rstd3),
                                     params + lo.n3w, dx, grads + lo.n3w, grads + lo.n3b, (float*)(ws + w.bnpart), (float*)(ws + w.bnsums), M, D,
                                     train, s, gdt, xdt));
            // ---- x_mid = x_in + path_drop(out_proj(MHA(BN1(x_in))))
            TRY(dropcast2d_launch(dx, ws + w.cA2, dt, Mi, D, pp, io->seed, 17 + 4 * l, RG(lo.ob), s, rc, xdt));
            TRY(wq.push(D, D, Mi, op(ws + w.cA2, es, 0, D, 0), op(ws + lw.ad, es, 0, D, 0), grads + lo.ow, D));
            {
                nbci_gemm_desc d = gd(Mi, D, D, dt, op(ws + w.cA2, es, 0, D, 1), op(W(lo.ow), es, 0, D, 0), ws + w.dAtt, D, dt);
                TRY(gemm_launch_timed(d, s));
            }
            const size_t pd = pa > 0.f ? lw.Pd : lw.P;
            const int64_t pz1 = (int64_t)nh * P * w.ldP, pz2 = (int64_t)P * w.ldP;
            const int64_t qz1 = (int64_t)P * 3 * D, az1 = (int64_t)P * D;
            const int nb = B * C * nh;
            if (w.small_attn) {
                if (w.flash)
                    TRY(fattn_bwd_launch(ws + lw.qkv, ws + lw.ad, ws + w.dAtt, (const float*)(ws + lw.lse), (float*)(ws + w.dsum), ws + w.dqkv,
                                         B * C, nh, P, D, pa, io->seed, 16 + 4 * l, s));
                else
                    TRY(sattn_bwd_launch(ws + lw.qkv, ws + lw.ad, ws + w.dAtt, (const float*)(ws + lw.lse), (float*)(ws + w.dsum), ws + w.dqkv, dt,
                                         B * C, nh, P, D, pa, io->seed, 16 + 4 * l, s));
            } else {
            {
                nbci_gemm_desc d = gd(P, P, hd, dt, op(ws + w.dAtt, es, 0, D, 1, 0, 0, az1, hd), op(ws + lw.qkv, es, 2 * D, 3 * D, 1, 0, 0, qz1, hd),
                                      ws + w.scores, w.ldS, NBCI_F32);
                d.batch = nb; d.zdiv = nh; d.czs1 = (int64_t)nh * P * w.ldS; d.czs2 = (int64_t)P * w.ldS;
                TRY(gemm_launch_timed(d, s));
            }
            {
                nbci_gemm_desc d = gd(P, hd, P, dt, op(ws + pd, es, 0, w.ldP, 0, 0, 0, pz1, pz2), op(ws + w.dAtt, es, 0, D, 0, 0, 0, az1, hd),
                                      (char*)(ws + w.dqkv) + (size_t)2 * D * es, 3 * D, dt);
                d.batch = nb; d.zdiv = nh; d.czs1 = qz1; d.czs2 = hd;
                TRY(gemm_launch_timed(d, s));
            }
            TRY(softmax_bwd_launch((const float*)(ws + w.scores), ws + lw.P, ws + w.dS, dt, B * C, nh, P, w.ldS, w.ldP, pa, io->seed, 16 + 4 * l, s));
            {
                nbci_gemm_desc d = gd(P, hd, P, dt, op(ws + w.dS, es, 0, w.ldP, 1, 0, 0, pz1, pz2), op(ws + lw.qkv, es, D, 3 * D, 0, 0, 0, qz1, hd),
                                      ws + w.dqkv, 3 * D, dt);
                d.batch = nb; d.zdiv = nh; d.czs1 = qz1; d.czs2 = hd; d.alpha = scale;
                TRY(gemm_launch_timed(d, s));
            }
            {
                nbci_gemm_desc d = gd(P, hd, P, dt, op(ws + w.dS, es, 0, w.ldP, 0, 0, 0, pz1, pz2), op(ws + lw.qkv, es, 0, 3 * D, 0, 0, 0, qz1, hd),
                                      (char*)(ws + w.dqkv) + (size_t)D * es, 3 * D, dt);
                d.batch = nb; d.zdiv = nh; d.czs1 = qz1; d.czs2 = hd; d.alpha = scale;
                TRY(gemm_launch_timed(d, s));
            }
            }
            TRY(colsum_launch(ws + w.dqkv, dt, 3 * D, Mi, 3 * D, RG(lo.qb), s, rc));
            TRY(wq.push(3 * D, D, Mi, op(ws + w.dqkv, es, 0, 3 * D, 0), op(ws + lw.y1, es, 0, D, 0), grads + lo.qw, D));
            TRY(wq.flush());
            {
                nbci_gemm_desc d = gd(Mi, D, 3 * D, dt, op(ws + w.dqkv, es, 0, 3 * D, 1), op(W(lo.qw), es, 0, D, 0), dtmp, D, gdt);
                TRY(gemm_launch_timed(d, s));
            }
            TRY(batchnorm_bwd_launch(dtmp, ws + lw.x_in, (const float*)(ws + lw.mean1), (const float*)(ws + lw.rstd1),
                                     params + lo.n1w, dx, grads + lo.n1w, grads + lo.n1b, (float*)(ws + w.bnpart), (float*)(ws + w.bnsums), M, D,
                                     train, s, gdt, xdt));
        } else {
            // ---- shared patch embedding (positions are fixed): K = patch_length, runs on the exact-f32 path
            const float* de = dx;
            if (ppos > 0.f || rb) {   // (a bf16 stream is widened on the way: the K = all-rows weight gradient below reads f32)
                TRY(dropcast2d_launch(dx, dtmp, NBCI_F32, Mi, D, ppos, io->seed, 4, RG(p.embb), s, rc, xdt));
                de = dtmp;
            } else {
                TRY(colsum_launch(dx, NBCI_F32, D, Mi, D, RG(p.embb), s, rc));
            }
            if (pl <= 16 && D % 4 == 0 && D / 4 <= 256 && 256 % (D / 4) == 0 && (256 / (D / 4)) * D * pl * 4 <= 65536) TRY(ptst_embed_wgrad_launch(de, (const float*)(ws + w.xm), grads + p.embw, Mi, pl, D, s));
            else TRY(wgrad(s, NBCI_F32, D, pl, Mi, op(de, 4, 0, D, 0), op(ws + w.xm, 4, 0, pl, 0), grads + p.embw, pl));
        }
    }
    // the replicated small-vector gradients of every segment of this call, folded in ONE launch (their compact ranges are adjacent)
    TRY(fold_replicas_launch(rep, rc.stride, rc.n, p.d_flat_of, p.cseg[seg_lo].first, p.cseg[seg_hi].second, grads, s));
    return NBCI_OK;
}

}  // namespace nbci

using namespace nbci;

extern "C" {

int nbci_ptst_plan_create(const nbci_ptst_config* cfg, nbci_ptst_plan* out) {
    if (!cfg || !out) return fail(NBCI_EINVAL, "ptst plan_create: null argument");
    const nbci_ptst_config& c = *cfg;
    NBCI_REQUIRE(c.d_model > 0 && c.num_attention_heads > 0 && c.d_model % c.num_attention_heads == 0, NBCI_ESHAPE,
                 "embed_dim must be divisible by num_heads");
    NBCI_REQUIRE(c.d_model % 8 == 0 && (c.d_model / c.num_attention_heads) % 8 == 0 && c.ffn_dim % 8 == 0, NBCI_ESHAPE,
                 "d_model, head size and ffn_dim must be multiples of 8");
    NBCI_REQUIRE(c.context_length > c.patch_length && c.patch_length > 0 && c.patch_stride > 0, NBCI_ESHAPE,
                 "Sequence length has to be greater than the patch length");
    NBCI_REQUIRE(c.patch_length <= 32, NBCI_ESHAPE, "patch_length must be <= 32");
    NBCI_REQUIRE(c.num_input_channels > 0 && c.num_hidden_layers >= 0, NBCI_ESHAPE, "bad PatchTST shape parameters");
    NBCI_REQUIRE(c.dtype == NBCI_F32 || c.dtype == NBCI_BF16, NBCI_EINVAL, "dtype must be f32 or bf16");
    NBCI_REQUIRE(c.residual_dtype == NBCI_F32 || (c.residual_dtype == NBCI_BF16 && c.dtype == NBCI_BF16), NBCI_EINVAL,
                 "residual_dtype must be f32, or bf16 together with dtype bf16");
    NBCI_REQUIRE(c.method == NBCI_PTST_CTC || c.method == NBCI_PTST_MLM, NBCI_EINVAL, "Method not implemented yet for PatchTST");
    NBCI_REQUIRE(c.method != NBCI_PTST_CTC || (c.vocab > 0 && c.blank_id >= 0 && c.blank_id < c.vocab), NBCI_EINVAL, "bad vocab / blank_id");
    NBCI_REQUIRE(c.method != NBCI_PTST_MLM || c.do_mask_input, NBCI_EINVAL, "Can't pretrain with inactive masking");
    NBCI_REQUIRE(!c.fp8_qkv || (c.dtype == NBCI_BF16 && c.d_model % 128 == 0 && c.d_model <= 512), NBCI_ESHAPE,
                 "patchtst: fp8 q/k/v needs the bf16 path and d_model = 128, 256, 384 or 512 (the fp8 GEMM keeps a row block's K in registers)");
    PtPlan* p = new PtPlan();
    p->c = c;
    p->P = (std::max(c.context_length, c.patch_length) - c.patch_length) / c.patch_stride + 1;
    p->start = c.context_length - (c.patch_length + c.patch_stride * (p->P - 1));
    pt_layout(*p);
    p->d_flat_of = nullptr;
    hipError_t e = hipMalloc(&p->d_flat_of, std::max<size_t>(4, p->flat_of.size() * sizeof(int)));
    if (e == hipSuccess && !p->flat_of.empty())
        e = hipMemcpy(p->d_flat_of, p->flat_of.data(), p->flat_of.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete p; return fail(NBCI_EHIP, std::string("ptst plan_create: ") + hipGetErrorString(e)); }
    *out = (nbci_ptst_plan)p;
    return NBCI_OK;
}

void nbci_ptst_plan_destroy(nbci_ptst_plan plan) {
    PtPlan* p = (PtPlan*)plan;
    if (!p) return;
    if (p->d_flat_of) (void)hipFree(p->d_flat_of);
    delete p;
}

int64_t nbci_ptst_param_count(nbci_ptst_plan plan) { return plan ? ((PtPlan*)plan)->total : -1; }
int32_t nbci_ptst_num_params(nbci_ptst_plan plan) { return plan ? (int32_t)((PtPlan*)plan)->params.size() : -1; }
int32_t nbci_ptst_num_segments(nbci_ptst_plan plan) { return plan ? (int32_t)((PtPlan*)plan)->seg.size() : -1; }
int32_t nbci_ptst_num_patches(nbci_ptst_plan plan) { return plan ? ((PtPlan*)plan)->P : -1; }
int64_t nbci_ptst_aux_floats(nbci_ptst_plan plan) {
    if (!plan) return -1;
    const PtPlan* p = (PtPlan*)plan;
    return (int64_t)p->P * p->c.d_model + (int64_t)p->c.num_hidden_layers * 4 * p->c.d_model;
}

int nbci_ptst_param_info(nbci_ptst_plan plan, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel, int32_t* rows,
                         int32_t* cols, int32_t* segment) {
    PtPlan* p = (PtPlan*)plan;
    if (!p || index < 0 || index >= (int)p->params.size()) return fail(NBCI_EINVAL, "ptst param_info: bad plan/index");
    const PInfo& i = p->params[index];
    if (name && name_cap > 0) { strncpy(name, i.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = i.off;
    if (numel) *numel = i.numel;
    if (rows) *rows = i.rows;
    if (cols) *cols = i.cols;
    if (segment) *segment = i.seg;
    return NBCI_OK;
}

int nbci_ptst_segment_range(nbci_ptst_plan plan, int32_t seg, int64_t* begin, int64_t* end) {
    PtPlan* p = (PtPlan*)plan;
    if (!p || seg < 0 || seg >= (int)p->seg.size()) return fail(NBCI_EINVAL, "ptst segment_range: bad plan/segment");
    *begin = p->seg[seg].first; *end = p->seg[seg].second;
    return NBCI_OK;
}

int64_t nbci_ptst_workspace_bytes(nbci_ptst_plan plan, int32_t B, int32_t S) {
    PtPlan* p = (PtPlan*)plan;
    if (!p) { fail(NBCI_EINVAL, "ptst workspace_bytes: null plan"); return -1; }
    PtWS w;
    if (pt_carve(*p, B, S, w) != NBCI_OK) return -1;
    return (int64_t)w.bytes;
}

int nbci_ptst_forward(nbci_ptst_plan plan, const float* params, const void* params_lp, const nbci_ptst_io* io, nbci_stream_t stream) {
    if (!plan) return fail(NBCI_EINVAL, "ptst forward: null plan");
    return ptst_forward(*(PtPlan*)plan, params, params_lp, io, (hipStream_t)stream);
}

int nbci_ptst_backward(nbci_ptst_plan plan, const float* params, const void* params_lp, const nbci_ptst_io* io, float* grads, int32_t seg_hi,
                       int32_t seg_lo, nbci_stream_t stream) {
    if (!plan) return fail(NBCI_EINVAL, "ptst backward: null plan");
    return ptst_backward(*(PtPlan*)plan, params, params_lp, io, grads, seg_hi, seg_lo, (hipStream_t)stream);
}

}  // extern "C"
