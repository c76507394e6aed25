// itransformer.hip — host-side orchestration of the iTransformer SSL (mlm) forward/backward on one stream.
//
// Replaces iTransformerEncoder.forward in `mlp` embedder mode + iTransformer.forward, method "mlm"
// (models/itransformer.py:175-210, 312-359) and their autograd graph: channel-as-token embedding MLP,
// LayerNorm'd channel/region embeddings, CLS token, post-norm torch.nn.TransformerEncoderLayer stack
// (itransformer.py:158-173), final norm, mlm decoder, masked Poisson-NLL / MSE. The maskers run before it
// (nbci_masker). Same conventions as ndt1.hip: ONE flat f32 parameter buffer (+ bf16 shadow), caller-owned
// workspace, gradient segments [embedding | layer 0..L-1 | final norm + decoder] = RCCL buckets.
//
// Post-norm data flow per layer (x = layer input, f32 copy in a ping-pong scratch + operand-dtype copy saved):
//   qkv = x W_in^T + b            r1 = x + drop1(attn(qkv) W_o^T + b_o)      x1 = LN1(r1)
//   g = drop(relu(x1 W_1^T + b1)) r2 = x1 + drop2(g W_2^T + b2)              x' = LN2(r2)
// Saved for backward: operand-dtype x, qkv, P/Pd, attn out, x1, g; f32 r1, r2 with their LN statistics.
#include <cmath>
#include <cstdlib>

#include "plan_common.h"

namespace nbci {

struct ItrLayerOff {
    int64_t inw, inb, ow, ob, w1, b1, w2, b2, n1w, n1b, n2w, n2b;
};
// one torch.nn.TransformerEncoder: its layers' parameter offsets + the final norm's
struct StackOff {
    std::vector<ItrLayerOff> L;
    int64_t fnw, fnb;
};

struct ItrPlan {
    nbci_itr_config c;
    std::vector<PInfo> params;
    StackOff enc, emb;   // the channel-token encoder (itransformer.py:158-173); the UnivariateTransformer's (:58-73, emb_mode 1)
    int64_t e0w, e0b, e3w, e3b, enw, enb, chw, chnw, chnb, rgw, rgnw, rgnb, cls, d0w, d0b, d2w, d2b;
    int64_t us0w, us0b, us2w, us2b, upos, ucls, upw, upb;        // embed.embed_spikes.{0,2}, embed.embed_pos, embed.cls_embed, embed_proj.0 (emb_mode 1)
    int64_t dp0w, dp0b, dp2w, dp2b, dpnw, dpnb;                   // depth_embeddings.{0,2,3}
    int64_t total;
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

static int64_t itr_add(ItrPlan& p, int64_t& cur, const std::string& name, int rows, int cols, int seg) {
    cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
    const int64_t off = cur;
    const int64_t n = (int64_t)rows * (cols > 0 ? cols : 1);
    p.params.push_back({name, off, n, rows, cols, seg});
    cur += n;
    return off;
}

// the parameters of one TransformerEncoderLayer in state-dict order
static ItrLayerOff itr_add_layer(ItrPlan& p, int64_t& cur, const std::string& pre, int H, int seg) {
    const int F = 4 * H;
    ItrLayerOff o;
    o.inw = itr_add(p, cur, pre + "self_attn.in_proj_weight", 3 * H, H, seg);
    o.inb = itr_add(p, cur, pre + "self_attn.in_proj_bias", 3 * H, 0, seg);
    o.ow = itr_add(p, cur, pre + "self_attn.out_proj.weight", H, H, seg);
    o.ob = itr_add(p, cur, pre + "self_attn.out_proj.bias", H, 0, seg);
    o.w1 = itr_add(p, cur, pre + "linear1.weight", F, H, seg);
    o.b1 = itr_add(p, cur, pre + "linear1.bias", F, 0, seg);
    o.w2 = itr_add(p, cur, pre + "linear2.weight", H, F, seg);
    o.b2 = itr_add(p, cur, pre + "linear2.bias", H, 0, seg);
    o.n1w = itr_add(p, cur, pre + "norm1.weight", H, 0, seg);
    o.n1b = itr_add(p, cur, pre + "norm1.bias", H, 0, seg);
    o.n2w = itr_add(p, cur, pre + "norm2.weight", H, 0, seg);
    o.n2b = itr_add(p, cur, pre + "norm2.bias", H, 0, seg);
    return o;
}

static void itr_layout(ItrPlan& p) {
    const auto& c = p.c;
    const int H = c.hidden, T = c.max_n_bins;
    int64_t cur = 0;
    p.e0w = p.e0b = p.e3w = p.e3b = -1;
    p.us0w = p.us0b = p.us2w = p.us2b = p.upos = p.ucls = p.upw = p.upb = -1;
    if (c.emb_mode == 0) {
        p.e0w = itr_add(p, cur, "encoder.embed.0.0.weight", H, T, 0);
        p.e0b = itr_add(p, cur, "encoder.embed.0.0.bias", H, 0, 0);
        p.e3w = itr_add(p, cur, "encoder.embed.0.3.weight", H, H, 0);
        p.e3b = itr_add(p, cur, "encoder.embed.0.3.bias", H, 0, 0);
        p.enw = itr_add(p, cur, "encoder.embed.1.weight", H, 0, 0);
        p.enb = itr_add(p, cur, "encoder.embed.1.bias", H, 0, 0);
    } else {   // UnivariateTransformer + embed_proj (state-dict order of itransformer.py:48-73,119-124)
        const int h = c.emb_hidden;
        p.us0w = itr_add(p, cur, "encoder.embed.embed_spikes.0.weight", h, 0, 0);   // (h,1): a column, summed like a bias
        p.us0b = itr_add(p, cur, "encoder.embed.embed_spikes.0.bias", h, 0, 0);
        p.us2w = itr_add(p, cur, "encoder.embed.embed_spikes.2.weight", h, h, 0);
        p.us2b = itr_add(p, cur, "encoder.embed.embed_spikes.2.bias", h, 0, 0);
        p.upos = itr_add(p, cur, "encoder.embed.embed_pos.weight", T, h, 0);
        p.ucls = itr_add(p, cur, "encoder.embed.cls_embed.weight", h, 0, 0);        // (1,h)
        for (int l = 0; l < c.emb_layers; ++l)
            p.emb.L.push_back(itr_add_layer(p, cur, "encoder.embed.transformer.layers." + std::to_string(l) + ".", h, 0));
        p.emb.fnw = itr_add(p, cur, "encoder.embed.transformer.norm.weight", h, 0, 0);
        p.emb.fnb = itr_add(p, cur, "encoder.embed.transformer.norm.bias", h, 0, 0);
        p.upw = itr_add(p, cur, "encoder.embed_proj.0.weight", H, h, 0);
        p.upb = itr_add(p, cur, "encoder.embed_proj.0.bias", H, 0, 0);
        p.enw = itr_add(p, cur, "encoder.embed_proj.1.weight", H, 0, 0);
        p.enb = itr_add(p, cur, "encoder.embed_proj.1.bias", H, 0, 0);
    }
    p.chw = p.chnw = p.chnb = p.rgw = p.rgnw = p.rgnb = p.cls = -1;
    p.dp0w = p.dp0b = p.dp2w = p.dp2b = p.dpnw = p.dpnb = -1;
    if (c.max_n_channels > 0) {
        p.chw = itr_add(p, cur, "encoder.channel_embeddings.0.weight", c.max_n_channels, H, 0);
        p.chnw = itr_add(p, cur, "encoder.channel_embeddings.1.weight", H, 0, 0);
        p.chnb = itr_add(p, cur, "encoder.channel_embeddings.1.bias", H, 0, 0);
    }
    if (c.n_regions > 0) {
        p.rgw = itr_add(p, cur, "encoder.region_embeddings.0.weight", c.n_regions, H, 0);
        p.rgnw = itr_add(p, cur, "encoder.region_embeddings.1.weight", H, 0, 0);
        p.rgnb = itr_add(p, cur, "encoder.region_embeddings.1.bias", H, 0, 0);
    }
    if (c.embed_depth) {
        p.dp0w = itr_add(p, cur, "encoder.depth_embeddings.0.weight", H, 0, 0);     // (H,1)
        p.dp0b = itr_add(p, cur, "encoder.depth_embeddings.0.bias", H, 0, 0);
        p.dp2w = itr_add(p, cur, "encoder.depth_embeddings.2.weight", H, H, 0);
        p.dp2b = itr_add(p, cur, "encoder.depth_embeddings.2.bias", H, 0, 0);
        p.dpnw = itr_add(p, cur, "encoder.depth_embeddings.3.weight", H, 0, 0);
        p.dpnb = itr_add(p, cur, "encoder.depth_embeddings.3.bias", H, 0, 0);
    }
    if (c.use_cls) p.cls = itr_add(p, cur, "encoder.cls_embed.weight", H, 0, 0);   // (1,H): summed like a bias
    cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
    p.seg.push_back({0, cur});
    for (int l = 0; l < c.n_layers; ++l) {
        const int64_t begin = cur;
        p.enc.L.push_back(itr_add_layer(p, cur, "encoder.transformer.layers." + std::to_string(l) + ".", H, l + 1));
        cur = (cur + PALIGN - 1) / PALIGN * PALIGN;
        p.seg.push_back({begin, cur});
    }
    const int64_t begin = cur;
    const int hs = c.n_layers + 1;
    p.enc.fnw = itr_add(p, cur, "encoder.transformer.norm.weight", H, 0, hs);
    p.enc.fnb = itr_add(p, cur, "encoder.transformer.norm.bias", H, 0, hs);
    p.d2w = p.d2b = -1;
    if (c.mlp_decoder) {
        p.d0w = itr_add(p, cur, "decoder.0.weight", H, H, hs);
        p.d0b = itr_add(p, cur, "decoder.0.bias", H, 0, hs);
        p.d2w = itr_add(p, cur, "decoder.2.weight", T, H, hs);
        p.d2b = itr_add(p, cur, "decoder.2.bias", T, 0, hs);
    } else {
        p.d0w = itr_add(p, cur, "decoder.0.weight", T, H, hs);
        p.d0b = itr_add(p, cur, "decoder.0.bias", T, 0, hs);
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

// ---- workspace -------------------------------------------------------------------------------
struct ItrLayerWS {
    size_t xb, qkv, P, Pd, ad, lse, r1, mean1, rstd1, x1b, g, r2, mean2, rstd2;
};
// what one encoder stack keeps: per-layer saved activations, the f32 ping-pong copies of the LayerNorm outputs (f32 streams only), the
// final norm's input / statistics / output, attention scratch, and the backward's streams and operand copies
struct StackWS {
    std::vector<ItrLayerWS> L;
    size_t yA, yB, xlast_b, mean_o, rstd_o, xo, scores, dsum;
    size_t dY, dR, cA, cA2, dU, dAtt, dqkv, dS;
    int Bq, S, M, H, nh, ldS, ldP;
    bool small_attn, flash;
};
struct ItrWS {
    size_t xs, h0, t2, mean_e, rstd_e, chtab, mean_c, rstd_c, dchtab, rgtab, mean_r, rstd_r, drgtab, ssidx;
    size_t dep_h, dep_t, dep_o, mean_d, rstd_d, dep_c, dep_du;    // depth embedding: relu(Linear(1->H)), Linear(H->H) f32, its LayerNorm, stats, backward operand copy, d(pre-activation)
    size_t uni_h, uni_du;                                          // UnivariateTransformer: relu(Linear(1->h)) rows, d(pre-activation)
    StackWS enc, emb;
    size_t d1, pred, dpred, dtok, dH0, rep;
    size_t bytes;
    int S, M, M0, ldT;
};

static int stack_carve(StackWS& k, size_t& cur, int dtype, int rdtype, int Bq, int S, int H, int nh, int nl) {
    const size_t es = dtype == NBCI_BF16 ? 2 : 4;
    const size_t rs = rdtype == NBCI_BF16 ? 2 : 4;   // the LayerNorm inputs r1 / r2 and the gradient streams dY / dR
    const size_t M = (size_t)Bq * S, F = 4 * (size_t)H;
    NBCI_REQUIRE(M * F < (1ull << 32), NBCI_ESHAPE, "itransformer: batch too large for the 32-bit dropout counter");
    NBCI_REQUIRE((size_t)Bq * nh * S * (size_t)S < (1ull << 32), NBCI_ESHAPE, "itransformer: attention too large for the 32-bit dropout counter");
    k.Bq = Bq; k.S = S; k.M = (int)M; k.H = H; k.nh = nh;
    k.ldS = (S + 3) / 4 * 4;
    k.ldP = (S + 7) / 8 * 8;
    k.L.resize(nl);
    k.flash = fattn_eligible(dtype, S, H, nh);
    k.small_attn = k.flash || sattn_eligible(dtype, S, H, nh);
    const size_t nP = k.small_attn ? 0 : (size_t)Bq * nh * S * k.ldP;
    const size_t nstat = sattn_stat_floats(Bq, nh, S);
    for (auto& l : k.L) {
        l.xb = bump(cur, M * H * es);
        l.qkv = bump(cur, M * 3 * H * es);
        l.P = bump(cur, nP * es);
        l.Pd = bump(cur, nP * es);
        l.ad = bump(cur, M * H * es);
        l.lse = bump(cur, nstat * 4);
        l.r1 = bump(cur, M * H * rs);
        l.mean1 = bump(cur, M * 4); l.rstd1 = bump(cur, M * 4);
        l.x1b = bump(cur, M * H * es);
        l.g = bump(cur, M * F * es);
        l.r2 = bump(cur, M * H * rs);
        l.mean2 = bump(cur, M * 4); l.rstd2 = bump(cur, M * 4);
    }
    k.yA = bump(cur, M * H * 4);
    k.yB = bump(cur, M * H * 4);
    k.xlast_b = bump(cur, M * H * es);
    k.mean_o = bump(cur, M * 4); k.rstd_o = bump(cur, M * 4);
    k.xo = bump(cur, M * H * es);
    k.scores = bump(cur, k.small_attn ? 0 : (size_t)Bq * nh * S * k.ldS * 4);
    k.dsum = bump(cur, nstat * 4);
    k.dY = bump(cur, M * H * rs);
    k.dR = bump(cur, M * H * 4);   // (f32-sized whatever the stream type: also an f32 scratch of the embedding side)
    k.cA = bump(cur, M * H * es);
    k.cA2 = bump(cur, M * H * es);
    k.dU = bump(cur, M * F * es);
    k.dAtt = bump(cur, M * H * es);
    k.dqkv = bump(cur, M * 3 * H * es);
    k.dS = bump(cur, nP * es);
    return NBCI_OK;
}

static int itr_carve(const ItrPlan& p, int B, int N, ItrWS& w) {
    const auto& c = p.c;
    NBCI_REQUIRE(B > 0 && N > 0, NBCI_ESHAPE, "itransformer: B and N must be positive");
    const int S = N + (c.use_cls ? 1 : 0);
    NBCI_REQUIRE(S <= 2048, NBCI_ESHAPE, "itransformer: at most 2048 tokens per sample");
    const size_t es = c.dtype == NBCI_BF16 ? 2 : 4;
    const size_t H = c.hidden, T = c.max_n_bins;
    const size_t M = (size_t)B * S, M0 = (size_t)B * N;
    w.S = S; w.M = (int)M; w.M0 = (int)M0;
    w.ldT = ((int)T + 7) / 8 * 8;
    size_t cur = 0;
    w.xs = bump(cur, M0 * T * 4);
    w.h0 = bump(cur, c.emb_mode == 0 ? M0 * H * es : 0);
    w.t2 = bump(cur, M0 * H * 4);
    w.mean_e = bump(cur, M0 * 4); w.rstd_e = bump(cur, M0 * 4);
    const size_t C = c.max_n_channels, R = c.n_regions;
    w.chtab = bump(cur, C * H * 4); w.mean_c = bump(cur, C * 4 + 4); w.rstd_c = bump(cur, C * 4 + 4); w.dchtab = bump(cur, C * H * 4);
    w.rgtab = bump(cur, R * H * 4); w.mean_r = bump(cur, R * 4 + 4); w.rstd_r = bump(cur, R * 4 + 4); w.drgtab = bump(cur, R * H * 4);
    w.ssidx = bump(cur, M0 * 8);
    const size_t D = c.embed_depth ? M0 : 0;
    w.dep_h = bump(cur, D * H * es); w.dep_t = bump(cur, D * H * 4); w.dep_o = bump(cur, D * H * 4);
    w.mean_d = bump(cur, D * 4 + 4); w.rstd_d = bump(cur, D * 4 + 4); w.dep_c = bump(cur, D * H * es); w.dep_du = bump(cur, D * H * 4);
    if (c.emb_mode == 1) {
        NBCI_REQUIRE(M0 * (T + 1) < (1ull << 31), NBCI_ESHAPE, "itransformer: too many embedder token rows (B x N x (1 + max_n_bins))");
        const size_t Me = M0 * (T + 1), h = c.emb_hidden;
        w.uni_h = bump(cur, Me * h * es);
        w.uni_du = bump(cur, Me * h * 4);
        TRY(stack_carve(w.emb, cur, c.dtype, c.residual_dtype, (int)M0, (int)T + 1, c.emb_hidden, c.emb_heads, c.emb_layers));
    } else {
        w.uni_h = w.uni_du = cur;
    }
    TRY(stack_carve(w.enc, cur, c.dtype, c.residual_dtype, B, S, c.hidden, c.n_heads, c.n_layers));
    w.d1 = bump(cur, M * H * es);
    w.pred = bump(cur, M * w.ldT * 4);
    w.dpred = bump(cur, M * w.ldT * es);
    w.dtok = bump(cur, M0 * H * 4);
    w.dH0 = bump(cur, M0 * H * 4);
    w.rep = bump(cur, (size_t)NREP * p.compact_total * 4);
    w.bytes = (cur + 255) / 256 * 256;
    return NBCI_OK;
}

static int itr_validate(const ItrPlan& p, const nbci_itr_io* io) {
    NBCI_REQUIRE(io, NBCI_EINVAL, "itransformer: null io");
    NBCI_REQUIRE(io->spikes && io->masked && io->mask && io->spikes_mask, NBCI_EINVAL,
                 "itransformer: spikes, masked, mask and spikes_mask are required");
    NBCI_REQUIRE(io->workspace, NBCI_EWORKSPACE, "itransformer: null workspace");
    NBCI_REQUIRE(((uintptr_t)io->workspace) % 256 == 0, NBCI_EALIGN, "itransformer: workspace must be 256-byte aligned");
    NBCI_REQUIRE(!(p.c.n_regions > 0 && !io->region_idx), NBCI_EINVAL, "itransformer: region_idx required when embed_region is on");
    NBCI_REQUIRE(!(p.c.embed_depth && !io->neuron_depths), NBCI_EINVAL, "itransformer: neuron_depths required when embed_depth is on");
    NBCI_REQUIRE(io->N <= p.c.max_n_channels || p.c.max_n_channels == 0 || io->spikes_spacestamp, NBCI_ESHAPE,
                 "itransformer: more channels than max_n_channels");
    return NBCI_OK;
}

__global__ void itr_iota_kernel(long long* out, int B, int N) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * N) out[i] = i % N;
}

struct ItrCtx {
    const ItrPlan& p;
    const float* pf;
    const void* pw;
    size_t es;
    char* ws;
    ItrWS w;
    hipStream_t s;
    const void* W(int64_t off) const { return (const char*)pw + off * (int64_t)es; }
};

// where a stack expects its input: the operand-dtype copy, and (f32 streams) the f32 copy the first residual add reads
static inline size_t stack_in_b(const StackWS& k) { return k.L.empty() ? k.xlast_b : k.L[0].xb; }

// One torch.nn.TransformerEncoder forward (post-norm layers + final norm; itransformer.py:58-73 / :158-173) over k.Bq sequences of k.S
// tokens. Input: stack_in_b(k) (operand dtype) and, with f32 streams, k.yA (f32). Output: k.xo (operand dtype, after the final norm).
// Dropout p = pl at sites site0 + 4 l + {0 attention probabilities, 1 dropout1, 2 FFN inner, 3 dropout2}.
static int stack_forward(const ItrCtx& x, const StackOff& so, const StackWS& k, int act, float pl, uint32_t seed, uint32_t site0) {
    const auto& c = x.p.c;
    const float* params = x.pf;
    char* ws = x.ws;
    hipStream_t s = x.s;
    const int dt = c.dtype, xdt = c.residual_dtype;
    const bool rb = xdt == NBCI_BF16;
    const size_t es = x.es;
    const int Bq = k.Bq, S = k.S, M = k.M, H = k.H, F = 4 * H, nh = k.nh, hd = H / nh;
    const float scale = 1.0f / sqrtf((float)hd);
    float* yA = (float*)(ws + k.yA);
    float* yB = (float*)(ws + k.yB);
    const int nl = (int)k.L.size();
    // residual_dtype bf16: r1 / r2 are stored in bf16 and the residual a sub-layer adds is the bf16 LayerNorm output its GEMMs read
    // (lw.xb / lw.x1b) - no f32 copies yA / yB of the LayerNorm outputs; sums in f32, one rounding per store
    for (int l = 0; l < nl; ++l) {
        const ItrLayerWS& lw = k.L[l];
        const ItrLayerOff& lo = so.L[l];
        const uint32_t sl = site0 + 4 * l;
        {
            nbci_gemm_desc d = gd(M, 3 * H, H, dt, op(ws + lw.xb, es, 0, H, 1), op(x.W(lo.inw), es, 0, H, 1), ws + lw.qkv, 3 * H, dt);
            d.bias = params + lo.inb;
            TRY(gemm_launch_timed(d, s));
        }
        if (k.small_attn) {
            if (k.flash) TRY(fattn_fwd_launch(ws + lw.qkv, ws + lw.ad, (float*)(ws + lw.lse), Bq, nh, S, H, pl, seed, sl, s));
            else TRY(sattn_fwd_launch(ws + lw.qkv, ws + lw.ad, (float*)(ws + lw.lse), dt, Bq, nh, S, H, pl, seed, sl, s));
        } else {
        {   // scores = q k^T / sqrt(hd), batched over (b, head); no mask (itransformer.py:209)
            nbci_gemm_desc d = gd(S, S, hd, dt, op(ws + lw.qkv, es, 0, 3 * H, 1, 0, 0, (int64_t)S * 3 * H, hd),
                                  op(ws + lw.qkv, es, H, 3 * H, 1, 0, 0, (int64_t)S * 3 * H, hd), ws + k.scores, k.ldS, NBCI_F32);
            d.batch = Bq * nh; d.zdiv = nh; d.czs1 = (int64_t)nh * S * k.ldS; d.czs2 = (int64_t)S * k.ldS; d.alpha = scale;
            TRY(gemm_launch_timed(d, s));
        }
        TRY(softmax_fwd_launch((const float*)(ws + k.scores), ws + lw.P, ws + (pl > 0.f ? lw.Pd : lw.P), dt, nullptr, Bq, nh, S, k.ldS,
                               k.ldP, -2, -2, pl, seed, sl, s));
        {
            const size_t pd = pl > 0.f ? lw.Pd : lw.P;
            nbci_gemm_desc d = gd(S, hd, S, dt, op(ws + pd, es, 0, k.ldP, 1, 0, 0, (int64_t)nh * S * k.ldP, (int64_t)S * k.ldP),
                                  op(ws + lw.qkv, es, 2 * H, 3 * H, 0, 0, 0, (int64_t)S * 3 * H, hd), ws + lw.ad, H, dt);
            d.batch = Bq * nh; d.zdiv = nh; d.czs1 = (int64_t)S * H; d.czs2 = hd;
            TRY(gemm_launch_timed(d, s));
        }
        }
        {   // r1 = x + dropout1(out_proj(a))
            nbci_gemm_desc d = gd(M, H, H, dt, op(ws + lw.ad, es, 0, H, 1), op(x.W(lo.ow), es, 0, H, 1), ws + lw.r1, H, xdt);
            d.bias = params + lo.ob; d.drop_p = pl; d.seed = seed; d.site = sl + 1; d.ldr = H;
            d.residual = rb ? (const void*)(ws + lw.xb) : (const void*)yA; d.residual_dtype = xdt;
            TRY(gemm_launch_timed(d, s));
        }
        TRY(layernorm_fwd_launch(ws + lw.r1, xdt, params + lo.n1w, params + lo.n1b, ws + lw.x1b, dt, (float*)(ws + lw.mean1),
                                 (float*)(ws + lw.rstd1), M, H, s, rb ? nullptr : yB));
        {   // g = dropout(act(linear1(x1)))
            nbci_gemm_desc d = gd(M, F, H, dt, op(ws + lw.x1b, es, 0, H, 1), op(x.W(lo.w1), es, 0, H, 1), ws + lw.g, F, dt);
            d.bias = params + lo.b1; d.act = act; d.drop_p = pl; d.seed = seed; d.site = sl + 2;
            TRY(gemm_launch_timed(d, s));
        }
        {   // r2 = x1 + dropout2(linear2(g))
            nbci_gemm_desc d = gd(M, H, F, dt, op(ws + lw.g, es, 0, F, 1), op(x.W(lo.w2), es, 0, F, 1), ws + lw.r2, H, xdt);
            d.bias = params + lo.b2; d.drop_p = pl; d.seed = seed; d.site = sl + 3; d.ldr = H;
            d.residual = rb ? (const void*)(ws + lw.x1b) : (const void*)yB; d.residual_dtype = xdt;
            TRY(gemm_launch_timed(d, s));
        }
        void* xb_next = ws + (l + 1 < nl ? k.L[l + 1].xb : k.xlast_b);
        TRY(layernorm_fwd_launch(ws + lw.r2, xdt, params + lo.n2w, params + lo.n2b, xb_next, dt, (float*)(ws + lw.mean2),
                                 (float*)(ws + lw.rstd2), M, H, s, rb ? nullptr : yA));
    }
    // final norm (TransformerEncoder(norm=...)); its input (the last layer's output, or the stack's input) stays where it is until the
    // next forward: yA in f32, or the bf16 operand copy
    const void* fn_in = rb ? (const void*)(ws + k.xlast_b) : (const void*)yA;
    TRY(layernorm_fwd_launch(fn_in, xdt, params + so.fnw, params + so.fnb, ws + k.xo, dt, (float*)(ws + k.mean_o), (float*)(ws + k.rstd_o), M, H, s));
    return NBCI_OK;
}

// what the backward pieces of a stack share
struct StackBwd {
    const ItrCtx& x;
    const StackOff& so;
    const StackWS& k;
    float* grads;
    float* rep;
    RepCfg rc;
    int act;
    float pl;
    uint32_t seed, site0;
    float* RG(int64_t flat_off) const { return rep + x.p.compact_of(flat_off); }
    LnCast cast_to(size_t buf, float pp, uint32_t site, int64_t bias_off) const {
        return LnCast{x.ws + buf, x.p.c.dtype == NBCI_BF16, drop_threshold(pp), pp > 0.f ? 1.f / (1.f - pp) : 1.f, drop_key(seed, site), RG(bias_off)};
    }
};
static const LnCast NO_CAST{nullptr, 0, 0u, 1.f, 0u, nullptr};

// final norm backward: d(output) in k.dR (stream dtype) -> k.dY = d(last layer's output)
static int stack_backward_final(const StackBwd& b) {
    const StackWS& k = b.k;
    char* ws = b.x.ws;
    const bool rb = b.x.p.c.residual_dtype == NBCI_BF16;
    return layernorm_bwd_launch(ws + k.dR, rb, rb ? (const void*)(ws + k.xlast_b) : (const void*)(ws + k.yA), b.x.pf + b.so.fnw,
                                (const float*)(ws + k.mean_o), (const float*)(ws + k.rstd_o), LnStreams{rb, nullptr, ws + k.dY, rb},
                                b.RG(b.so.fnw), b.RG(b.so.fnb), k.M, k.H, b.x.s, b.rc, NO_CAST);
}

// one layer's backward: k.dY = d(layer output) -> k.dY = d(layer input); parameter gradients into grads / the replicas
static int stack_backward_layer(const StackBwd& b, int l) {
    const ItrCtx& x = b.x;
    const auto& c = x.p.c;
    const StackWS& k = b.k;
    const float* params = x.pf;
    char* ws = x.ws;
    hipStream_t s = x.s;
    float* grads = b.grads;
    const int dt = c.dtype, xdt = c.residual_dtype;
    const bool rb = xdt == NBCI_BF16;
    const size_t es = x.es;
    const int Bq = k.Bq, S = k.S, M = k.M, H = k.H, F = 4 * H, nh = k.nh, hd = H / nh;
    const float scale = 1.0f / sqrtf((float)hd);
    const float pl = b.pl;
    const uint32_t sl = b.site0 + 4 * l;
    float* dY = (float*)(ws + k.dY);   // (bf16 elements with residual_dtype bf16, as dR, r1, r2)
    float* dR = (float*)(ws + k.dR);
    const RepCfg rc = b.rc;
    const ItrLayerWS& lw = k.L[l];
    const ItrLayerOff& lo = b.so.L[l];
    WgradQueue wq; wq.dtype = dt; wq.s = s;
    // ---- x' = LN2(r2), r2 = x1 + dropout2(linear2(g)), g = dropout(act(linear1(x1)))
    TRY(layernorm_bwd_launch(dY, rb, ws + lw.r2, params + lo.n2w, (const float*)(ws + lw.mean2), (const float*)(ws + lw.rstd2),
                             LnStreams{rb, nullptr, dR, rb}, b.RG(lo.n2w), b.RG(lo.n2b), M, H, s, rc, b.cast_to(k.cA, pl, sl + 3, lo.b2)));
    TRY(wq.push(H, F, M, op(ws + k.cA, es, 0, H, 0), op(ws + lw.g, es, 0, F, 0), grads + lo.w2, F));
    {   // du = (c W_2) * act'(u) * keep: for ReLU both factors are read off g itself (g > 0 <=> u > 0 and kept)
        nbci_gemm_desc d = gd(M, F, H, dt, op(ws + k.cA, es, 0, H, 1), op(x.W(lo.w2), es, 0, F, 0), ws + k.dU, F, dt);
        d.gate = ws + lw.g; d.ldg = F; d.gate_act = b.act;
        d.drop_p = pl; d.seed = b.seed; d.site = sl + 2;
        d.colsum = b.RG(lo.b1); d.colsum_rep_stride = rc.stride; d.colsum_nrep = rc.n;
        TRY(gemm_launch_timed(d, s));
    }
    TRY(wq.push(F, H, M, op(ws + k.dU, es, 0, F, 0), op(ws + lw.x1b, es, 0, H, 0), grads + lo.w1, H));
    {   // d x1 = du W_1 + d r2
        nbci_gemm_desc d = gd(M, H, F, dt, op(ws + k.dU, es, 0, F, 1), op(x.W(lo.w1), es, 0, H, 0), dY, H, xdt);
        d.residual = dR; d.ldr = H; d.residual_dtype = xdt;
        TRY(gemm_launch_timed(d, s));
    }
    // ---- x1 = LN1(r1), r1 = x + dropout1(out_proj(attn(x)))
    TRY(layernorm_bwd_launch(dY, rb, ws + lw.r1, params + lo.n1w, (const float*)(ws + lw.mean1), (const float*)(ws + lw.rstd1),
                             LnStreams{rb, nullptr, dR, rb}, b.RG(lo.n1w), b.RG(lo.n1b), M, H, s, rc, b.cast_to(k.cA2, pl, sl + 1, lo.ob)));
    TRY(wq.push(H, H, M, op(ws + k.cA2, es, 0, H, 0), op(ws + lw.ad, es, 0, H, 0), grads + lo.ow, H));
    {
        nbci_gemm_desc d = gd(M, H, H, dt, op(ws + k.cA2, es, 0, H, 1), op(x.W(lo.ow), es, 0, H, 0), ws + k.dAtt, H, dt);
        TRY(gemm_launch_timed(d, s));
    }
    const size_t pd = pl > 0.f ? lw.Pd : lw.P;
    const int64_t pz1 = (int64_t)nh * S * k.ldP, pz2 = (int64_t)S * k.ldP;
    const int64_t qz1 = (int64_t)S * 3 * H, az1 = (int64_t)S * H;
    if (k.small_attn) {
        if (k.flash)
            TRY(fattn_bwd_launch(ws + lw.qkv, ws + lw.ad, ws + k.dAtt, (const float*)(ws + lw.lse), (float*)(ws + k.dsum), ws + k.dqkv, Bq, nh, S,
                                 H, pl, b.seed, sl, s));
        else
            TRY(sattn_bwd_launch(ws + lw.qkv, ws + lw.ad, ws + k.dAtt, (const float*)(ws + lw.lse), (float*)(ws + k.dsum), ws + k.dqkv, dt, Bq, nh,
                                 S, H, pl, b.seed, sl, s));
    } else {
    {   // dPd = da v^T (f32, reuses the score buffer)
        nbci_gemm_desc d = gd(S, S, hd, dt, op(ws + k.dAtt, es, 0, H, 1, 0, 0, az1, hd),
                              op(ws + lw.qkv, es, 2 * H, 3 * H, 1, 0, 0, qz1, hd), ws + k.scores, k.ldS, NBCI_F32);
        d.batch = Bq * nh; d.zdiv = nh; d.czs1 = (int64_t)nh * S * k.ldS; d.czs2 = (int64_t)S * k.ldS;
        TRY(gemm_launch_timed(d, s));
    }
    {   // dv = Pd^T da
        nbci_gemm_desc d = gd(S, hd, S, dt, op(ws + pd, es, 0, k.ldP, 0, 0, 0, pz1, pz2), op(ws + k.dAtt, es, 0, H, 0, 0, 0, az1, hd),
                              (char*)(ws + k.dqkv) + (size_t)2 * H * es, 3 * H, dt);
        d.batch = Bq * nh; d.zdiv = nh; d.czs1 = qz1; d.czs2 = hd;
        TRY(gemm_launch_timed(d, s));
    }
    TRY(softmax_bwd_launch((const float*)(ws + k.scores), ws + lw.P, ws + k.dS, dt, Bq, nh, S, k.ldS, k.ldP, pl, b.seed, sl, s));
    {   // dq = dS k * scale
        nbci_gemm_desc d = gd(S, hd, S, dt, op(ws + k.dS, es, 0, k.ldP, 1, 0, 0, pz1, pz2),
                              op(ws + lw.qkv, es, H, 3 * H, 0, 0, 0, qz1, hd), ws + k.dqkv, 3 * H, dt);
        d.batch = Bq * nh; d.zdiv = nh; d.czs1 = qz1; d.czs2 = hd; d.alpha = scale;
        TRY(gemm_launch_timed(d, s));
    }
    {   // dk = dS^T q * scale
        nbci_gemm_desc d = gd(S, hd, S, dt, op(ws + k.dS, es, 0, k.ldP, 0, 0, 0, pz1, pz2),
                              op(ws + lw.qkv, es, 0, 3 * H, 0, 0, 0, qz1, hd), (char*)(ws + k.dqkv) + (size_t)H * es, 3 * H, dt);
        d.batch = Bq * nh; d.zdiv = nh; d.czs1 = qz1; d.czs2 = hd; d.alpha = scale;
        TRY(gemm_launch_timed(d, s));
    }
    }
    TRY(colsum_launch(ws + k.dqkv, dt, 3 * H, M, 3 * H, b.RG(lo.inb), s, rc));
    TRY(wq.push(3 * H, H, M, op(ws + k.dqkv, es, 0, 3 * H, 0), op(ws + lw.xb, es, 0, H, 0), grads + lo.inw, H));
    TRY(wq.flush());
    {   // d x = dqkv W_in + d r1
        nbci_gemm_desc d = gd(M, H, 3 * H, dt, op(ws + k.dqkv, es, 0, 3 * H, 1), op(x.W(lo.inw), es, 0, H, 0), dY, H, xdt);
        d.residual = dR; d.ldr = H; d.residual_dtype = xdt;
        TRY(gemm_launch_timed(d, s));
    }
    return NBCI_OK;
}

int itr_forward(const ItrPlan& p, const float* params, const void* params_lp, const nbci_itr_io* io, hipStream_t s) {
    TRY(itr_validate(p, io));
    const auto& c = p.c;
    NBCI_REQUIRE(params, NBCI_EINVAL, "itransformer: null params");
    NBCI_REQUIRE(c.dtype == NBCI_F32 || params_lp, NBCI_EINVAL, "itransformer: bf16 mode needs the bf16 parameter shadow");
    NBCI_REQUIRE(io->preds && io->loss && io->n_examples && io->mask_out, NBCI_EINVAL, "itransformer: preds, mask_out, loss, n_examples outputs are required");
    ItrCtx x{p, params, c.dtype == NBCI_BF16 ? params_lp : (const void*)params, (size_t)(c.dtype == NBCI_BF16 ? 2 : 4),
             (char*)io->workspace, {}, s};
    const int B = io->B, N = io->N;
    TRY(itr_carve(p, B, N, x.w));
    NBCI_REQUIRE((size_t)io->workspace_bytes >= x.w.bytes, NBCI_EWORKSPACE, "itransformer: workspace too small");
    const ItrWS& w = x.w;
    const int M = w.M, M0 = w.M0, H = c.hidden, T = c.max_n_bins;
    const int dt = c.dtype;
    const size_t es = x.es;
    const bool train = io->train != 0;
    const float pe = train ? c.embed_dropout : 0.f, pl = train ? c.dropout : 0.f;
    char* ws = x.ws;
    const int xdt = c.residual_dtype;
    const bool rb = xdt == NBCI_BF16;

    if (io->want_grad) {
        NBCI_CHECK_HIP(hipMemsetAsync(ws + w.rep, 0, (size_t)NREP * p.compact_total * 4, s));
        if (c.max_n_channels > 0) NBCI_CHECK_HIP(hipMemsetAsync(ws + w.dchtab, 0, (size_t)c.max_n_channels * H * 4, s));
        if (c.n_regions > 0) NBCI_CHECK_HIP(hipMemsetAsync(ws + w.drgtab, 0, (size_t)c.n_regions * H * 4, s));
    }
    // 0. channel-as-token view of the masked spikes (itransformer.py:187; the UnivariateTransformer walks the same (B,N,T) view, :86)
    TRY(btn_to_bnt_launch(io->masked, (float*)(ws + w.xs), B, T, N, s));
    float emb_tail_drop = pe;   // dropout between the embedder's last Linear and its LayerNorm (the torchvision MLP's trailing Dropout; none in mode 1)
    if (c.emb_mode == 0) {
        // 1. embedding MLP (torchvision MLP: Linear, act, Dropout, Linear, Dropout; itransformer.py:110-116). The first
        //    Linear has K = max_n_bins (100: rows of the bf16 shadow would not be 16-byte aligned) and 0.1 % of the
        //    FLOPs: it runs on the exact-f32 MFMA path straight from the f32 parameters.
        {
            nbci_gemm_desc d = gd(M0, H, T, NBCI_F32, op(ws + w.xs, 4, 0, T, 1), op(params, 4, p.e0w, T, 1), ws + w.h0, H, dt);
            d.bias = params + p.e0b; d.act = c.act; d.drop_p = pe; d.seed = io->seed; d.site = 4;
            TRY(gemm_launch_timed(d, s));
        }
        {
            nbci_gemm_desc d = gd(M0, H, H, dt, op(ws + w.h0, es, 0, H, 1), op(x.W(p.e3w), es, 0, H, 1), ws + w.t2, H, NBCI_F32);
            d.bias = params + p.e3b; d.drop_p = pe; d.seed = io->seed; d.site = 5;
            TRY(gemm_launch_timed(d, s));
        }
    } else {
        // 1'. UnivariateTransformer (itransformer.py:75-93) + embed_proj's Linear (:121-122). Every (sample, channel) is a sequence of
        //     1 + T rows [cls | bins]: bin token = Linear(act(Linear(count))) + embed_pos[timestamp]; post-norm encoder; the CLS row's output is
        //     the channel's embedding. (The module's own embed_dropout is never applied in its forward; the layers' dropout is embedder.dropout.)
        emb_tail_drop = 0.f;
        const StackWS& k = w.emb;
        const int h = c.emb_hidden, Me = k.M;
        TRY(scalar_lin_fwd_launch((const float*)(ws + w.xs), params + p.us0w, params + p.us0b, ws + w.uni_h, dt, Me, h, T + 1, s));
        float* t_in = (float*)(ws + k.dR);   // (backward scratch, free during the forward: f32, Me x h)
        {
            nbci_gemm_desc d = gd(Me, h, h, dt, op(ws + w.uni_h, es, 0, h, 1), op(x.W(p.us2w), es, 0, h, 1), t_in, h, NBCI_F32);
            d.bias = params + p.us2b;
            TRY(gemm_launch_timed(d, s));
        }
        TRY(uni_finish_fwd_launch(t_in, params + p.upos, io->spikes_timestamp, params + p.ucls, rb ? nullptr : (float*)(ws + k.yA), ws + stack_in_b(k),
                                  dt, B, N, T, h, s));
        TRY(stack_forward(x, p.emb, k, c.act, pe, io->seed, 128));
        {   // embed_proj.0 on the CLS rows: A = every (1 + T)-th row of the stack's output
            nbci_gemm_desc d = gd(M0, H, h, dt, op(ws + k.xo, es, 0, (int64_t)(T + 1) * h, 1), op(x.W(p.upw), es, 0, h, 1), ws + w.t2, H, NBCI_F32);
            d.bias = params + p.upb;
            TRY(gemm_launch_timed(d, s));
        }
    }
    (void)emb_tail_drop;
    // 2. LayerNorm'd embedding tables (itransformer.py:125-139,192-201): normalise the whole table once per step
    const int64_t* ss = io->spikes_spacestamp;
    if (c.max_n_channels > 0) {
        TRY(layernorm_fwd_launch(params + p.chw, params + p.chnw, params + p.chnb, ws + w.chtab, NBCI_F32, (float*)(ws + w.mean_c),
                                 (float*)(ws + w.rstd_c), c.max_n_channels, H, s));
        if (!ss) {
            hipLaunchKernelGGL(itr_iota_kernel, dim3((M0 + 255) / 256), dim3(256), 0, s, (long long*)(ws + w.ssidx), B, N);
            ss = (const int64_t*)(ws + w.ssidx);
        }
    }
    if (c.n_regions > 0)
        TRY(layernorm_fwd_launch(params + p.rgw, params + p.rgnw, params + p.rgnb, ws + w.rgtab, NBCI_F32, (float*)(ws + w.mean_r),
                                 (float*)(ws + w.rstd_r), c.n_regions, H, s));
    // 2'. depth embeddings (itransformer.py:143-150,200-202): LayerNorm(Linear(act(Linear(depth)))) per neuron, added to its token
    if (c.embed_depth) {
        TRY(scalar_lin_fwd_launch(io->neuron_depths, params + p.dp0w, params + p.dp0b, ws + w.dep_h, dt, M0, H, 0, s));
        nbci_gemm_desc d = gd(M0, H, H, dt, op(ws + w.dep_h, es, 0, H, 1), op(x.W(p.dp2w), es, 0, H, 1), ws + w.dep_t, H, NBCI_F32);
        d.bias = params + p.dp2b;
        TRY(gemm_launch_timed(d, s));
        TRY(layernorm_fwd_launch((const float*)(ws + w.dep_t), params + p.dpnw, params + p.dpnb, ws + w.dep_o, NBCI_F32, (float*)(ws + w.mean_d),
                                 (float*)(ws + w.rstd_d), M0, H, s));
    }
    // 3. tokens = [cls | LN(t2) + channel (+ region) (+ depth)] -> embed dropout (itransformer.py:187-209)
    const StackWS& ke = w.enc;
    TRY(itr_assemble_fwd_launch((const float*)(ws + w.t2), params + p.enw, params + p.enb,
                                c.max_n_channels > 0 ? (const float*)(ws + w.chtab) : nullptr, ss,
                                c.n_regions > 0 ? (const float*)(ws + w.rgtab) : nullptr, io->region_idx,
                                c.use_cls ? params + p.cls : nullptr, (float*)(ws + ke.yA), ws + stack_in_b(ke), dt, (float*)(ws + w.mean_e),
                                (float*)(ws + w.rstd_e), B, N, H, c.use_cls ? 1 : 0, pe, io->seed, 6, s,
                                c.embed_depth ? (const float*)(ws + w.dep_o) : nullptr));
    TRY(stack_forward(x, p.enc, ke, c.act, pl, io->seed, 16));
    if (io->hidden_out)
        NBCI_CHECK_HIP(hipMemcpyAsync(io->hidden_out, ws + ke.xo, (size_t)M * H * es, hipMemcpyDeviceToDevice, s));
    // decoder over every token row (the CLS rows are computed and ignored: 1/(N+1) extra work, no gather)
    const void* dec_in = ws + ke.xo;
    int64_t ow = p.d0w, ob = p.d0b;
    if (c.mlp_decoder) {
        nbci_gemm_desc d = gd(M, H, H, dt, op(ws + ke.xo, es, 0, H, 1), op(x.W(p.d0w), es, 0, H, 1), ws + w.d1, H, dt);
        d.bias = params + p.d0b; d.act = c.dec_act;
        TRY(gemm_launch_timed(d, s));
        dec_in = ws + w.d1; ow = p.d2w; ob = p.d2b;
    }
    {
        nbci_gemm_desc d = gd(M, T, H, dt, op(dec_in, es, 0, H, 1), op(x.W(ow), es, 0, H, 1), ws + w.pred, w.ldT, NBCI_F32);
        d.bias = params + ob;
        TRY(gemm_launch_timed(d, s));
    }
    TRY(itr_mlm_loss_launch((const float*)(ws + w.pred), w.ldT, io->spikes, io->mask, io->spikes_mask, io->preds, io->mask_out,
                            io->want_grad ? ws + w.dpred : nullptr, dt, io->loss, io->n_examples, B, T, N, c.use_cls ? 1 : 0, c.loss,
                            io->grad_scale, s));
    return NBCI_OK;
}

int itr_backward(const ItrPlan& p, const float* params, const void* params_lp, const nbci_itr_io* io, float* grads, int seg_hi, int seg_lo,
                 hipStream_t s) {
    TRY(itr_validate(p, io));
    const auto& c = p.c;
    NBCI_REQUIRE(params && grads, NBCI_EINVAL, "itransformer: null params/grads");
    NBCI_REQUIRE(c.dtype == NBCI_F32 || params_lp, NBCI_EINVAL, "itransformer: bf16 mode needs the bf16 parameter shadow");
    NBCI_REQUIRE(seg_hi <= c.n_layers + 1 && seg_lo >= 0 && seg_lo <= seg_hi, NBCI_EINVAL, "itransformer: bad segment range");
    ItrCtx x{p, params, c.dtype == NBCI_BF16 ? params_lp : (const void*)params, (size_t)(c.dtype == NBCI_BF16 ? 2 : 4),
             (char*)io->workspace, {}, s};
    const int B = io->B, N = io->N;
    TRY(itr_carve(p, B, N, x.w));
    NBCI_REQUIRE((size_t)io->workspace_bytes >= x.w.bytes, NBCI_EWORKSPACE, "itransformer: workspace too small");
    const ItrWS& w = x.w;
    const StackWS& ke = w.enc;
    const int M = w.M, M0 = w.M0, H = c.hidden, T = c.max_n_bins;
    const int dt = c.dtype;
    const size_t es = x.es;
    const bool train = io->train != 0;
    const float pe = train ? c.embed_dropout : 0.f, pl = train ? c.dropout : 0.f;
    char* ws = x.ws;
    float* dY = (float*)(ws + ke.dY);   // (bf16 elements with residual_dtype bf16, as dR, r1, r2)
    float* dR = (float*)(ws + ke.dR);
    const int xdt = c.residual_dtype;
    float* rep = (float*)(ws + w.rep);
    const RepCfg rc{p.compact_total, NREP};
    const StackBwd be{x, p.enc, ke, grads, rep, rc, c.act, pl, io->seed, 16};
    auto RG = [&](int64_t flat_off) -> float* { return rep + p.compact_of(flat_off); };

    for (int seg = seg_hi; seg >= seg_lo; --seg) {
        if (seg == c.n_layers + 1) {
            // ---- mlm decoder + final norm
            const void* dp = ws + w.dpred;
            const void* dec_in = c.mlp_decoder ? ws + w.d1 : ws + ke.xo;
            const int64_t ow = c.mlp_decoder ? p.d2w : p.d0w, ob = c.mlp_decoder ? p.d2b : p.d0b;
            TRY(colsum_launch(dp, dt, w.ldT, M, T, RG(ob), s, rc));
            TRY(wgrad(s, dt, T, H, M, op(dp, es, 0, w.ldT, 0), op(dec_in, es, 0, H, 0), grads + ow, H));
            if (c.mlp_decoder) {
                {   // dd1 = (dpred W_2) * act'(d1), decoder.0 bias grad = its column sums
                    nbci_gemm_desc d = gd(M, H, T, dt, op(dp, es, 0, w.ldT, 1), op(x.W(p.d2w), es, 0, H, 0), ws + ke.cA, H, dt);
                    d.gate = ws + w.d1; d.ldg = H; d.gate_act = c.dec_act;
                    d.colsum = RG(p.d0b); d.colsum_rep_stride = rc.stride; d.colsum_nrep = rc.n;
                    TRY(gemm_launch_timed(d, s));
                }
                TRY(wgrad(s, dt, H, H, M, op(ws + ke.cA, es, 0, H, 0), op(ws + ke.xo, es, 0, H, 0), grads + p.d0w, H));
                nbci_gemm_desc d = gd(M, H, H, dt, op(ws + ke.cA, es, 0, H, 1), op(x.W(p.d0w), es, 0, H, 0), dR, H, xdt);
                TRY(gemm_launch_timed(d, s));
            } else {
                nbci_gemm_desc d = gd(M, H, T, dt, op(dp, es, 0, w.ldT, 1), op(x.W(p.d0w), es, 0, H, 0), dR, H, xdt);
                TRY(gemm_launch_timed(d, s));
            }
            TRY(stack_backward_final(be));
        } else if (seg >= 1) {
            TRY(stack_backward_layer(be, seg - 1));
        } else {
            // ---- embedding side (itransformer.py:187-209)
            float* dtok = (float*)(ws + w.dtok);
            const int64_t* ss = io->spikes_spacestamp ? io->spikes_spacestamp : (const int64_t*)(ws + w.ssidx);
            TRY(itr_assemble_bwd_launch(dY, dtok, c.max_n_channels > 0 ? (float*)(ws + w.dchtab) : nullptr, ss,
                                        c.n_regions > 0 ? (float*)(ws + w.drgtab) : nullptr, io->region_idx,
                                        c.use_cls ? RG(p.cls) : nullptr, rc, B, N, H, c.use_cls ? 1 : 0, pe, io->seed, 6, s, xdt));
            auto cast_e = [&](size_t buf, float pp, uint32_t site, int64_t bias_off) -> LnCast {
                return LnCast{ws + buf, dt == NBCI_BF16, drop_threshold(pp), pp > 0.f ? 1.f / (1.f - pp) : 1.f, drop_key(io->seed, site), RG(bias_off)};
            };
            if (c.embed_depth) {   // depth_embeddings: LayerNorm -> Linear(H,H) -> relu -> Linear(1,H)
                TRY(layernorm_bwd_launch(dtok, (const float*)(ws + w.dep_t), params + p.dpnw, (const float*)(ws + w.mean_d),
                                         (const float*)(ws + w.rstd_d), dR, RG(p.dpnw), RG(p.dpnb), M0, H, 0, s, rc, cast_e(w.dep_c, 0.f, 0, p.dp2b)));
                TRY(wgrad(s, dt, H, H, M0, op(ws + w.dep_c, es, 0, H, 0), op(ws + w.dep_h, es, 0, H, 0), grads + p.dp2w, H));
                nbci_gemm_desc d = gd(M0, H, H, dt, op(ws + w.dep_c, es, 0, H, 1), op(x.W(p.dp2w), es, 0, H, 0), ws + w.dep_du, H, NBCI_F32);
                d.gate = ws + w.dep_h; d.ldg = H; d.gate_act = c.act;
                TRY(gemm_launch_timed(d, s));
                TRY(scalar_lin_bwd_launch(ws + w.dep_du, NBCI_F32, io->neuron_depths, RG(p.dp0w), RG(p.dp0b), rc, M0, H, 0, s));
            }
            if (c.emb_mode == 0) {
                // embed.1 LayerNorm; its output gradient feeds embed.0.3 through the MLP's trailing Dropout (site 5)
                TRY(layernorm_bwd_launch(dtok, (const float*)(ws + w.t2), params + p.enw, (const float*)(ws + w.mean_e),
                                         (const float*)(ws + w.rstd_e), dR, RG(p.enw), RG(p.enb), M0, H, 0, s, rc, cast_e(ke.cA, pe, 5, p.e3b)));
                TRY(wgrad(s, dt, H, H, M0, op(ws + ke.cA, es, 0, H, 0), op(ws + w.h0, es, 0, H, 0), grads + p.e3w, H));
                {   // d u0 = (c W_3) * act'(u0) * keep -> f32 (the K = max_n_bins weight gradient runs on the f32 path)
                    nbci_gemm_desc d = gd(M0, H, H, dt, op(ws + ke.cA, es, 0, H, 1), op(x.W(p.e3w), es, 0, H, 0), ws + w.dH0, H, NBCI_F32);
                    d.gate = ws + w.h0; d.ldg = H; d.gate_act = c.act;
                    d.drop_p = pe; d.seed = io->seed; d.site = 4;
                    d.colsum = RG(p.e0b); d.colsum_rep_stride = rc.stride; d.colsum_nrep = rc.n;
                    TRY(gemm_launch_timed(d, s));
                }
                TRY(wgrad(s, NBCI_F32, H, T, M0, op(ws + w.dH0, 4, 0, H, 0), op(ws + w.xs, 4, 0, T, 0), grads + p.e0w, T));
            } else {
                const StackWS& k = w.emb;
                const int h = c.emb_hidden, Me = k.M;
                const StackBwd bi{x, p.emb, k, grads, rep, rc, c.act, pe, io->seed, 128};
                // embed_proj.1 LayerNorm -> embed_proj.0 (no dropout in between)
                TRY(layernorm_bwd_launch(dtok, (const float*)(ws + w.t2), params + p.enw, (const float*)(ws + w.mean_e),
                                         (const float*)(ws + w.rstd_e), dR, RG(p.enw), RG(p.enb), M0, H, 0, s, rc, cast_e(ke.cA, 0.f, 0, p.upb)));
                TRY(wgrad(s, dt, H, h, M0, op(ws + ke.cA, es, 0, H, 0), op(ws + k.xo, es, 0, (int64_t)(T + 1) * h, 0), grads + p.upw, h));
                // d(stack output): zero except the CLS rows = c W_p
                const size_t rs = xdt == NBCI_BF16 ? 2 : 4;
                NBCI_CHECK_HIP(hipMemsetAsync(ws + k.dR, 0, (size_t)Me * h * rs, s));
                {
                    nbci_gemm_desc d = gd(M0, h, H, dt, op(ws + ke.cA, es, 0, H, 1), op(x.W(p.upw), es, 0, h, 0), ws + k.dR, (int64_t)(T + 1) * h, xdt);
                    TRY(gemm_launch_timed(d, s));
                }
                TRY(stack_backward_final(bi));
                for (int l = c.emb_layers - 1; l >= 0; --l) TRY(stack_backward_layer(bi, l));
                // the stack's input: cls_embed (CLS rows), embed_pos (summed over channels), embed_spikes (Linear, relu, Linear(1 -> h))
                TRY(uni_split_bwd_launch(ws + k.dY, xdt, ws + k.cA, dt, RG(p.ucls), rc, Me, T, h, s));
                TRY(uni_posgrad_launch(ws + k.dY, xdt, io->spikes_timestamp, grads + p.upos, B, N, T, h, s));
                TRY(colsum_launch(ws + k.cA, dt, h, Me, h, RG(p.us2b), s, rc));
                TRY(wgrad(s, dt, h, h, Me, op(ws + k.cA, es, 0, h, 0), op(ws + w.uni_h, es, 0, h, 0), grads + p.us2w, h));
                {
                    nbci_gemm_desc d = gd(Me, h, h, dt, op(ws + k.cA, es, 0, h, 1), op(x.W(p.us2w), es, 0, h, 0), ws + w.uni_du, h, NBCI_F32);
                    d.gate = ws + w.uni_h; d.ldg = h; d.gate_act = c.act;
                    TRY(gemm_launch_timed(d, s));
                }
                TRY(scalar_lin_bwd_launch(ws + w.uni_du, NBCI_F32, (const float*)(ws + w.xs), RG(p.us0w), RG(p.us0b), rc, Me, h, T + 1, s));
            }
            // LayerNorm'd tables: the scatter-added table gradient goes back through the table's LayerNorm
            if (c.max_n_channels > 0)
                TRY(layernorm_bwd_launch((const float*)(ws + w.dchtab), params + p.chw, params + p.chnw, (const float*)(ws + w.mean_c),
                                         (const float*)(ws + w.rstd_c), grads + p.chw, RG(p.chnw), RG(p.chnb), c.max_n_channels, H, 1, s, rc,
                                         NO_CAST));
            if (c.n_regions > 0)
                TRY(layernorm_bwd_launch((const float*)(ws + w.drgtab), params + p.rgw, params + p.rgnw, (const float*)(ws + w.mean_r),
                                         (const float*)(ws + w.rstd_r), grads + p.rgw, RG(p.rgnw), RG(p.rgnb), c.n_regions, H, 1, s, rc,
                                         NO_CAST));
        }
    }
    // the replicated small-vector gradients of every segment of this call, folded in ONE launch (their compact ranges are adjacent)
    TRY(fold_replicas_launch(rep, rc.stride, rc.n, p.d_flat_of, p.cseg[seg_lo].first, p.cseg[seg_hi].second, grads, s));
    return NBCI_OK;
}

}  // namespace nbci

using namespace nbci;

extern "C" {

int nbci_masker(const nbci_masker_desc* desc, nbci_stream_t stream) {
    if (!desc) return fail(NBCI_EINVAL, "masker: null descriptor");
    return masker_launch(*desc, (hipStream_t)stream);
}

int nbci_itr_plan_create(const nbci_itr_config* cfg, nbci_itr_plan* out) {
    if (!cfg || !out) return fail(NBCI_EINVAL, "itr plan_create: null argument");
    const nbci_itr_config& c = *cfg;
    NBCI_REQUIRE(c.hidden > 0 && c.n_heads > 0 && c.hidden % c.n_heads == 0, NBCI_ESHAPE, "embed_dim must be divisible by num_heads");
    NBCI_REQUIRE(c.hidden % 8 == 0 && (c.hidden / c.n_heads) % 8 == 0, NBCI_ESHAPE, "hidden and head size must be multiples of 8");
    NBCI_REQUIRE(c.max_n_bins > 0 && c.max_n_bins % 4 == 0, NBCI_ESHAPE, "max_n_bins must be a positive multiple of 4");
    NBCI_REQUIRE(c.emb_mode == 0 || c.emb_mode == 1, NBCI_EINVAL, "emb_mode must be 0 (mlp) or 1 (transformer)");
    if (c.emb_mode == 1) {
        NBCI_REQUIRE(c.emb_hidden > 0 && c.emb_heads > 0 && c.emb_layers >= 0 && c.emb_hidden % c.emb_heads == 0, NBCI_ESHAPE,
                     "embedder: embed_dim must be divisible by num_heads");
        NBCI_REQUIRE(c.emb_hidden % 8 == 0 && (c.emb_hidden / c.emb_heads) % 8 == 0, NBCI_ESHAPE, "embedder hidden and head size must be multiples of 8");
    }
    NBCI_REQUIRE(c.n_layers >= 0 && c.max_n_channels >= 0 && c.n_regions >= 0, NBCI_ESHAPE, "bad iTransformer shape parameters");
    NBCI_REQUIRE(c.dtype == NBCI_F32 || c.dtype == NBCI_BF16, NBCI_EINVAL, "dtype must be f32 or bf16");
    NBCI_REQUIRE(c.residual_dtype == NBCI_F32 || (c.residual_dtype == NBCI_BF16 && c.dtype == NBCI_BF16), NBCI_EINVAL,
                 "residual_dtype must be f32, or bf16 together with dtype bf16");
    NBCI_REQUIRE(c.act == ACT_RELU && (!c.mlp_decoder || c.dec_act == ACT_RELU), NBCI_EINVAL,
                 "iTransformer HIP path supports activation: relu (the backward reads act' off the saved outputs)");
    NBCI_REQUIRE(c.loss >= NBCI_LOSS_POISSON_LOG && c.loss <= NBCI_LOSS_MSE, NBCI_EINVAL, "unknown loss");
    ItrPlan* p = new ItrPlan();
    p->c = c;
    itr_layout(*p);
    p->d_flat_of = nullptr;
    hipError_t e = hipMalloc(&p->d_flat_of, std::max<size_t>(4, p->flat_of.size() * sizeof(int)));
    if (e == hipSuccess && !p->flat_of.empty())
        e = hipMemcpy(p->d_flat_of, p->flat_of.data(), p->flat_of.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete p; return fail(NBCI_EHIP, std::string("itr plan_create: ") + hipGetErrorString(e)); }
    *out = (nbci_itr_plan)p;
    return NBCI_OK;
}

void nbci_itr_plan_destroy(nbci_itr_plan plan) {
    ItrPlan* p = (ItrPlan*)plan;
    if (!p) return;
    if (p->d_flat_of) (void)hipFree(p->d_flat_of);
    delete p;
}

int64_t nbci_itr_param_count(nbci_itr_plan plan) { return plan ? ((ItrPlan*)plan)->total : -1; }
int32_t nbci_itr_num_params(nbci_itr_plan plan) { return plan ? (int32_t)((ItrPlan*)plan)->params.size() : -1; }
int32_t nbci_itr_num_segments(nbci_itr_plan plan) { return plan ? (int32_t)((ItrPlan*)plan)->seg.size() : -1; }

int nbci_itr_param_info(nbci_itr_plan plan, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel, int32_t* rows,
                        int32_t* cols, int32_t* segment) {
    ItrPlan* p = (ItrPlan*)plan;
    if (!p || index < 0 || index >= (int)p->params.size()) return fail(NBCI_EINVAL, "itr param_info: bad plan/index");
    const PInfo& i = p->params[index];
    if (name && name_cap > 0) { strncpy(name, i.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = i.off;
    if (numel) *numel = i.numel;
    if (rows) *rows = i.rows;
    if (cols) *cols = i.cols;
    if (segment) *segment = i.seg;
    return NBCI_OK;
}

int nbci_itr_segment_range(nbci_itr_plan plan, int32_t seg, int64_t* begin, int64_t* end) {
    ItrPlan* p = (ItrPlan*)plan;
    if (!p || seg < 0 || seg >= (int)p->seg.size()) return fail(NBCI_EINVAL, "itr segment_range: bad plan/segment");
    *begin = p->seg[seg].first; *end = p->seg[seg].second;
    return NBCI_OK;
}

int64_t nbci_itr_workspace_bytes(nbci_itr_plan plan, int32_t B, int32_t N) {
    ItrPlan* p = (ItrPlan*)plan;
    if (!p) { fail(NBCI_EINVAL, "itr workspace_bytes: null plan"); return -1; }
    ItrWS w;
    if (itr_carve(*p, B, N, w) != NBCI_OK) return -1;
    return (int64_t)w.bytes;
}

int nbci_itr_forward(nbci_itr_plan plan, const float* params, const void* params_lp, const nbci_itr_io* io, nbci_stream_t stream) {
    if (!plan) return fail(NBCI_EINVAL, "itr forward: null plan");
    return itr_forward(*(ItrPlan*)plan, params, params_lp, io, (hipStream_t)stream);
}

int nbci_itr_backward(nbci_itr_plan plan, const float* params, const void* params_lp, const nbci_itr_io* io, float* grads, int32_t seg_hi,
                      int32_t seg_lo, nbci_stream_t stream) {
    if (!plan) return fail(NBCI_EINVAL, "itr backward: null plan");
    return itr_backward(*(ItrPlan*)plan, params, params_lp, io, grads, seg_hi, seg_lo, (hipStream_t)stream);
}

}  // extern "C"
