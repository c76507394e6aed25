"""numpy restatement of the reference's iTransformer SSL path: Masker (models/masker.py:44-110),
iTransformerEncoder.forward (models/itransformer.py:175-210, ctor :107-173) with either embedder - `mlp` (:108-118) or `transformer`
(UnivariateTransformer, :40-93, + embed_proj :119-124) -, channel / region (:133-141,195-198) / depth (:143-150,200-202) embeddings,
and iTransformer.forward, method 'mlm' (:312-359), forward AND hand-derived backward.
Test infrastructure only — see oracle/__init__.py.

Third-party arithmetic restated from its published definition (versions unpinned by the reference; pinned
here by tests/golden/g_itr_*.npz generated against torch 2.10.0):
  * torchvision.ops.MLP (itransformer.py:110-116) = Linear, act, Dropout, Linear, Dropout;
  * torch.nn.TransformerEncoderLayer, norm_first=False (itransformer.py:158-173):
        x = norm1(x + dropout1(MHA(x)));  x = norm2(x + dropout2(linear2(dropout(act(linear1(x))))))
    MHA = packed in_proj (3H,H), softmax(q k^T / sqrt(hd)) with dropout on the probabilities, out_proj; no mask;
  * torch.nn.PoissonNLLLoss(reduction='none', log_input, full=False, eps=1e-8), nn.MSELoss.
Parameter names are the reference's state-dict keys.
"""
import math

import numpy as np

from . import rng as R
from .ndt1 import act_bwd, act_fwd, layer_norm, layer_norm_bwd

# dropout sites (shared numbering with csrc/itransformer.hip)
SITE_EMB_HID, SITE_EMB_OUT, SITE_TOKENS = 4, 5, 6
SITE_MASKER = 64  # + 8 * (index of the masker in the config dict): +0 mask, +1 zero, +2 random-select, +3 random values, +4 timespan


def site_layer(l, k):
    """k: 0 attention probabilities, 1 dropout1, 2 FFN inner dropout, 3 dropout2."""
    return 16 + 4 * l + k


def site_emb_layer(l, k):
    """the UnivariateTransformer embedder's layers (same k as site_layer)"""
    return 128 + 4 * l + k


# n_regions > 0: embed_region with that many table rows (batch["region_idx"] (B,N) = the reference's region_to_indx of neuron_regions);
# embed_depth: batch["neuron_depths"] (B,N) f32; embedder_mode "transformer": emb_* = configs/itransformer.yaml encoder.embedder.*
DEFAULTS = dict(max_n_bins=100, hidden=768, n_heads=8, n_layers=5, max_n_channels=1500, act="relu", embed_dropout=0.2,
                dropout=0.4, use_cls=True, mlp_decoder=True, dec_act="relu", log_input=True, loss="poisson_nll",
                n_regions=0, embed_depth=False, embedder_mode="mlp", emb_hidden=128, emb_heads=4, emb_layers=4, emb_act="relu")


def make_config(**kw):
    c = dict(DEFAULTS)
    c.update(kw)
    return c


# ----------------------------------------------------------------------------- masker
def uniform(seed, site, n):
    """n uniforms in [0,1) with 24 bits, as the device draws them."""
    return (R.rng_u32(seed, site, np.arange(n, dtype=np.uint32)) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def expand_timesteps(mask, width):
    """masker.py:107-110: conv1d(mask, ones(width), padding='same') >= 1; 'same' pads (width-1)//2 on the left."""
    B, T = mask.shape
    left = (width - 1) // 2
    out = np.zeros_like(mask)
    for j in range(width):
        sh = j - left
        lo, hi = max(0, -sh), min(T, T - sh)
        out[:, lo:hi] |= mask[:, lo + sh:hi + sh]
    return out


def masker_timespan(seed, site, expand_prob, max_timespan):
    """masker.py:55-59 with the counter RNG: (timespan) drawn on the host from site+4."""
    u = uniform(seed, site + 4, 2)
    if u[0] < np.float32(expand_prob):
        return 1 + int(R.rng_u32(seed, site + 4, np.array([1], np.uint32))[0] % np.uint32(max_timespan))
    return 1


def region_sample(seed, site, regions, n):
    """random.sample(regions, n) ("masker copy.py":91,99) drawn from the counter RNG at site + 5 (partial Fisher-Yates), as the
    host side of the HIP path draws it."""
    pool = list(regions)
    for j in range(int(n)):
        k = j + int(R.rng_u32(seed, site + 5, np.array([j], np.uint32))[0] % np.uint32(len(pool) - j))
        pool[j], pool[k] = pool[k], pool[j]
    return pool[:int(n)]


def masker(mc, spikes, training, seed, site, probs=None, neuron_regions=None):
    """Masker.forward (masker.py:44-104) plus the modes of "models/masker copy.py":81-104,117,133 (forward-pred, inter-region,
    intra-region). mc: dict(active, force_active, mode, ratio, zero_ratio, random_ratio, expand_prob, max_timespan, channels,
    timesteps, mask_regions, target_regions, n_mask_regions). `probs`: (B,N) 0/1 table for `region` mode (the reference builds it
    from region names on the host); `neuron_regions`: (B,N) region names for the inter- / intra-region modes.
    Returns (masked copy, int64 mask); the reference mutates `spikes` in place, this does not."""
    B, T, N = spikes.shape
    if not mc.get("active", True) or (not training and not mc.get("force_active", False)):
        return spikes.copy(), np.zeros((B, T, N), np.int64)
    mode, ratio = mc["mode"], float(mc["ratio"])
    if mode == "temporal":
        span = masker_timespan(seed, site, mc.get("expand_prob", 0.0), mc.get("max_timespan", 1))
        m = (uniform(seed, site, B * T) < np.float32(ratio / span)).reshape(B, T)
        if span > 1:
            m = expand_timesteps(m, span)
        mask = np.broadcast_to(m[:, :, None], (B, T, N))
    elif mode == "neuron":
        mask = np.broadcast_to((uniform(seed, site, B * N) < np.float32(ratio)).reshape(B, 1, N), (B, T, N))
    elif mode == "random":
        mask = (uniform(seed, site, B * T * N) < np.float32(ratio)).reshape(B, T, N)
    elif mode == "region":
        mask = np.broadcast_to((uniform(seed, site, B * N).reshape(B, N) < probs.astype(np.float32))[:, None, :], (B, T, N))
    elif mode == "co-smooth":
        pr = np.zeros(N, np.float32)
        pr[list(mc["channels"])] = 1
        mask = np.broadcast_to((uniform(seed, site, N) < pr)[None, None, :], (B, T, N))
    elif mode == "forward-pred":
        pr = np.zeros(T, np.float32)
        pr[list(mc["timesteps"])] = 1
        mask = np.broadcast_to((uniform(seed, site, T) < pr)[None, :, None], (B, T, N))
    elif mode == "inter-region":
        chosen = region_sample(seed, site, mc["mask_regions"], mc.get("n_mask_regions", 1))
        pr = np.isin(np.asarray(neuron_regions), chosen).astype(np.float32) * np.float32(ratio)
        mask = np.broadcast_to((uniform(seed, site, B * N).reshape(B, N) < pr)[:, None, :], (B, T, N))
    elif mode == "intra-region":
        chosen = region_sample(seed, site, mc["target_regions"], mc.get("n_mask_regions", 1))
        tgt = np.isin(np.asarray(neuron_regions), chosen)
        pr = np.where(tgt, np.float32(ratio), np.float32(1.0)).astype(np.float32)
        mask = np.broadcast_to((uniform(seed, site, B * N).reshape(B, N) < pr)[:, None, :], (B, T, N))
        out, full = apply_mask(mc, spikes, mask, seed, site)
        return out, full * tgt[:, None, :].astype(np.int64)      # targets: the masked bins of the target regions only (:133)
    else:
        raise Exception(f"Masking mode {mode} not implemented")
    return apply_mask(mc, spikes, mask, seed, site)


def apply_mask(mc, spikes, mask, seed, site):
    """masker.py:95-102: zero `zero_ratio` of the masked bins, randomise `random_ratio` of the rest with U(0, max)."""
    B, T, N = spikes.shape
    mask = np.ascontiguousarray(mask).astype(bool)
    out = spikes.copy()
    zero = (uniform(seed, site + 1, B * T * N) < np.float32(mc["zero_ratio"])).reshape(B, T, N) & mask
    out[zero] = 0
    rnd = (uniform(seed, site + 2, B * T * N) < np.float32(mc["random_ratio"])).reshape(B, T, N) & mask & ~zero
    vals = (out.max() * uniform(seed, site + 3, B * T * N).reshape(B, T, N)).astype(spikes.dtype)
    out[rnd] = vals[rnd]
    return out, mask.astype(np.int64)


# ----------------------------------------------------------------------------- model
def _lin_init(g, o, i, dtype):
    b = 1 / math.sqrt(i)
    return g.uniform(-b, b, (o, i)).astype(dtype), g.uniform(-b, b, (o,)).astype(dtype)


def _stack_init(p, g, pre, H, L, dtype):
    for l in range(L):
        q = f"{pre}layers.{l}."
        p[q + "self_attn.in_proj_weight"], p[q + "self_attn.in_proj_bias"] = _lin_init(g, 3 * H, H, dtype)
        p[q + "self_attn.out_proj.weight"], p[q + "self_attn.out_proj.bias"] = _lin_init(g, H, H, dtype)
        p[q + "linear1.weight"], p[q + "linear1.bias"] = _lin_init(g, 4 * H, H, dtype)
        p[q + "linear2.weight"], p[q + "linear2.bias"] = _lin_init(g, H, 4 * H, dtype)
        for nm in ("norm1", "norm2"):
            p[q + nm + ".weight"], p[q + nm + ".bias"] = np.ones(H, dtype), np.zeros(H, dtype)
    p[pre + "norm.weight"], p[pre + "norm.bias"] = np.ones(H, dtype), np.zeros(H, dtype)


def init_params(cfg, seed=0, dtype=np.float32):
    """Random parameters with the reference's shapes (NOT torch's init stream; golden tests load the reference's weights)."""
    g = np.random.default_rng(seed)
    T, H, L, C = cfg["max_n_bins"], cfg["hidden"], cfg["n_layers"], cfg["max_n_channels"]
    p = {}
    if cfg["embedder_mode"] == "mlp":
        p["encoder.embed.0.0.weight"], p["encoder.embed.0.0.bias"] = _lin_init(g, H, T, dtype)
        p["encoder.embed.0.3.weight"], p["encoder.embed.0.3.bias"] = _lin_init(g, H, H, dtype)
        p["encoder.embed.1.weight"], p["encoder.embed.1.bias"] = np.ones(H, dtype), np.zeros(H, dtype)
    else:
        h = cfg["emb_hidden"]
        p["encoder.embed.embed_spikes.0.weight"], p["encoder.embed.embed_spikes.0.bias"] = _lin_init(g, h, 1, dtype)
        p["encoder.embed.embed_spikes.2.weight"], p["encoder.embed.embed_spikes.2.bias"] = _lin_init(g, h, h, dtype)
        p["encoder.embed.embed_pos.weight"] = g.standard_normal((T, h)).astype(dtype)
        p["encoder.embed.cls_embed.weight"] = g.standard_normal((1, h)).astype(dtype)
        _stack_init(p, g, "encoder.embed.transformer.", h, cfg["emb_layers"], dtype)
        p["encoder.embed_proj.0.weight"], p["encoder.embed_proj.0.bias"] = _lin_init(g, H, h, dtype)
        p["encoder.embed_proj.1.weight"], p["encoder.embed_proj.1.bias"] = np.ones(H, dtype), np.zeros(H, dtype)
    if C:
        p["encoder.channel_embeddings.0.weight"] = g.standard_normal((C, H)).astype(dtype)
        p["encoder.channel_embeddings.1.weight"], p["encoder.channel_embeddings.1.bias"] = np.ones(H, dtype), np.zeros(H, dtype)
    if cfg["n_regions"]:
        p["encoder.region_embeddings.0.weight"] = g.standard_normal((cfg["n_regions"], H)).astype(dtype)
        p["encoder.region_embeddings.1.weight"], p["encoder.region_embeddings.1.bias"] = np.ones(H, dtype), np.zeros(H, dtype)
    if cfg["embed_depth"]:
        p["encoder.depth_embeddings.0.weight"], p["encoder.depth_embeddings.0.bias"] = _lin_init(g, H, 1, dtype)
        p["encoder.depth_embeddings.2.weight"], p["encoder.depth_embeddings.2.bias"] = _lin_init(g, H, H, dtype)
        p["encoder.depth_embeddings.3.weight"], p["encoder.depth_embeddings.3.bias"] = np.ones(H, dtype), np.zeros(H, dtype)
    if cfg["use_cls"]:
        p["encoder.cls_embed.weight"] = g.standard_normal((1, H)).astype(dtype)
    _stack_init(p, g, "encoder.transformer.", H, L, dtype)
    if cfg["mlp_decoder"]:
        p["decoder.0.weight"], p["decoder.0.bias"] = _lin_init(g, H, H, dtype)
        p["decoder.2.weight"], p["decoder.2.bias"] = _lin_init(g, T, H, dtype)
    else:
        p["decoder.0.weight"], p["decoder.0.bias"] = _lin_init(g, T, H, dtype)
    return p


def _head_names(cfg):
    return ("decoder.0", "decoder.2") if cfg["mlp_decoder"] else (None, "decoder.0")


def stack_fwd(P, pre, x, nh, L, act, pl, seed, site, f):
    """torch.nn.TransformerEncoder of L post-norm TransformerEncoderLayers + final norm (itransformer.py:58-73 / :158-173; module
    docstring) on x (Bq, S, H); parameters P[pre + 'layers.N. ...'], P[pre + 'norm. ...']; dropout p = pl with the counter RNG at
    site(l, k). Returns (output after the final norm, cache)."""
    Bq, S, H = x.shape
    hd = H // nh
    scale = f(1.0 / math.sqrt(hd))

    def heads(t):
        return t.reshape(Bq, S, nh, hd).transpose(0, 2, 1, 3)

    layers = []
    for l in range(L):
        q_ = f"{pre}layers.{l}."
        lc = {"x_in": x}
        qkv = x @ P[q_ + "self_attn.in_proj_weight"].T + P[q_ + "self_attn.in_proj_bias"]
        q, k, v = heads(qkv[..., :H]), heads(qkv[..., H:2 * H]), heads(qkv[..., 2 * H:])
        s = (q @ k.transpose(0, 1, 3, 2)) * scale
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        prob = e / e.sum(-1, keepdims=True)
        pm = R.keep_mask(seed, site(l, 0), Bq * nh * S * S, pl).reshape(Bq, nh, S, S).astype(f)
        pd = prob * pm
        a = (pd @ v).transpose(0, 2, 1, 3).reshape(Bq, S, H)
        d1 = R.keep_mask(seed, site(l, 1), Bq * S * H, pl).reshape(Bq, S, H).astype(f)
        r1 = x + (a @ P[q_ + "self_attn.out_proj.weight"].T + P[q_ + "self_attn.out_proj.bias"]) * d1
        x1, lc["xhat1"], lc["rstd1"] = layer_norm(r1, P[q_ + "norm1.weight"], P[q_ + "norm1.bias"])
        u = x1 @ P[q_ + "linear1.weight"].T + P[q_ + "linear1.bias"]
        di = R.keep_mask(seed, site(l, 2), Bq * S * 4 * H, pl).reshape(Bq, S, 4 * H).astype(f)
        g = act_fwd(act, u) * di
        d2 = R.keep_mask(seed, site(l, 3), Bq * S * H, pl).reshape(Bq, S, H).astype(f)
        r2 = x1 + (g @ P[q_ + "linear2.weight"].T + P[q_ + "linear2.bias"]) * d2
        x, lc["xhat2"], lc["rstd2"] = layer_norm(r2, P[q_ + "norm2.weight"], P[q_ + "norm2.bias"])
        lc.update(q=q, k=k, v=v, prob=prob, pm=pm, pd=pd, a=a, d1=d1, x1=x1, u=u, di=di, g=g, d2=d2, out=x)
        layers.append(lc)
    xo, xhat_o, rstd_o = layer_norm(x, P[pre + "norm.weight"], P[pre + "norm.bias"])
    return xo, dict(layers=layers, xhat_o=xhat_o, rstd_o=rstd_o, x_last=x, pre=pre, nh=nh, L=L, act=act, shape=(Bq, S, H))


def stack_bwd(P, sc, dxo, g, f):
    """gradient of stack_fwd: dxo = d/d(output) -> returns d/d(input x); parameter gradients into g."""
    pre, nh, L, act = sc["pre"], sc["nh"], sc["L"], sc["act"]
    Bq, S, H = sc["shape"]
    hd = H // nh
    scale = f(1.0 / math.sqrt(hd))
    dx, g[pre + "norm.weight"], g[pre + "norm.bias"] = layer_norm_bwd(dxo, sc["xhat_o"], sc["rstd_o"], P[pre + "norm.weight"])

    def merge(t):
        return t.transpose(0, 2, 1, 3).reshape(Bq * S, H)

    for l in range(L - 1, -1, -1):
        q_ = f"{pre}layers.{l}."
        lc = sc["layers"][l]
        dr2, g[q_ + "norm2.weight"], g[q_ + "norm2.bias"] = layer_norm_bwd(dx, lc["xhat2"], lc["rstd2"], P[q_ + "norm2.weight"])
        c2 = (dr2 * lc["d2"]).reshape(Bq * S, H)
        g[q_ + "linear2.weight"] = c2.T @ lc["g"].reshape(Bq * S, -1)
        g[q_ + "linear2.bias"] = c2.sum(0)
        du = (c2 @ P[q_ + "linear2.weight"]) * (lc["di"] * act_bwd(act, lc["u"])).reshape(Bq * S, -1)
        g[q_ + "linear1.weight"] = du.T @ lc["x1"].reshape(Bq * S, H)
        g[q_ + "linear1.bias"] = du.sum(0)
        dx1 = dr2 + (du @ P[q_ + "linear1.weight"]).reshape(Bq, S, H)
        dr1, g[q_ + "norm1.weight"], g[q_ + "norm1.bias"] = layer_norm_bwd(dx1, lc["xhat1"], lc["rstd1"], P[q_ + "norm1.weight"])
        c1 = (dr1 * lc["d1"]).reshape(Bq * S, H)
        g[q_ + "self_attn.out_proj.weight"] = c1.T @ lc["a"].reshape(Bq * S, H)
        g[q_ + "self_attn.out_proj.bias"] = c1.sum(0)
        da = (c1 @ P[q_ + "self_attn.out_proj.weight"]).reshape(Bq, S, nh, hd).transpose(0, 2, 1, 3)
        dv = lc["pd"].transpose(0, 1, 3, 2) @ da
        dp = (da @ lc["v"].transpose(0, 1, 3, 2)) * lc["pm"]
        ds = lc["prob"] * (dp - (dp * lc["prob"]).sum(-1, keepdims=True))
        dq = (ds @ lc["k"]) * scale
        dk = (ds.transpose(0, 1, 3, 2) @ lc["q"]) * scale
        dqkv = np.concatenate([merge(dq), merge(dk), merge(dv)], 1)     # (Bq*S, 3H)
        g[q_ + "self_attn.in_proj_weight"] = dqkv.T @ lc["x_in"].reshape(Bq * S, H)
        g[q_ + "self_attn.in_proj_bias"] = dqkv.sum(0)
        dx = dr1 + (dqkv @ P[q_ + "self_attn.in_proj_weight"]).reshape(Bq, S, H)
    return dx


def forward(cfg, p, batch, masked, mask, train=False, seed=0, dtype=np.float32):
    """iTransformer.forward, 'mlm' (itransformer.py:312-359) AFTER the maskers: `masked` = masked spikes, `mask` = OR of
    the maskers' masks (B,T,N), batch['spikes'] = the untouched targets. Returns (out, cache)."""
    f = dtype
    P = {k: np.asarray(v, f) for k, v in p.items()}
    targets = np.asarray(batch["spikes"], f)
    smask = np.asarray(batch["spikes_mask"], np.int64)
    xm = np.asarray(masked, f)
    B, T, N = xm.shape
    H, L, nh = cfg["hidden"], cfg["n_layers"], cfg["n_heads"]
    pe = cfg["embed_dropout"] if train else 0.0
    pl = cfg["dropout"] if train else 0.0
    c = {}
    if cfg["embedder_mode"] == "mlp":
        # --- embed MLP over (B, N, T) + LayerNorm (itransformer.py:110-119,187)
        xs = np.ascontiguousarray(xm.transpose(0, 2, 1)).reshape(B * N, T)
        u0 = xs @ P["encoder.embed.0.0.weight"].T + P["encoder.embed.0.0.bias"]
        m0 = R.keep_mask(seed, SITE_EMB_HID, B * N * H, pe).reshape(B * N, H).astype(f)
        h0 = act_fwd(cfg["act"], u0) * m0
        m1 = R.keep_mask(seed, SITE_EMB_OUT, B * N * H, pe).reshape(B * N, H).astype(f)
        t2 = (h0 @ P["encoder.embed.0.3.weight"].T + P["encoder.embed.0.3.bias"]) * m1
        tok, c["xhat_e"], c["rstd_e"] = layer_norm(t2, P["encoder.embed.1.weight"], P["encoder.embed.1.bias"])
        c.update(xs=xs, u0=u0, m0=m0, h0=h0, m1=m1)
    else:
        # --- UnivariateTransformer (itransformer.py:75-93): every (sample, channel) is a sequence of its T bins; token = MLP(1 -> h -> h) of the
        # bin's count + the position embedding of its timestamp; CLS in front; a post-norm encoder (dropout = embedder.dropout inside the layers
        # only: the module's own embed_dropout is never applied, :56 vs :75-93); the CLS output is the channel's embedding; then
        # embed_proj = Linear(h -> H) + LayerNorm (:121-124,187)
        h, eh, eL = cfg["emb_hidden"], cfg["emb_heads"], cfg["emb_layers"]
        ts = batch.get("spikes_timestamp")
        ts = np.broadcast_to(np.arange(T, dtype=np.int64), (B, T)) if ts is None else np.asarray(ts, np.int64)
        w0, b0 = P["encoder.embed.embed_spikes.0.weight"].reshape(h), P["encoder.embed.embed_spikes.0.bias"]
        xbnt = np.ascontiguousarray(xm.transpose(0, 2, 1))                                   # (B,N,T)
        ue = xbnt[..., None] * w0 + b0                                                       # (B,N,T,h)
        he = act_fwd(cfg["emb_act"], ue)
        te = he @ P["encoder.embed.embed_spikes.2.weight"].T + P["encoder.embed.embed_spikes.2.bias"]
        te = te + P["encoder.embed.embed_pos.weight"][ts][:, None, :, :]                     # (B,1,T,h) over channels
        seq = np.concatenate([np.broadcast_to(P["encoder.embed.cls_embed.weight"][None, None], (B, N, 1, h)), te], 2).reshape(B * N, T + 1, h)
        eo, esc = stack_fwd(P, "encoder.embed.transformer.", seq, eh, eL, cfg["emb_act"], pe, seed, site_emb_layer, f)
        ecls = eo[:, 0, :]                                                                   # (B*N, h)
        t2 = ecls @ P["encoder.embed_proj.0.weight"].T + P["encoder.embed_proj.0.bias"]
        tok, c["xhat_e"], c["rstd_e"] = layer_norm(t2, P["encoder.embed_proj.1.weight"], P["encoder.embed_proj.1.bias"])
        c.update(ts=ts, xbnt=xbnt, ue=ue, he=he, esc=esc, ecls=ecls, emb_seq=seq, emb_out=eo)
    tok = tok.reshape(B, N, H)
    embed_hook = tok.copy()   # what a forward hook on `encoder.embed_proj` (transformer mode) sees before the in-place adds below
    # --- channel embeddings (itransformer.py:192-196)
    ss = batch.get("spikes_spacestamp")
    if cfg["max_n_channels"]:
        ss = np.broadcast_to(np.arange(N, dtype=np.int64), (B, N)) if ss is None else np.asarray(ss, np.int64).reshape(B, N)
        ce, c["xhat_c"], c["rstd_c"] = layer_norm(P["encoder.channel_embeddings.0.weight"], P["encoder.channel_embeddings.1.weight"],
                                                 P["encoder.channel_embeddings.1.bias"])
        tok = tok + ce[ss]
        c["ss"] = ss
    # --- region embeddings (itransformer.py:133-141,195-198): LayerNorm'd table rows picked by the neuron's brain region
    if cfg["n_regions"]:
        ridx = np.asarray(batch["region_idx"], np.int64).reshape(B, N)
        re_, c["xhat_r"], c["rstd_r"] = layer_norm(P["encoder.region_embeddings.0.weight"], P["encoder.region_embeddings.1.weight"],
                                                  P["encoder.region_embeddings.1.bias"])
        tok = tok + re_[ridx]
        c["ridx"] = ridx
    # --- depth embeddings (itransformer.py:143-150,200-202): LayerNorm(Linear(act(Linear(depth)))) of the neuron's depth (a scalar)
    if cfg["embed_depth"]:
        dep = np.asarray(batch["neuron_depths"], f).reshape(B * N, 1)
        ud = dep * P["encoder.depth_embeddings.0.weight"].reshape(1, H) + P["encoder.depth_embeddings.0.bias"]
        hd_ = act_fwd(cfg["act"], ud)
        td = hd_ @ P["encoder.depth_embeddings.2.weight"].T + P["encoder.depth_embeddings.2.bias"]
        de, c["xhat_d"], c["rstd_d"] = layer_norm(td, P["encoder.depth_embeddings.3.weight"], P["encoder.depth_embeddings.3.bias"])
        tok = tok + de.reshape(B, N, H)
        c.update(dep=dep, ud_dep=ud, hd=hd_)
    embed_out = tok   # what a forward hook on `encoder.embed` (mlp mode) sees: the `tokens += ...` adds are in place (itransformer.py:193-202)
    # --- CLS + embed dropout (itransformer.py:206-209)
    if cfg["use_cls"]:
        tok = np.concatenate([np.broadcast_to(P["encoder.cls_embed.weight"][None], (B, 1, H)), tok], 1)
    S = tok.shape[1]
    m2 = R.keep_mask(seed, SITE_TOKENS, B * S * H, pe).reshape(B, S, H).astype(f)
    x = tok * m2
    tokens = x
    xo, sc = stack_fwd(P, "encoder.transformer.", x, nh, L, cfg["act"], pl, seed, site_layer, f)
    layers = sc["layers"]
    # --- decoder on the channel tokens (itransformer.py:264-279,333-339) + masked loss (:341-352)
    xd = xo[:, 1:, :] if cfg["use_cls"] else xo
    hn, on = _head_names(cfg)
    if hn:
        ud_ = xd @ P[hn + ".weight"].T + P[hn + ".bias"]
        dd = act_fwd(cfg["dec_act"], ud_)
    else:
        ud_, dd = None, xd
    raw = dd @ P[on + ".weight"].T + P[on + ".bias"]                 # (B,N,T)
    rate_relu = cfg["loss"] == "poisson_nll" and not cfg["log_input"]  # trailing nn.ReLU (itransformer.py:281-282)
    pr = np.maximum(raw, 0) if rate_relu else raw
    preds = pr.transpose(0, 2, 1)                                     # (B,T,N)
    tmask = (np.asarray(mask, np.int64) & smask[:, :, None]).astype(np.int64)
    if cfg["loss"] == "poisson_nll":
        if cfg["log_input"]:
            el = np.exp(preds) - targets * preds
            dl = np.exp(preds) - targets
        else:
            el = preds - targets * np.log(preds + f(1e-8))
            dl = 1 - targets / (preds + f(1e-8))
    elif cfg["loss"] == "mse":
        el = (preds - targets) ** 2
        dl = 2 * (preds - targets)
    else:
        raise Exception(f"Loss {cfg['loss']} not implemented yet for mlm")
    loss = (el * tmask).sum()
    dpred = (dl * tmask).astype(f)                                    # (B,T,N)
    if rate_relu:
        dpred = dpred * (raw.transpose(0, 2, 1) > 0)
    out = {"loss": f(loss), "n_examples": np.int64(tmask.sum()), "preds": preds.astype(f), "targets": targets, "mask": tmask,
           "embed": embed_out, "embed_proj": embed_hook, "tokens": tokens, "layer_out": [lc["out"] for lc in layers], "encoder_out": xo}
    if cfg["embedder_mode"] != "mlp":
        out["emb_layer_out"] = [lc["out"] for lc in c["esc"]["layers"]]
        out["emb_out"] = c["emb_out"]
    c.update(cfg=cfg, P=P, B=B, T=T, N=N, S=S, f=f, m2=m2, sc=sc, layers=layers, xo=xo, ud=ud_, dd=dd, dpred=dpred, x_last=sc["x_last"])
    return out, c


def backward(c, grad_scale=1.0):
    """d(sum-loss)/d(params); keys = state-dict names."""
    cfg, P, B, T, N, S, f = c["cfg"], c["P"], c["B"], c["T"], c["N"], c["S"], c["f"]
    H = cfg["hidden"]
    g = {}
    hn, on = _head_names(cfg)
    draw = (c["dpred"] * f(grad_scale)).transpose(0, 2, 1).reshape(B * N, T)
    dd = c["dd"].reshape(B * N, -1)
    g[on + ".weight"] = draw.T @ dd
    g[on + ".bias"] = draw.sum(0)
    dxd = draw @ P[on + ".weight"]
    if hn:
        dud = dxd * act_bwd(cfg["dec_act"], c["ud"].reshape(B * N, H))
        xo_tok = (c["xo"][:, 1:, :] if cfg["use_cls"] else c["xo"]).reshape(B * N, H)
        g[hn + ".weight"] = dud.T @ xo_tok
        g[hn + ".bias"] = dud.sum(0)
        dxd = dud @ P[hn + ".weight"]
    dxo = np.zeros((B, S, H), f)
    if cfg["use_cls"]:
        dxo[:, 1:, :] = dxd.reshape(B, N, H)
    else:
        dxo[:] = dxd.reshape(B, N, H)
    dx = stack_bwd(P, c["sc"], dxo, g, f)
    # --- embedding side
    dtok = dx * c["m2"]
    if cfg["use_cls"]:
        g["encoder.cls_embed.weight"] = dtok[:, 0, :].sum(0, keepdims=True)
        dtok = dtok[:, 1:, :]
    if cfg["max_n_channels"]:
        dce = np.zeros_like(P["encoder.channel_embeddings.0.weight"])
        np.add.at(dce, c["ss"].reshape(-1), dtok.reshape(-1, H))
        g["encoder.channel_embeddings.0.weight"], g["encoder.channel_embeddings.1.weight"], g["encoder.channel_embeddings.1.bias"] = \
            layer_norm_bwd(dce, c["xhat_c"], c["rstd_c"], P["encoder.channel_embeddings.1.weight"])
    if cfg["n_regions"]:
        dre = np.zeros_like(P["encoder.region_embeddings.0.weight"])
        np.add.at(dre, c["ridx"].reshape(-1), dtok.reshape(-1, H))
        g["encoder.region_embeddings.0.weight"], g["encoder.region_embeddings.1.weight"], g["encoder.region_embeddings.1.bias"] = \
            layer_norm_bwd(dre, c["xhat_r"], c["rstd_r"], P["encoder.region_embeddings.1.weight"])
    if cfg["embed_depth"]:
        dtd, g["encoder.depth_embeddings.3.weight"], g["encoder.depth_embeddings.3.bias"] = layer_norm_bwd(
            dtok.reshape(B * N, H), c["xhat_d"], c["rstd_d"], P["encoder.depth_embeddings.3.weight"])
        g["encoder.depth_embeddings.2.weight"] = dtd.T @ c["hd"]
        g["encoder.depth_embeddings.2.bias"] = dtd.sum(0)
        dud_ = (dtd @ P["encoder.depth_embeddings.2.weight"]) * act_bwd(cfg["act"], c["ud_dep"])
        g["encoder.depth_embeddings.0.weight"] = (dud_ * c["dep"]).sum(0).reshape(H, 1)
        g["encoder.depth_embeddings.0.bias"] = dud_.sum(0)
    if cfg["embedder_mode"] == "mlp":
        dt2, g["encoder.embed.1.weight"], g["encoder.embed.1.bias"] = layer_norm_bwd(dtok.reshape(B * N, H), c["xhat_e"], c["rstd_e"],
                                                                                     P["encoder.embed.1.weight"])
        ce = dt2 * c["m1"]
        g["encoder.embed.0.3.weight"] = ce.T @ c["h0"]
        g["encoder.embed.0.3.bias"] = ce.sum(0)
        du0 = (ce @ P["encoder.embed.0.3.weight"]) * c["m0"] * act_bwd(cfg["act"], c["u0"])
        g["encoder.embed.0.0.weight"] = du0.T @ c["xs"]
        g["encoder.embed.0.0.bias"] = du0.sum(0)
    else:
        h = cfg["emb_hidden"]
        dt2, g["encoder.embed_proj.1.weight"], g["encoder.embed_proj.1.bias"] = layer_norm_bwd(dtok.reshape(B * N, H), c["xhat_e"], c["rstd_e"],
                                                                                               P["encoder.embed_proj.1.weight"])
        g["encoder.embed_proj.0.weight"] = dt2.T @ c["ecls"]
        g["encoder.embed_proj.0.bias"] = dt2.sum(0)
        deo = np.zeros((B * N, T + 1, h), f)
        deo[:, 0, :] = dt2 @ P["encoder.embed_proj.0.weight"]
        dseq = stack_bwd(P, c["esc"], deo, g, f).reshape(B, N, T + 1, h)
        g["encoder.embed.cls_embed.weight"] = dseq[:, :, 0, :].sum((0, 1)).reshape(1, h)
        dte = dseq[:, :, 1:, :]                                                                # (B,N,T,h)
        dpos = np.zeros_like(P["encoder.embed.embed_pos.weight"])
        np.add.at(dpos, c["ts"].reshape(-1), dte.sum(1).reshape(-1, h))
        g["encoder.embed.embed_pos.weight"] = dpos
        dte2 = dte.reshape(-1, h)
        g["encoder.embed.embed_spikes.2.weight"] = dte2.T @ c["he"].reshape(-1, h)
        g["encoder.embed.embed_spikes.2.bias"] = dte2.sum(0)
        due = (dte2 @ P["encoder.embed.embed_spikes.2.weight"]) * act_bwd(cfg["emb_act"], c["ue"].reshape(-1, h))
        g["encoder.embed.embed_spikes.0.weight"] = (due * c["xbnt"].reshape(-1, 1)).sum(0).reshape(h, 1)
        g["encoder.embed.embed_spikes.0.bias"] = due.sum(0)
    return {k: v.astype(f) for k, v in g.items()}
