"""numpy restatement of the reference's NDT1 encoder + CTC head, forward AND hand-derived
backward (models/ndt1.py). Test infrastructure only — see oracle/__init__.py.

Parameter names are the reference's state-dict keys ("encoder.embedder.embed_spikes.weight",
..., "decoder.0.weight"). All math is float32 unless `dtype=np.float64` is
requested (used to measure the f32 rounding floor in tests).
"""
import math

import numpy as np

from . import rng as R
from .ctc import ctc_loss_and_grad

try:  # erf for exact GELU (transformers ACT2FN["gelu"] = erf form; ndt1.py:220, ndt1.yaml:65)
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)


# ----------------------------------------------------------------------------- config
DEFAULTS = dict(  # configs/ndt1.yaml defaults, flattened
    n_channels=256, input_dim=256, stack_size=32, stack_stride=4, hidden=1024, n_layers=5,
    n_heads=8, inter=1024, vocab=41, max_F=1024, smooth_sd=2, noise=True, white_noise_sd=1.0,
    constant_offset_sd=0.2, embed_act="softsign", mlp_act="gelu", embed_dropout=0.2, dropout=0.4,
    use_rope=False, rope_theta=10000.0, context_forward=-2, context_backward=-2, pos=True,
    blank_id=0, zero_infinity=True,
    factors_size=0, factors_act="relu", factors_bias=True,   # NeuralFactorsProjection (ndt1.py:348-373); size 0 = inactive (identity)
    day_token_days=0, block_token_blocks=0,   # embedder.day_token / block_token: learned prefix tokens, table sizes (ndt1.py:151-155,192-201); 0 = off
    adapt_days=0,   # embedder.adapt: one embed_spikes Linear per recording day, picked by batch["day_idx"] (ndt1.py:124-129,170-171); 0 = shared
)


def make_config(**kw):
    cfg = dict(DEFAULTS)
    cfg.update(kw)
    return cfg


# ----------------------------------------------------------------------------- pieces
def gaussian_taps(sd):
    """ndt1.py:87-88: scipy.signal.gaussian(1 + 6*sd, sd) normalised, built in float64."""
    n = 1 + 6 * sd
    k = np.arange(n, dtype=np.float64) - (n - 1) / 2.0
    w = np.exp(-0.5 * (k / sd) ** 2)
    return w / w.sum()


def smooth(spikes, taps):
    """ndt1.py:96-97: depthwise 'same' zero-padded correlation along time, per channel."""
    B, T, N = spikes.shape
    K = len(taps)
    half = (K - 1) // 2
    pad = np.zeros((B, T + K - 1, N), spikes.dtype)
    pad[:, half:half + T] = spikes
    out = np.zeros_like(spikes)
    tw = taps.astype(spikes.dtype)
    for i in range(K):
        out += tw[i] * pad[:, i:i + T]
    return out


def stacked_len(lens, size, stride):
    """ndt1.py:207-208: (1 + (len - size)/stride) as float division, truncating cast."""
    return (1 + (np.asarray(lens, np.float64) - size) / stride).astype(np.int64)


def context_mask(fwd, bwd, max_F):
    """ndt1.py:30-41 (create_context_mask)."""
    if fwd == -2 and bwd == -2:
        return np.ones((max_F, max_F), np.int64)
    f = fwd if fwd >= -1 else max_F
    b = bwd if bwd >= -1 else max_F
    ones = np.ones((max_F, max_F), np.int64)
    mask = np.triu(ones, k=-f).T
    if b >= -1:
        mask = mask & np.triu(ones, k=-b)
    return mask


def rope_tables(dim, max_F, base):
    """ndt1.py:46-53."""
    inv = 1.0 / (base ** (np.arange(0, dim, 2, dtype=np.float32) / dim))
    t = np.arange(max_F, dtype=np.float32)
    fr = np.einsum("i,j->ij", t, inv)
    emb = np.concatenate([fr, fr], -1)
    return np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)


def rotate_half(x):
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], -1)


def rotate_half_T(g):
    """transpose of rotate_half (for the backward)."""
    h = g.shape[-1] // 2
    return np.concatenate([g[..., h:], -g[..., :h]], -1)


def act_fwd(name, x):
    if name == "softsign":
        return x / (1 + np.abs(x))
    if name == "gelu":
        return (0.5 * x * (1 + _erf(x / np.sqrt(2.0)))).astype(x.dtype)
    if name == "relu":
        return np.maximum(x, 0)
    if name == "tanh":
        return np.tanh(x)
    if name in ("identity", None):
        return x
    raise ValueError(name)


def act_bwd(name, x):
    if name == "softsign":
        return 1 / (1 + np.abs(x)) ** 2
    if name == "gelu":
        cdf = 0.5 * (1 + _erf(x / np.sqrt(2.0)))
        pdf = np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)
        return (cdf + x * pdf).astype(x.dtype)
    if name == "relu":
        return (x > 0).astype(x.dtype)
    if name == "tanh":
        return (1 - np.tanh(x) ** 2).astype(x.dtype)
    if name in ("identity", None):
        return np.ones_like(x)
    raise ValueError(name)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1 / np.sqrt(var + eps)
    xhat = (x - mu) * rstd
    return xhat * w + b, xhat, rstd


def layer_norm_bwd(dy, xhat, rstd, w):
    dxhat = dy * w
    dx = rstd * (dxhat - dxhat.mean(-1, keepdims=True) - xhat * (dxhat * xhat).mean(-1, keepdims=True))
    return dx, (dy * xhat).reshape(-1, dy.shape[-1]).sum(0), dy.reshape(-1, dy.shape[-1]).sum(0)


def init_params(cfg, seed=0, dtype=np.float32):
    """Random parameters with the reference's shapes and init scales (ndt1.py:332-344 fixup,
    nn.Linear U(+-1/sqrt(fan_in)), nn.Embedding N(0,1)). NOT bit-equal to torch's init; the
    golden tests load the reference's own weights instead."""
    g = np.random.default_rng(seed)
    N, D, S, H, L, I, V = (cfg[k] for k in ("n_channels", "input_dim", "stack_size", "hidden", "n_layers", "inter", "vocab"))

    def lin(o, i):
        b = 1 / math.sqrt(i)
        return g.uniform(-b, b, (o, i)).astype(dtype), g.uniform(-b, b, (o,)).astype(dtype)

    p = {}
    if cfg.get("adapt_days", 0):
        for d in range(cfg["adapt_days"]):
            p[f"encoder.embedder.embed_spikes.{d}.weight"], p[f"encoder.embedder.embed_spikes.{d}.bias"] = lin(D, N)
    else:
        p["encoder.embedder.embed_spikes.weight"], p["encoder.embedder.embed_spikes.bias"] = lin(D, N)
    p["encoder.embedder.stack_projection.weight"], p["encoder.embedder.stack_projection.bias"] = lin(H, D * S)
    p["encoder.embedder.embed_pos.weight"] = g.standard_normal((cfg["max_F"], H)).astype(dtype)
    if cfg.get("block_token_blocks", 0):
        p["encoder.embedder.block_embedding.weight"] = g.standard_normal((cfg["block_token_blocks"], H)).astype(dtype)
    if cfg.get("day_token_days", 0):
        p["encoder.embedder.day_embedding.weight"] = g.standard_normal((cfg["day_token_days"], H)).astype(dtype)
    fix = 0.67 * L ** (-0.25)
    for l in range(L):
        pre = f"encoder.layers.{l}."
        p[pre + "ln1.weight"] = np.ones(H, dtype); p[pre + "ln1.bias"] = np.zeros(H, dtype)
        for nm in ("query", "key", "value", "out_proj"):
            w, b = lin(H, H)
            if nm == "value":
                w = w * dtype(fix * math.sqrt(2))
            if nm == "out_proj":
                w = w * dtype(fix)
            p[pre + f"attn.{nm}.weight"], p[pre + f"attn.{nm}.bias"] = w.astype(dtype), b
        p[pre + "ln2.weight"] = np.ones(H, dtype); p[pre + "ln2.bias"] = np.zeros(H, dtype)
        w, b = lin(I, H); p[pre + "mlp.up_proj.weight"], p[pre + "mlp.up_proj.bias"] = (w * dtype(fix)).astype(dtype), b
        w, b = lin(H, I); p[pre + "mlp.down_proj.weight"], p[pre + "mlp.down_proj.bias"] = (w * dtype(fix)).astype(dtype), b
    p["encoder.out_norm.weight"] = np.ones(H, dtype); p["encoder.out_norm.bias"] = np.zeros(H, dtype)
    Fs = cfg.get("factors_size", 0)
    if Fs:
        w, b = lin(Fs, H)
        p["encoder.out_proj.proj.0.weight"] = w
        if cfg.get("factors_bias", True):
            p["encoder.out_proj.proj.0.bias"] = b
    p["decoder.0.weight"], p["decoder.0.bias"] = lin(V, Fs if Fs else H)
    return p


# ----------------------------------------------------------------------------- forward
def forward(cfg, p, batch, train=False, seed=0, dtype=np.float32, keep_cache=True):
    """NDT1.forward, method 'ctc' (ndt1.py:523-545,580-589) through NeuralEncoder.forward
    (:408-450). batch: spikes (B,T,N) f32, spikes_mask (B,T) i64, spikes_timestamp (B,T) i64,
    spikes_lengths (B,), targets (B,S) i64, targets_lengths (B,).
    Returns dict(loss, n_examples, preds=(B,T',V) log-probs, ...) and the cache for backward()."""
    f = dtype
    spikes = np.asarray(batch["spikes"], f)
    smask = np.asarray(batch["spikes_mask"], np.int64)
    ts = np.asarray(batch["spikes_timestamp"], np.int64)
    lens = np.asarray(batch["spikes_lengths"], np.int64).reshape(-1)
    B, T, N = spikes.shape
    D, S, st, H, L, nh, V = (cfg[k] for k in ("input_dim", "stack_size", "stack_stride", "hidden", "n_layers", "n_heads", "vocab"))
    hd = H // nh
    P = {k: np.asarray(v, f) for k, v in p.items()}
    c = {}

    # 1-2. smoothing + train-time noise (ndt1.py:92-107)
    xs = smooth(spikes, gaussian_taps(cfg["smooth_sd"])) if cfg["smooth_sd"] is not None else spikes.copy()
    if train and cfg["noise"]:
        if cfg["white_noise_sd"] is not None:
            xs = xs + f(cfg["white_noise_sd"]) * R.white_noise(seed, B, T, N).astype(f)
        if cfg["constant_offset_sd"] is not None:
            xs = xs + f(cfg["constant_offset_sd"]) * R.normal(seed, R.SITE_NOISE_OFFSET, B * N).reshape(B, 1, N).astype(f)
    c["xs"] = xs
    # 4. embed + activation (ndt1.py:173-176)
    if cfg.get("adapt_days", 0):   # per-sample day-specific layer (ndt1.py:170-171)
        days = np.asarray(batch["day_idx"], np.int64).reshape(-1)
        pre = np.stack([xs[b] @ P[f"encoder.embedder.embed_spikes.{days[b]}.weight"].T + P[f"encoder.embedder.embed_spikes.{days[b]}.bias"]
                        for b in range(B)], 0)
        c["days"] = days
    else:
        pre = xs @ P["encoder.embedder.embed_spikes.weight"].T + P["encoder.embedder.embed_spikes.bias"]
    y = act_fwd(cfg["embed_act"], pre)
    c["y"] = y
    # 5. stack (nn.Unfold row-major flatten) + projection (ndt1.py:138-140,180)
    Tp = 1 + (T - S) // st
    win = np.stack([y[:, j * st:j * st + S, :].reshape(B, S * D) for j in range(Tp)], 1)  # (B,Tp,S*D)
    x = win @ P["encoder.embedder.stack_projection.weight"].T + P["encoder.embedder.stack_projection.bias"]
    c["win"] = win
    # 6. token mask / timestamps / lengths (ndt1.py:181-183,207-208)
    tmask = np.stack([smask[:, j * st:j * st + S].prod(-1) for j in range(Tp)], 1)
    tts = ts[:, :Tp]
    tlens = stacked_len(lens, S, st)
    # 7. position + dropout (ndt1.py:188-189,203)
    if cfg["pos"]:
        x = x + P["encoder.embedder.embed_pos.weight"][tts]
    # block token, then day token, prepended (ndt1.py:192-201): the sequence becomes [day, block, tokens...]; their mask entries are 1
    pref = []
    if cfg.get("day_token_days", 0):
        c["day_sel"] = np.asarray(batch["day_idx"], np.int64).reshape(-1)
        pref.append(P["encoder.embedder.day_embedding.weight"][c["day_sel"]])
    if cfg.get("block_token_blocks", 0):
        c["block_sel"] = np.asarray(batch["block_idx"], np.int64).reshape(-1)
        pref.append(P["encoder.embedder.block_embedding.weight"][c["block_sel"]])
    npre = len(pref)
    if npre:
        assert not cfg["use_rope"], "rope + prefix tokens: the reference passes T' timestamps for T'+n tokens (ndt1.py:181,441)"
        x = np.concatenate([t[:, None, :] for t in pref] + [x], 1)
        tmask = np.concatenate([np.ones((B, npre), tmask.dtype), tmask], 1)
    Tq = Tp + npre   # sequence length inside the transformer
    ed = R.keep_mask(seed, R.SITE_EMBED_DROP, B * Tq * H, cfg["embed_dropout"] if train else 0.0).reshape(B, Tq, H).astype(f)
    x = x * ed
    c["tts"] = tts
    # 8. attention mask (ndt1.py:435-437): eye | (ctx & key_valid)
    ctx = context_mask(cfg["context_forward"], cfg["context_backward"], cfg["max_F"])[:Tq, :Tq]
    amask = (np.eye(Tq, dtype=np.int64)[None] | (ctx[None] & tmask[:, None, :])).astype(bool)  # (B,Tq,Tq)
    c["amask"] = amask
    if cfg["use_rope"]:
        cos_t, sin_t = rope_tables(hd, cfg["max_F"], cfg["rope_theta"])
        cos, sin = cos_t[tts][:, None].astype(f), sin_t[tts][:, None].astype(f)  # (B,1,Tq,hd)
        c["cos"], c["sin"] = cos, sin
    pl = cfg["dropout"] if train else 0.0
    scale = f(1.0 / math.sqrt(hd))
    layers = []
    for l in range(L):
        pre_ = f"encoder.layers.{l}."
        lc = {"x_in": x}
        h1, lc["xhat1"], lc["rstd1"] = layer_norm(x, P[pre_ + "ln1.weight"], P[pre_ + "ln1.bias"])
        lc["h1"] = h1

        def heads(t):
            return t.reshape(B, Tq, nh, hd).transpose(0, 2, 1, 3)

        q = heads(h1 @ P[pre_ + "attn.query.weight"].T + P[pre_ + "attn.query.bias"])
        k = heads(h1 @ P[pre_ + "attn.key.weight"].T + P[pre_ + "attn.key.bias"])
        v = heads(h1 @ P[pre_ + "attn.value.weight"].T + P[pre_ + "attn.value.bias"])
        if cfg["use_rope"]:
            q = q * cos + rotate_half(q) * sin
            k = k * cos + rotate_half(k) * sin
        s = (q @ k.transpose(0, 1, 3, 2)) * scale
        s = np.where(amask[:, None], s, f(-np.inf))
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        prob = e / e.sum(-1, keepdims=True)
        pm = R.keep_mask(seed, R.site_attn_prob(l), B * nh * Tq * Tq, pl).reshape(B, nh, Tq, Tq).astype(f)
        pd = prob * pm
        a = (pd @ v).transpose(0, 2, 1, 3).reshape(B, Tq, H)
        am = R.keep_mask(seed, R.site_attn_out(l), B * Tq * H, pl).reshape(B, Tq, H).astype(f)
        ad = a * am
        x = x + ad @ P[pre_ + "attn.out_proj.weight"].T + P[pre_ + "attn.out_proj.bias"]
        lc.update(q=q, k=k, v=v, prob=prob, pm=pm, pd=pd, am=am, ad=ad, x_mid=x)
        h2, lc["xhat2"], lc["rstd2"] = layer_norm(x, P[pre_ + "ln2.weight"], P[pre_ + "ln2.bias"])
        u = h2 @ P[pre_ + "mlp.up_proj.weight"].T + P[pre_ + "mlp.up_proj.bias"]
        gact = act_fwd(cfg["mlp_act"], u)
        m = gact @ P[pre_ + "mlp.down_proj.weight"].T + P[pre_ + "mlp.down_proj.bias"]
        mm = R.keep_mask(seed, R.site_mlp_out(l), B * Tq * H, pl).reshape(B, Tq, H).astype(f)
        x = x + m * mm
        lc.update(h2=h2, u=u, g=gact, mm=mm)
        layers.append(lc)
    c["layers"] = layers
    c["x_last"] = x
    xo, c["xhat_o"], c["rstd_o"] = layer_norm(x, P["encoder.out_norm.weight"], P["encoder.out_norm.bias"])
    xo_full = xo
    xo = xo[:, npre:]              # prefix tokens dropped after out_norm, before out_proj / decoder (ndt1.py:444-450)
    c["xo"] = xo
    enc_out = xo
    if cfg.get("factors_size", 0):   # out_proj = act(Linear(dropout_{p=0}(x))) (ndt1.py:362-365,372-373)
        fu = xo @ P["encoder.out_proj.proj.0.weight"].T
        if "encoder.out_proj.proj.0.bias" in P:
            fu = fu + P["encoder.out_proj.proj.0.bias"]
        enc_out = act_fwd(cfg["factors_act"], fu)
        c["fu"], c["fo"] = fu, enc_out
    logits = enc_out @ P["decoder.0.weight"].T + P["decoder.0.bias"]
    z = logits - logits.max(-1, keepdims=True)
    lp = z - np.log(np.exp(z).sum(-1, keepdims=True))
    out = {"preds": lp.astype(f), "logits": logits, "token_mask": tmask, "token_lens": tlens, "x_embed": layers[0]["x_in"] if L else x,
           "layer_out": [lc["x_mid"] for lc in layers], "x_final": xo_full, "enc_out": enc_out, "xs": xs, "y": y}
    if "targets" in batch and batch["targets"] is not None:
        losses, dlogits = ctc_loss_and_grad(lp, batch["targets"], tlens, np.asarray(batch["targets_lengths"]).reshape(-1),
                                            blank=cfg["blank_id"], zero_infinity=cfg["zero_infinity"])
        out["loss_per_sample"] = losses
        out["loss"] = f(losses.sum())
        out["n_examples"] = np.int64(B)
        c["dlogits"] = dlogits.astype(f)
    c.update(cfg=cfg, P=P, B=B, T=T, Tp=Tp, Tq=Tq, npre=npre, train=train, seed=seed, ed=ed, f=f, pre_embed=pre)
    return out, (c if keep_cache else None)


# ----------------------------------------------------------------------------- backward
def backward(c, grad_scale=1.0, d_enc_out=None):
    """d(sum-loss)/d(params), hand-derived; keys = state-dict names. grad_scale multiplies the
    loss (trainer.py:339 divides by gradient_accumulation_steps).
    d_enc_out (B,T',H or factors size): start from a gradient of the ENCODER OUTPUT instead of the CTC head (the encoder as BCI's
    feature extractor, models/bci.py:125); the decoder then gets no gradient."""
    cfg, P, B, T, Tp, f = c["cfg"], c["P"], c["B"], c["T"], c["Tp"], c["f"]
    D, S, st, H, L, nh = (cfg[k] for k in ("input_dim", "stack_size", "stack_stride", "hidden", "n_layers", "n_heads"))
    hd = H // nh
    scale = f(1.0 / math.sqrt(hd))
    g = {}
    Fs = cfg.get("factors_size", 0)
    if d_enc_out is not None:
        dxo = np.asarray(d_enc_out, f) * f(grad_scale)
        g["decoder.0.weight"] = np.zeros_like(P["decoder.0.weight"])
        g["decoder.0.bias"] = np.zeros_like(P["decoder.0.bias"])
    else:
        dlogits = c["dlogits"] * f(grad_scale)                          # (B,Tp,V)
        dec_in = c["fo"] if Fs else c["xo"]
        g["decoder.0.weight"] = dlogits.reshape(-1, dlogits.shape[-1]).T @ dec_in.reshape(-1, dec_in.shape[-1])
        g["decoder.0.bias"] = dlogits.reshape(-1, dlogits.shape[-1]).sum(0)
        dxo = dlogits @ P["decoder.0.weight"]
    if Fs:
        dfu = dxo * act_bwd(cfg["factors_act"], c["fu"])
        g["encoder.out_proj.proj.0.weight"] = dfu.reshape(-1, Fs).T @ c["xo"].reshape(-1, H)
        if "encoder.out_proj.proj.0.bias" in P:
            g["encoder.out_proj.proj.0.bias"] = dfu.reshape(-1, Fs).sum(0)
        dxo = dfu @ P["encoder.out_proj.proj.0.weight"]
    Tq, npre = c["Tq"], c["npre"]
    if npre:   # the stripped prefix positions get no gradient from the head
        dxo = np.concatenate([np.zeros((B, npre, H), f), dxo], 1)
    dx, g["encoder.out_norm.weight"], g["encoder.out_norm.bias"] = layer_norm_bwd(dxo, c["xhat_o"], c["rstd_o"], P["encoder.out_norm.weight"])
    for l in range(L - 1, -1, -1):
        pre_ = f"encoder.layers.{l}."
        lc = c["layers"][l]
        # MLP: x = x_mid + dropout(down(gelu(up(ln2(x_mid)))))      (ndt1.py:224-227,328)
        dm = dx * lc["mm"]
        g[pre_ + "mlp.down_proj.weight"] = dm.reshape(-1, H).T @ lc["g"].reshape(-1, lc["g"].shape[-1])
        g[pre_ + "mlp.down_proj.bias"] = dm.reshape(-1, H).sum(0)
        dg = dm @ P[pre_ + "mlp.down_proj.weight"]
        du = dg * act_bwd(cfg["mlp_act"], lc["u"])
        g[pre_ + "mlp.up_proj.weight"] = du.reshape(-1, du.shape[-1]).T @ lc["h2"].reshape(-1, H)
        g[pre_ + "mlp.up_proj.bias"] = du.reshape(-1, du.shape[-1]).sum(0)
        dh2 = du @ P[pre_ + "mlp.up_proj.weight"]
        d2, g[pre_ + "ln2.weight"], g[pre_ + "ln2.bias"] = layer_norm_bwd(dh2, lc["xhat2"], lc["rstd2"], P[pre_ + "ln2.weight"])
        dx = dx + d2
        # attention: x_mid = x_in + out_proj(dropout(merge(dropout(softmax(qk^T)) v)))   (:266-292,325)
        g[pre_ + "attn.out_proj.weight"] = dx.reshape(-1, H).T @ lc["ad"].reshape(-1, H)
        g[pre_ + "attn.out_proj.bias"] = dx.reshape(-1, H).sum(0)
        da = (dx @ P[pre_ + "attn.out_proj.weight"]) * lc["am"]
        da = da.reshape(B, Tq, nh, hd).transpose(0, 2, 1, 3)         # (B,nh,Tq,hd)
        dv = lc["pd"].transpose(0, 1, 3, 2) @ da
        dpd = da @ lc["v"].transpose(0, 1, 3, 2)
        dp = dpd * lc["pm"]
        ds = lc["prob"] * (dp - (dp * lc["prob"]).sum(-1, keepdims=True))
        dq = (ds @ lc["k"]) * scale
        dk = (ds.transpose(0, 1, 3, 2) @ lc["q"]) * scale
        if cfg["use_rope"]:
            cos, sin = c["cos"], c["sin"]
            dq = dq * cos + rotate_half_T(dq * sin)
            dk = dk * cos + rotate_half_T(dk * sin)

        def merge(t):
            return t.transpose(0, 2, 1, 3).reshape(B * Tq, H)

        dq, dk, dv = merge(dq), merge(dk), merge(dv)
        h1 = lc["h1"].reshape(-1, H)
        dh1 = np.zeros((B * Tq, H), f)
        for nm, dd in (("query", dq), ("key", dk), ("value", dv)):
            g[pre_ + f"attn.{nm}.weight"] = dd.T @ h1
            g[pre_ + f"attn.{nm}.bias"] = dd.sum(0)
            dh1 = dh1 + dd @ P[pre_ + f"attn.{nm}.weight"]
        d1, g[pre_ + "ln1.weight"], g[pre_ + "ln1.bias"] = layer_norm_bwd(dh1.reshape(B, Tq, H), lc["xhat1"], lc["rstd1"], P[pre_ + "ln1.weight"])
        dx = dx + d1
    # embedder (ndt1.py:160-203)
    dx0 = dx * c["ed"]
    if npre:   # prefix token tables: scatter-add of their (dropout-masked) gradient rows; then continue with the spike tokens
        k = 0
        if cfg.get("day_token_days", 0):
            gd_ = np.zeros_like(P["encoder.embedder.day_embedding.weight"]); np.add.at(gd_, c["day_sel"], dx0[:, k]); k += 1
            g["encoder.embedder.day_embedding.weight"] = gd_
        if cfg.get("block_token_blocks", 0):
            gb_ = np.zeros_like(P["encoder.embedder.block_embedding.weight"]); np.add.at(gb_, c["block_sel"], dx0[:, k])
            g["encoder.embedder.block_embedding.weight"] = gb_
        dx0 = dx0[:, npre:]
    gpos = np.zeros_like(P["encoder.embedder.embed_pos.weight"])
    if cfg["pos"]:
        np.add.at(gpos, c["tts"].reshape(-1), dx0.reshape(-1, H))
    g["encoder.embedder.embed_pos.weight"] = gpos
    g["encoder.embedder.stack_projection.weight"] = dx0.reshape(-1, H).T @ c["win"].reshape(-1, S * D)
    g["encoder.embedder.stack_projection.bias"] = dx0.reshape(-1, H).sum(0)
    dwin = dx0 @ P["encoder.embedder.stack_projection.weight"]      # (B,Tp,S*D)
    dy = np.zeros((B, T, D), f)
    for j in range(Tp):
        dy[:, j * st:j * st + S, :] += dwin[:, j].reshape(B, S, D)
    dpre = dy * act_bwd(cfg["embed_act"], c["pre_embed"])
    if cfg.get("adapt_days", 0):
        for d in range(cfg["adapt_days"]):
            sel = [b for b in range(B) if c["days"][b] == d]
            g[f"encoder.embedder.embed_spikes.{d}.weight"] = sum((dpre[b].T @ c["xs"][b] for b in sel), np.zeros((D, c["xs"].shape[-1]), f))
            g[f"encoder.embedder.embed_spikes.{d}.bias"] = sum((dpre[b].sum(0) for b in sel), np.zeros(D, f))
    else:
        g["encoder.embedder.embed_spikes.weight"] = dpre.reshape(-1, D).T @ c["xs"].reshape(-1, c["xs"].shape[-1])
        g["encoder.embedder.embed_spikes.bias"] = dpre.reshape(-1, D).sum(0)
    return {k: v.astype(f) for k, v in g.items()}
