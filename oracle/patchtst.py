"""numpy restatement of the reference's PatchTST path: PatchTSTForSpikingActivity.forward (models/patchtst.py:214-255),
PredictHead (:70-94), PretrainHead (:139-154) over HF transformers' PatchTSTModel (patchtst.py:8,176), forward AND
hand-derived backward. Test infrastructure only — see oracle/__init__.py.

The encoder is third-party arithmetic (transformers, version unpinned by the reference; restated from
transformers 5.15 `models/patchtst/modeling_patchtst.py` and pinned by tests/golden/g_ptst_*.npz):
  * NOP scaler; patchify: x[:, start:, :].unfold(-2, patch_length, stride) -> (B,C,P,pl), start = T - (pl + stride*(P-1));
  * random masking (when do_mask_input; ALSO in eval): per (b,c) row exactly P - int(P*(1-ratio)) patches, those with the
    largest noise, are set to mask_value;
  * shared Linear(pl -> D) + fixed sincos position_enc (P,D); positional dropout;
  * L pre-norm layers over rows (b,c,p):  h += path_drop(out_proj(MHA(BN1(h))));  h += path_drop(ff(BN3(h)))
    BN = nn.BatchNorm1d(D) over ALL rows (batch statistics in train mode, running statistics in eval, momentum 0.1,
    unbiased variance into running_var), MHA = separate q/k/v/out Linear, softmax(q k^T / sqrt(hd)) + dropout per (b,c),
    ff = Linear(D,F), GELU(erf), Dropout, Linear(F,D).
Parameter / buffer names are the reference's state-dict keys.
"""
import math

import numpy as np

from . import rng as R
from .ctc import ctc_loss_and_grad
from .ndt1 import act_bwd, act_fwd

SITE_POS_DROP, SITE_HEAD_DROP, SITE_MASK = 4, 5, 6


def site_layer(l, k):
    """k: 0 attention probabilities, 1 path dropout 1, 2 ff dropout, 3 path dropout 3."""
    return 16 + 4 * l + k


DEFAULTS = dict(num_input_channels=128, context_length=45, patch_length=10, patch_stride=10, num_hidden_layers=4, d_model=256,
                num_attention_heads=8, ffn_dim=1024, norm_eps=1e-5, attention_dropout=0.4, positional_dropout=0.0, path_dropout=0.0,
                ff_dropout=0.4, activation_function="gelu", do_mask_input=True, random_mask_ratio=0.1, channel_consistent_masking=False,
                mask_value=0.0, head_dropout=0.0, mlp_decoder=False, mlp_activation="gelu", method="ctc", vocab=41, blank_id=0,
                zero_infinity=True, log_input=True, loss="poisson_nll")


def make_config(**kw):
    c = dict(DEFAULTS)
    c.update(kw)
    return c


def num_patches(cfg):
    T, pl, st = cfg["context_length"], cfg["patch_length"], cfg["patch_stride"]
    P = (max(T, pl) - pl) // st + 1
    return P, T - (pl + st * (P - 1))


def position_enc(P, D):
    """PatchTSTPositionalEncoding._init_pe, 'sincos' (float32 arithmetic as torch does it)."""
    pe = np.zeros((P, D), np.float32)
    pos = np.arange(P, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, D, 2).astype(np.float32) * np.float32(-(math.log(10000.0) / D)))
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    pe = pe - pe.mean()
    return (pe / (pe.std(ddof=1) * 10)).astype(np.float32)


def patchify(x, cfg):
    P, start = num_patches(cfg)
    pl, st = cfg["patch_length"], cfg["patch_stride"]
    return np.stack([x[:, start + p * st:start + p * st + pl, :] for p in range(P)], 1).transpose(0, 3, 1, 2)  # (B,C,P,pl)


def random_mask(cfg, B, C, P, seed):
    """random_masking with the counter RNG: noise per (b,c,p) (per (b,p) when channel-consistent); the P - len_keep
    patches with the largest noise (ties: larger index) are masked."""
    keep = int(P * (1 - cfg["random_mask_ratio"]))
    if cfg["channel_consistent_masking"]:
        noise = np.repeat((R.rng_u32(seed, SITE_MASK, np.arange(B * P, dtype=np.uint32))).reshape(B, 1, P), C, 1)
    else:
        noise = R.rng_u32(seed, SITE_MASK, np.arange(B * C * P, dtype=np.uint32)).reshape(B, C, P)
    rank = np.argsort(np.argsort(noise, axis=-1, kind="stable"), axis=-1, kind="stable")
    return rank >= keep


def bn_fwd(x, w, b, rm, rv, train, eps):
    """BatchNorm1d over rows. Returns y, cache, (new running mean, var)."""
    M = x.shape[0]
    if train:
        mu = x.mean(0)
        var = ((x - mu) ** 2).mean(0)
        nrm, nrv = 0.9 * rm + 0.1 * mu, 0.9 * rv + 0.1 * var * (M / max(M - 1, 1))
    else:
        mu, var, nrm, nrv = rm, rv, rm, rv
    rstd = 1 / np.sqrt(var + eps)
    xhat = (x - mu) * rstd
    return (xhat * w + b).astype(x.dtype), (xhat, rstd, train), (nrm.astype(x.dtype), nrv.astype(x.dtype))


def bn_bwd(dy, cache, w):
    xhat, rstd, train = cache
    dw, db = (dy * xhat).sum(0), dy.sum(0)
    if train:
        M = dy.shape[0]
        dx = (w * rstd / M) * (M * dy - db - xhat * dw)
    else:
        dx = dy * w * rstd
    return dx.astype(dy.dtype), dw, db


def _patch_valid(smask, cfg):
    """patchtst.py:228: spikes_mask.unfold(-1, pl, stride).prod(-1) — windows start at 0, NOT at the patchifier's start."""
    P, _ = num_patches(cfg)
    pl, st = cfg["patch_length"], cfg["patch_stride"]
    return np.stack([smask[:, p * st:p * st + pl].prod(-1) for p in range(P)], 1).astype(bool)


def forward(cfg, p, bufs, batch, mask=None, train=False, seed=0, dtype=np.float32):
    """Returns (out, cache, new_bufs). `mask`: (B,C,P) bool replaces the random draw (fixtures replay the reference's)."""
    f = dtype
    P_ = {k: np.asarray(v, f) for k, v in p.items()}
    x = np.asarray(batch["spikes"], f)
    B, T, C = x.shape
    D, L, nh, F = cfg["d_model"], cfg["num_hidden_layers"], cfg["num_attention_heads"], cfg["ffn_dim"]
    hd = D // nh
    P, _ = num_patches(cfg)
    pl = cfg["patch_length"]
    patch = patchify(x, cfg)
    if cfg["do_mask_input"]:
        if mask is None:
            mask = random_mask(cfg, B, C, P, seed)
        mask = np.asarray(mask).astype(bool)
        xin = np.where(mask[..., None], f(cfg["mask_value"]), patch)
    else:
        mask, xin = None, patch
    M = B * C * P
    xm = xin.reshape(M, pl)
    pre = "encoder.encoder."
    h = xm @ P_[pre + "embedder.input_embedding.weight"].T + P_[pre + "embedder.input_embedding.bias"]
    h = (h.reshape(B * C, P, D) + P_[pre + "positional_encoder.position_enc"]).reshape(M, D)
    pdm = R.keep_mask(seed, SITE_POS_DROP, M * D, cfg["positional_dropout"] if train else 0.0).reshape(M, D).astype(f)
    h = h * pdm
    embed = h
    pa = cfg["attention_dropout"] if train else 0.0
    pp = cfg["path_dropout"] if train else 0.0
    pf = cfg["ff_dropout"] if train else 0.0
    scale = f(hd ** -0.5)
    layers, new_bufs = [], dict(bufs)

    def heads(t):
        return t.reshape(B * C, P, nh, hd).transpose(0, 2, 1, 3)

    for l in range(L):
        lp = pre + f"layers.{l}."
        lc = {}
        n1 = lp + "norm_sublayer1.batchnorm."
        y1, lc["bn1"], (new_bufs[n1 + "running_mean"], new_bufs[n1 + "running_var"]) = bn_fwd(
            h, P_[n1 + "weight"], P_[n1 + "bias"], np.asarray(bufs[n1 + "running_mean"], f), np.asarray(bufs[n1 + "running_var"], f), train, cfg["norm_eps"])
        if cfg.get("fp8_qkv"):   # the HIP path's MX e4m3 projections (oracle/fp8.py): FORWARD values only, the backward is straight-through
            from .fp8 import linear_fp8
            lin = lambda nm: linear_fp8(y1, P_[lp + f"self_attn.{nm}.weight"], P_[lp + f"self_attn.{nm}.bias"]).astype(y1.dtype)
        else:
            lin = lambda nm: y1 @ P_[lp + f"self_attn.{nm}.weight"].T + P_[lp + f"self_attn.{nm}.bias"]
        q, k, v = heads(lin("q_proj")), heads(lin("k_proj")), heads(lin("v_proj"))
        s = (q @ k.transpose(0, 1, 3, 2)) * scale
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        prob = e / e.sum(-1, keepdims=True)
        pm = R.keep_mask(seed, site_layer(l, 0), B * C * nh * P * P, pa).reshape(B * C, nh, P, P).astype(f)
        pd = prob * pm
        a = (pd @ v).transpose(0, 2, 1, 3).reshape(M, D)
        d1 = R.keep_mask(seed, site_layer(l, 1), M * D, pp).reshape(M, D).astype(f)
        h_mid = h + (a @ P_[lp + "self_attn.out_proj.weight"].T + P_[lp + "self_attn.out_proj.bias"]) * d1
        n3 = lp + "norm_sublayer3.batchnorm."
        y3, lc["bn3"], (new_bufs[n3 + "running_mean"], new_bufs[n3 + "running_var"]) = bn_fwd(
            h_mid, P_[n3 + "weight"], P_[n3 + "bias"], np.asarray(bufs[n3 + "running_mean"], f), np.asarray(bufs[n3 + "running_var"], f), train, cfg["norm_eps"])
        u = y3 @ P_[lp + "ff.0.weight"].T + P_[lp + "ff.0.bias"]
        fm = R.keep_mask(seed, site_layer(l, 2), M * F, pf).reshape(M, F).astype(f)
        g = act_fwd(cfg["activation_function"], u) * fm
        d3 = R.keep_mask(seed, site_layer(l, 3), M * D, pp).reshape(M, D).astype(f)
        h_out = h_mid + (g @ P_[lp + "ff.3.weight"].T + P_[lp + "ff.3.bias"]) * d3
        if train:
            for n in (n1, n3):
                new_bufs[n + "num_batches_tracked"] = np.asarray(bufs[n + "num_batches_tracked"]) + 1
        lc.update(y1=y1, q=q, k=k, v=v, prob=prob, pm=pm, pd=pd, a=a, d1=d1, y3=y3, u=u, fm=fm, g=g, d3=d3, out=h_out)
        layers.append(lc)
        h = h_out
    out = {"embed": embed.reshape(B, C, P, D), "layer_out": [lc["out"].reshape(B, C, P, D) for lc in layers], "patch_input": patch}
    c = dict(cfg=cfg, P=P_, B=B, C=C, Pn=P, f=f, xm=xm, pdm=pdm, layers=layers, h_last=h)
    if cfg["head_dropout"] > 0:
        raise Exception("head_dropout > 0 is not restated (no reference config uses it)")
    if cfg["method"] == "ctc":
        pooled = h.reshape(B, C, P, D).mean(1).reshape(B * P, D)       # patchtst.py:89
        if cfg["mlp_decoder"]:
            ud = pooled @ P_["decoder.projection.0.weight"].T + P_["decoder.projection.0.bias"]
            dd = act_fwd(cfg["mlp_activation"], ud)
            logits = dd @ P_["decoder.projection.2.weight"].T + P_["decoder.projection.2.bias"]
            c.update(ud=ud, dd=dd)
        else:
            logits = pooled @ P_["decoder.projection.weight"].T + P_["decoder.projection.bias"]
        z = logits - logits.max(-1, keepdims=True)
        lp_ = (z - np.log(np.exp(z).sum(-1, keepdims=True))).reshape(B, P, -1)
        lens = np.trunc(1 + (np.asarray(batch["spikes_lengths"], np.float64) - pl) / cfg["patch_stride"]).astype(np.int64)   # :239
        out.update(preds=lp_.astype(f), token_lens=lens)
        if batch.get("targets") is not None:
            losses, dlogits = ctc_loss_and_grad(lp_, batch["targets"], lens, np.asarray(batch["targets_lengths"]).reshape(-1),
                                                blank=cfg["blank_id"], zero_infinity=cfg["zero_infinity"])
            out.update(loss=f(losses.sum()), loss_per_sample=losses, n_examples=np.int64(B))
            c["dlogits"] = dlogits.astype(f).reshape(B * P, -1)
        c["pooled"] = pooled
    else:
        if cfg["mlp_decoder"]:
            ud = h @ P_["decoder.projection.0.weight"].T + P_["decoder.projection.0.bias"]
            dd = act_fwd(cfg["mlp_activation"], ud)
            raw = dd @ P_["decoder.projection.2.weight"].T + P_["decoder.projection.2.bias"]
            c.update(ud=ud, dd=dd)
        else:
            raw = h @ P_["decoder.projection.weight"].T + P_["decoder.projection.bias"]
        rate_relu = not cfg["log_input"]                                 # PretrainHead.post_proj (patchtst.py:137)
        pr = np.maximum(raw, 0) if rate_relu else raw
        tg = patch.reshape(M, pl)
        valid = _patch_valid(np.asarray(batch["spikes_mask"], np.int64), cfg)      # (B,P)
        m = (mask & valid[:, None, :]).reshape(M)
        if cfg["loss"] == "poisson_nll":
            if cfg["log_input"]:
                el, dl = np.exp(pr) - tg * pr, np.exp(pr) - tg
            else:
                el, dl = pr - tg * np.log(pr + f(1e-8)), 1 - tg / (pr + f(1e-8))
        elif cfg["loss"] == "mse":
            el, dl = (pr - tg) ** 2, 2 * (pr - tg)
        else:
            raise Exception(f"Loss {cfg['loss']} not implemented yet for mlm")
        draw = (dl * m[:, None]).astype(f)
        if rate_relu:
            draw = draw * (raw > 0)
        out.update(preds=pr.reshape(B, C, P, pl).astype(f), loss=f((el * m[:, None]).sum()), n_examples=np.int64(m.sum()),
                   mask=m.reshape(B, C, P), raw_mask=mask)
        c["draw"] = draw
    return out, c, new_bufs


def backward(c, grad_scale=1.0):
    cfg, P_, B, C, P, f = c["cfg"], c["P"], c["B"], c["C"], c["Pn"], c["f"]
    D, L, nh = cfg["d_model"], cfg["num_hidden_layers"], cfg["num_attention_heads"]
    hd = D // nh
    M = B * C * P
    scale = f(hd ** -0.5)
    g = {}
    if cfg["method"] == "ctc":
        dl = c["dlogits"] * f(grad_scale)
        src = c["pooled"]
    else:
        dl = c["draw"] * f(grad_scale)
        src = c["h_last"]
    if cfg["mlp_decoder"]:
        g["decoder.projection.2.weight"] = dl.T @ c["dd"]; g["decoder.projection.2.bias"] = dl.sum(0)
        dud = (dl @ P_["decoder.projection.2.weight"]) * act_bwd(cfg["mlp_activation"], c["ud"])
        g["decoder.projection.0.weight"] = dud.T @ src; g["decoder.projection.0.bias"] = dud.sum(0)
        dsrc = dud @ P_["decoder.projection.0.weight"]
    else:
        g["decoder.projection.weight"] = dl.T @ src; g["decoder.projection.bias"] = dl.sum(0)
        dsrc = dl @ P_["decoder.projection.weight"]
    if cfg["method"] == "ctc":
        dh = np.broadcast_to((dsrc / f(C)).reshape(B, 1, P, D), (B, C, P, D)).reshape(M, D).astype(f)
    else:
        dh = dsrc.astype(f)
    pre = "encoder.encoder."

    def merge(t):
        return t.transpose(0, 2, 1, 3).reshape(M, D)

    for l in range(L - 1, -1, -1):
        lp = pre + f"layers.{l}."
        lc = c["layers"][l]
        dm = dh * lc["d3"]
        g[lp + "ff.3.weight"] = dm.T @ lc["g"]; g[lp + "ff.3.bias"] = dm.sum(0)
        du = (dm @ P_[lp + "ff.3.weight"]) * lc["fm"] * act_bwd(cfg["activation_function"], lc["u"])
        g[lp + "ff.0.weight"] = du.T @ lc["y3"]; g[lp + "ff.0.bias"] = du.sum(0)
        n3 = lp + "norm_sublayer3.batchnorm."
        d3, g[n3 + "weight"], g[n3 + "bias"] = bn_bwd(du @ P_[lp + "ff.0.weight"], lc["bn3"], P_[n3 + "weight"])
        dh = dh + d3
        c1 = dh * lc["d1"]
        g[lp + "self_attn.out_proj.weight"] = c1.T @ lc["a"]; g[lp + "self_attn.out_proj.bias"] = c1.sum(0)
        da = (c1 @ P_[lp + "self_attn.out_proj.weight"]).reshape(B * C, P, nh, hd).transpose(0, 2, 1, 3)
        dv = lc["pd"].transpose(0, 1, 3, 2) @ da
        dp = (da @ lc["v"].transpose(0, 1, 3, 2)) * lc["pm"]
        ds = lc["prob"] * (dp - (dp * lc["prob"]).sum(-1, keepdims=True))
        dq = merge((ds @ lc["k"]) * scale)
        dk = merge((ds.transpose(0, 1, 3, 2) @ lc["q"]) * scale)
        dv = merge(dv)
        dy1 = np.zeros((M, D), f)
        for nm, dd in (("q_proj", dq), ("k_proj", dk), ("v_proj", dv)):
            g[lp + f"self_attn.{nm}.weight"] = dd.T @ lc["y1"]; g[lp + f"self_attn.{nm}.bias"] = dd.sum(0)
            dy1 = dy1 + dd @ P_[lp + f"self_attn.{nm}.weight"]
        n1 = lp + "norm_sublayer1.batchnorm."
        d1, g[n1 + "weight"], g[n1 + "bias"] = bn_bwd(dy1, lc["bn1"], P_[n1 + "weight"])
        dh = dh + d1
    de = dh * c["pdm"]
    g[pre + "embedder.input_embedding.weight"] = de.T @ c["xm"]
    g[pre + "embedder.input_embedding.bias"] = de.sum(0)
    return {k: v.astype(f) for k, v in g.items()}
