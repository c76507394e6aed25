"""numpy restatement of the reference's BCI coupler (models/bci.py): `projector` MLP (:88-96), `prepare_embeds` stacking + splice
(:107-168) and the shifted cross-entropy sum of `forward` (:201-212), forward AND hand-derived backward.
Test infrastructure only — see oracle/__init__.py. The HF LLM between the splice and the loss is third-party and is NOT restated:
tests feed this file the LLM's logits / the gradient of its input embeddings.

Pinned by tests/golden/g_bci.npz and g_bci_fwd.npz (generated from the reference's own BCI.prepare_embeds / BCI.forward).
Parameter names are the reference's `projector` state-dict keys: "0.weight", "0.bias", "2.weight", "2.bias" for
nn.Sequential(Linear, act, Linear) (inter_size set), "weight" / "bias" for the single Linear (inter_size null).
The 41-input / tanh / no-bias coupler of configs/phoneme_coupler.yaml:1-7 is the same function with other arguments.
"""
import math

import numpy as np

from .ndt1 import act_bwd, act_fwd


def stack_tokens(hidden, token_mask, stacking):
    """bci.py:127-141: zero-pad T' to a multiple of `stacking`, view (B, T'/s, H*s); a stacked feature is valid iff all of its
    `stacking` source tokens are valid."""
    B, T, H = hidden.shape
    s = int(stacking)
    if T % s:
        new_T = math.ceil(T / s) * s
        hidden = np.concatenate([hidden, np.zeros((B, new_T - T, H), hidden.dtype)], 1)
        token_mask = np.concatenate([token_mask, np.zeros((B, new_T - T), token_mask.dtype)], 1)
        T = new_T
    x = hidden.reshape(B, T // s, H * s)
    valid = (token_mask.reshape(B, T // s, s).sum(-1) == s).astype(np.int64)
    return x, valid


def projector_fwd(x, p, act="relu"):
    """bci.py:88-96. x (..., in). Returns y and the cache for projector_bwd."""
    f = x.dtype
    x2 = x.reshape(-1, x.shape[-1])
    if "0.weight" in p:
        u = x2 @ np.asarray(p["0.weight"], f).T
        if "0.bias" in p:
            u = u + np.asarray(p["0.bias"], f)
        h = act_fwd(act, u)
        y = h @ np.asarray(p["2.weight"], f).T
        if "2.bias" in p:
            y = y + np.asarray(p["2.bias"], f)
        cache = dict(x2=x2, u=u, h=h, act=act, two=True)
    else:
        y = x2 @ np.asarray(p["weight"], f).T
        if "bias" in p:
            y = y + np.asarray(p["bias"], f)
        cache = dict(x2=x2, two=False)
    return y.reshape(x.shape[:-1] + (y.shape[-1],)), cache


def projector_bwd(dy, p, cache):
    """gradients wrt the projector parameters and its input; dy has y's shape."""
    f = dy.dtype
    dy2 = dy.reshape(-1, dy.shape[-1])
    g = {}
    if cache["two"]:
        g["2.weight"] = dy2.T @ cache["h"]
        if "2.bias" in p:
            g["2.bias"] = dy2.sum(0)
        dh = dy2 @ np.asarray(p["2.weight"], f)
        du = dh * act_bwd(cache["act"], cache["u"])
        g["0.weight"] = du.T @ cache["x2"]
        if "0.bias" in p:
            g["0.bias"] = du.sum(0)
        dx = du @ np.asarray(p["0.weight"], f)
    else:
        g["weight"] = dy2.T @ cache["x2"]
        if "bias" in p:
            g["bias"] = dy2.sum(0)
        dx = dy2 @ np.asarray(p["weight"], f)
    return g, dx


def splice_fwd(text, spikes, text_mask, spikes_valid, targets, split):
    """bci.py:143-166: per example cat(text[:d], spikes, text[d:]) for the embeddings, the attention mask (text mask / spike
    validity) and the targets (-100 over the spike span)."""
    B, Lt, H = text.shape
    Ts = spikes.shape[1]
    emb = np.zeros((B, Lt + Ts, H), spikes.dtype)
    mask = np.zeros((B, Lt + Ts), np.int64)
    tg = None if targets is None else np.zeros((B, Lt + Ts), np.int64)
    for b in range(B):
        d = int(split[b])
        emb[b] = np.concatenate([text[b, :d], spikes[b], text[b, d:]], 0)
        mask[b] = np.concatenate([text_mask[b, :d], spikes_valid[b], text_mask[b, d:]], 0)
        if tg is not None:
            tg[b] = np.concatenate([targets[b, :d], np.full(Ts, -100, np.int64), targets[b, d:]], 0)
    return emb, mask, tg


def splice_bwd(d_emb, split, Lt, Ts):
    B, L, H = d_emb.shape
    d_text = np.zeros((B, Lt, H), d_emb.dtype)
    d_sp = np.zeros((B, Ts, H), d_emb.dtype)
    for b in range(B):
        d = int(split[b])
        d_text[b, :d] = d_emb[b, :d]
        d_sp[b] = d_emb[b, d:d + Ts]
        d_text[b, d:] = d_emb[b, d + Ts:]
    return d_text, d_sp


def shifted_ce_sum(logits, targets):
    """bci.py:201-212: tokens < n predict n; CrossEntropyLoss(reduction="sum", ignore_index=-100) on the flattened shifted
    logits; n_examples = number of labels != -100. Returns loss, n_examples, d loss / d logits (float64 inside)."""
    lg = np.asarray(logits, np.float64)[:, :-1]
    tg = np.asarray(targets)[:, 1:]
    z = lg - lg.max(-1, keepdims=True)
    lp = z - np.log(np.exp(z).sum(-1, keepdims=True))
    keep = tg != -100
    idx = np.where(keep, tg, 0)
    picked = np.take_along_axis(lp, idx[..., None], -1)[..., 0]
    loss = -(picked * keep).sum()
    d = np.exp(lp) * keep[..., None]
    np.put_along_axis(d, idx[..., None], np.take_along_axis(d, idx[..., None], -1) - keep[..., None], -1)
    dlogits = np.zeros(np.asarray(logits).shape, np.float64)
    dlogits[:, :-1] = d
    return loss, int(keep.sum()), dlogits
