"""CTC loss + gradient, restating torch.nn.CTCLoss(reduction="none", blank, zero_infinity)
as the reference uses it (models/ndt1.py:517,581). Test infrastructure only.

The algorithm lives in PyTorch (ATen/native/LossCTC.cpp, torch 2.10 in the build container),
not under /root/reference; this file restates the published alpha/beta recursion (Graves et
al. 2006) with PyTorch's conventions, pinned by tests/golden fixtures produced from
torch.nn.CTCLoss itself:
  * log_probs are log-softmax outputs (T, B, V); targets (B, S) padded, only [:target_len] read;
  * loss_b = -log p(target_b | log_probs[:input_len_b, b]); infeasible -> +inf, or 0 with
    zero_infinity;
  * backward returns, for t < input_len: exp(lp[t,c]) - exp(logsum_{s: l'_s = c}(alpha+beta) + nll - lp[t,c])
    and 0 for t >= input_len (and all-zero for a sample whose loss was infinite when
    zero_infinity). Rows sum to 0, so it is also d loss / d logits.
"""
import numpy as np

NEG_INF = -np.inf


def _logaddexp3(a, b, c):
    m = np.maximum(np.maximum(a, b), c)
    m_safe = np.where(np.isfinite(m), m, 0.0)
    s = np.exp(a - m_safe) + np.exp(b - m_safe) + np.exp(c - m_safe)
    with np.errstate(divide="ignore"):
        return np.where(np.isfinite(m), m_safe + np.log(s), NEG_INF)


def _shift_right(a, n):
    out = np.full_like(a, NEG_INF)
    if n < len(a):
        out[n:] = a[:len(a) - n]
    return out


def _shift_left(a, n):
    out = np.full_like(a, NEG_INF)
    if n < len(a):
        out[:len(a) - n] = a[n:]
    return out


def ctc_loss_and_grad(log_probs_btv, targets, input_lengths, target_lengths, blank=0, zero_infinity=True,
                      want_grad=True):
    """log_probs_btv: (B, T, V) float (log-softmax). Returns (loss[B], grad[B,T,V] or None)."""
    lp_all = np.asarray(log_probs_btv, dtype=np.float64)
    B, T, V = lp_all.shape
    losses = np.zeros(B, np.float64)
    grad = np.zeros((B, T, V), np.float64) if want_grad else None
    for b in range(B):
        Tb = int(input_lengths[b])
        Sb = int(target_lengths[b])
        tgt = np.asarray(targets[b][:Sb], dtype=np.int64)
        L = 2 * Sb + 1
        ext = np.full(L, blank, np.int64)
        ext[1::2] = tgt
        lp = lp_all[b]
        # can_skip[s]: transition s-2 -> s allowed (s odd label differing from the previous label)
        can_skip = np.zeros(L, bool)
        if L > 2:
            can_skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        alpha = np.full((max(Tb, 1), L), NEG_INF)
        if Tb > 0:
            alpha[0, 0] = lp[0, ext[0]]
            if L > 1:
                alpha[0, 1] = lp[0, ext[1]]
            for t in range(1, Tb):
                a0 = alpha[t - 1]
                a1 = _shift_right(a0, 1)
                a2 = _shift_right(a0, 2)
                a2 = np.where(can_skip, a2, NEG_INF)
                alpha[t] = _logaddexp3(a0, a1, a2) + lp[t, ext]
            ll = np.logaddexp(alpha[Tb - 1, L - 1], alpha[Tb - 1, L - 2] if L > 1 else NEG_INF)
        else:
            ll = 0.0 if L == 1 and Sb == 0 else NEG_INF
            if Tb == 0 and Sb == 0:
                ll = 0.0
        nll = -ll
        if not np.isfinite(nll):
            losses[b] = 0.0 if zero_infinity else np.inf
            if want_grad and not zero_infinity and Tb > 0:
                grad[b, :Tb] = np.nan
            continue
        losses[b] = nll
        if not want_grad or Tb == 0:
            continue
        beta = np.full((Tb, L), NEG_INF)
        beta[Tb - 1, L - 1] = lp[Tb - 1, ext[L - 1]]
        if L > 1:
            beta[Tb - 1, L - 2] = lp[Tb - 1, ext[L - 2]]
        # skip_fwd[s]: transition s -> s+2 allowed
        skip_fwd = np.zeros(L, bool)
        if L > 2:
            skip_fwd[:-2] = can_skip[2:]
        for t in range(Tb - 2, -1, -1):
            b0 = beta[t + 1]
            b1 = _shift_left(b0, 1)
            b2 = _shift_left(b0, 2)
            b2 = np.where(skip_fwd, b2, NEG_INF)
            beta[t] = _logaddexp3(b0, b1, b2) + lp[t, ext]
        ab = alpha[:Tb] + beta  # (Tb, L); note both include lp[t, ext[s]] once -> subtract lp below
        # log-sum over states sharing a class
        lcab = np.full((Tb, V), NEG_INF)
        for s in range(L):
            lcab[:, ext[s]] = np.logaddexp(lcab[:, ext[s]], ab[:, s])
        with np.errstate(over="ignore"):
            occ = np.exp(lcab + nll - lp[:Tb])
        grad[b, :Tb] = np.exp(lp[:Tb]) - occ
    return losses, grad


def ctc_greedy_path(log_probs_btv):
    """argmax over V for ALL frames (main.py:69 does no length masking). int64 (B, T)."""
    return np.argmax(np.asarray(log_probs_btv), axis=-1).astype(np.int64)
