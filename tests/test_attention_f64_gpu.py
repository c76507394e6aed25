"""GPU, kernel level: nbci_attention_fwd / nbci_attention_bwd (attention.hip: head 128, T' <= 160, bf16) against an f64 numpy
attention on the SAME bf16-rounded inputs, with the mask of models/ndt1.py:435-437 (eye | context & key validity) and the dropout
masks of oracle/rng.py. The tolerance is DERIVED per element from the reference's own magnitudes, not a flat number:
what the kernels round is (a) the probabilities / dS to bf16 before the second product (2^-9 relative each), (b) the outputs to bf16
(2^-9), (c) in the backward, delta = da . O from the stored bf16 forward output (2^-9 per term). Every bound below is those sums with
a factor 2 of head room; the test prints the worst ratio error / bound so that a kernel drifting toward its bound is visible.
Covers the one-launch backward (T' <= 144: dS / Pd through LDS), the dq + dk/dv pair (T' = 150), ragged key validity, context
spans (ndt1.py:30-41), dropout off / on, and the fused q/k/v bias sums."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import rng as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
HD = 128
EPS = 2.0 ** -8   # two bf16 roundings' worth (2 x 2^-9): the head room factor


def _bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(torch.bfloat16)


def _f64(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _ctx_ok(i, j, f, bk):
    ok = np.ones((len(i), len(j)), bool)
    d = j[None, :] - i[:, None]
    if f >= -1:
        ok &= ~(d > f)
    if bk >= -1:
        ok &= ~(-d > bk)
    return ok


def _keep(seed, site, idx, p):
    thr = R.drop_threshold(p)
    if thr == 0:
        return np.ones(idx.shape, np.float64)
    idx = idx.astype(np.uint32)
    h = R.mix32((idx >> np.uint32(1)) ^ np.uint32(R.drop_key(seed, site)))
    draws = np.where((idx & np.uint32(1)) == 1, h >> np.uint32(16), h & np.uint32(0xFFFF))
    return np.where(draws >= np.uint32(thr), 1.0 / (1.0 - float(np.float32(p))), 0.0)


def _reference(qkv, tmask, g_out, B, nh, Tp, cf, cb, p, seed, sp, so):
    H = nh * HD
    x = qkv.reshape(B, Tp, 3, nh, HD)
    scale = 1.0 / np.sqrt(HD)
    ar = np.arange(Tp)
    out = dict(ad=np.zeros((B, Tp, H)), lse=np.zeros((B, nh, Tp)), dqkv=np.zeros((B, Tp, 3, nh, HD)), tol=np.zeros((B, Tp, 3, nh, HD)),
               tol_ad=np.zeros((B, Tp, H)), da=np.zeros((B, Tp, H)))
    for b in range(B):
        for h in range(nh):
            q, k, v = x[b, :, 0, h], x[b, :, 1, h], x[b, :, 2, h]
            valid = np.eye(Tp, dtype=bool) | (_ctx_ok(ar, ar, cf, cb) & (tmask[b][None, :] != 0))
            S = np.where(valid, q @ k.T * scale, -np.inf)
            mx = S.max(1, keepdims=True)
            E = np.exp(S - mx)
            P = E / E.sum(1, keepdims=True)
            out["lse"][b, h] = (mx[:, 0] + np.log(E.sum(1)))
            kp = _keep(seed, sp, ((b * nh + h) * Tp + ar[:, None]) * Tp + ar[None, :], p)
            Pd = P * kp
            O = Pd @ v
            ko = _keep(seed, so, (b * Tp + ar[:, None]) * H + h * HD + np.arange(HD)[None, :], p)
            out["ad"][b, :, h * HD:(h + 1) * HD] = O * ko
            out["tol_ad"][b, :, h * HD:(h + 1) * HD] = (EPS * (np.abs(Pd) @ np.abs(v)) + EPS * np.abs(O)) * ko + 1e-6
            da = g_out[b, :, h * HD:(h + 1) * HD] * (ko > 0)      # what the out_proj data gradient hands over: zero where dropped
            out["da"][b, :, h * HD:(h + 1) * HD] = da
    return out, x


def _backward_reference(x, tmask, da_all, ad_stored, B, nh, Tp, cf, cb, p, seed, sp):
    """f64 backward from the bf16-rounded da and the STORED (bf16) forward output, as the kernels see them."""
    H = nh * HD
    scale = 1.0 / np.sqrt(HD)
    o_scale = 1.0 / (1.0 - float(np.float32(p))) if p > 0 else 1.0
    ar = np.arange(Tp)
    dqkv = np.zeros((B, Tp, 3, nh, HD)); tol = np.zeros_like(dqkv)
    for b in range(B):
        for h in range(nh):
            q, k, v = x[b, :, 0, h], x[b, :, 1, h], x[b, :, 2, h]
            valid = np.eye(Tp, dtype=bool) | (_ctx_ok(ar, ar, cf, cb) & (tmask[b][None, :] != 0))
            S = np.where(valid, q @ k.T * scale, -np.inf)
            E = np.exp(S - S.max(1, keepdims=True))
            P = E / E.sum(1, keepdims=True)
            kp = _keep(seed, sp, ((b * nh + h) * Tp + ar[:, None]) * Tp + ar[None, :], p)
            Pd = P * kp
            da = da_all[b, :, h * HD:(h + 1) * HD]
            ad = ad_stored[b, :, h * HD:(h + 1) * HD]
            delta = (da * ad).sum(1) / o_scale
            d_abs = np.abs(da * ad).sum(1) / o_scale
            dP = (da @ v.T) * kp
            dS = P * (dP - delta[:, None]) * scale
            dS_err = EPS * np.abs(dS) + EPS * scale * P * d_abs[:, None] + EPS * scale * P * (np.abs(da) @ np.abs(v).T) * kp
            dqkv[b, :, 0, h] = dS @ k
            dqkv[b, :, 1, h] = dS.T @ q
            dqkv[b, :, 2, h] = Pd.T @ da
            tol[b, :, 0, h] = dS_err @ np.abs(k) + EPS * np.abs(dqkv[b, :, 0, h]) + 1e-6
            tol[b, :, 1, h] = dS_err.T @ np.abs(q) + EPS * np.abs(dqkv[b, :, 1, h]) + 1e-6
            tol[b, :, 2, h] = EPS * (np.abs(Pd).T @ np.abs(da)) + EPS * np.abs(dqkv[b, :, 2, h]) + 1e-6
    return dqkv, tol


CASES = [  # B, heads, T', lengths, (ctx fwd, ctx bwd), dropout
    (3, 2, 143, [143, 101, 60], (-2, -2), 0.0),
    (3, 2, 143, [143, 101, 60], (-2, -2), 0.4),
    (2, 2, 144, [144, 17], (-2, -2), 0.4),
    (2, 3, 150, [150, 99], (-2, -2), 0.4),      # > 144: the dq + dk/dv kernel pair
    (2, 2, 97, [97, 33], (3, 10), 0.4),
    (2, 1, 143, [120, 143], (-1, 5), 0.0),
    (4, 1, 16, [16, 9, 1, 5], (-2, -2), 0.2),
    (2, 2, 5, [5, 2], (-2, -2), 0.0),
    (2, 2, 33, [33, 20], (0, -2), 0.3),
]


@pytest.mark.parametrize("B,nh,Tp,lens,ctx,p", CASES)
def test_fused_attention_forward_and_backward_against_f64(B, nh, Tp, lens, ctx, p):
    from llm_bci_amd._lib import check, lib
    l = lib()
    H = nh * HD
    g = np.random.default_rng(Tp * 7 + nh)
    qkv_t = _bf16(g.standard_normal((B * Tp, 3 * H)) * 1.5)
    gout_t = _bf16(g.standard_normal((B, Tp, H)))
    tmask = np.zeros((B, Tp), np.int32)
    for b, L in enumerate(lens):
        tmask[b, :L] = 1
    tm_t = torch.from_numpy(tmask).to(DEV)
    seed, sp, so = 1234, 16, 17
    cf, cb = ctx
    ref, x = _reference(_f64(qkv_t), tmask, _f64(gout_t), B, nh, Tp, cf, cb, p, seed, sp, so)

    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ad = torch.zeros(B * Tp, H, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(B, nh, Tp, dtype=torch.float32, device=DEV)
    check(l.nbci_attention_fwd(P(qkv_t), P(tm_t), P(ad), P(lse), B, nh, Tp, H, cf, cb, p, seed, sp, so, st), "attention fwd")
    torch.cuda.synchronize()
    got_ad = _f64(ad).reshape(B, Tp, H)
    err = np.abs(got_ad - ref["ad"])
    r_fwd = float((err / ref["tol_ad"]).max())
    assert r_fwd <= 1.0, f"forward output: error / bound = {r_fwd:.3f}"
    assert np.abs(_f64(lse) - ref["lse"]).max() <= 2e-4

    # backward: da = (gradient) * keep_out, rounded to bf16, as the out_proj data-gradient GEMM writes it
    da_t = _bf16(ref["da"])
    ldP = (Tp + 7) // 8 * 8
    dS = torch.zeros(B * nh * Tp * ldP, dtype=torch.bfloat16, device=DEV)
    Pd = torch.zeros_like(dS)
    dqkv = torch.zeros(B * Tp, 3 * H, dtype=torch.bfloat16, device=DEV)
    bias = torch.zeros(3 * H, dtype=torch.float32, device=DEV)
    check(l.nbci_attention_bwd(P(qkv_t), P(tm_t), P(ad), P(lse), P(da_t), P(dS), P(Pd), ldP, P(dqkv), P(bias), B, nh, Tp, H, cf, cb, p, seed, sp, st),
          "attention bwd")
    torch.cuda.synchronize()
    want, tol = _backward_reference(x, tmask, _f64(da_t).reshape(B, Tp, H), got_ad, B, nh, Tp, cf, cb, p, seed, sp)
    got = _f64(dqkv).reshape(B, Tp, 3, nh, HD)
    worst = {}
    for i, nm in enumerate(("dq", "dk", "dv")):
        r = float((np.abs(got[:, :, i] - want[:, :, i]) / tol[:, :, i]).max())
        worst[nm] = round(r, 3)
        assert r <= 1.0, f"{nm}: error / bound = {r:.3f}"
        assert np.abs(want[:, :, i]).max() > 0
    print(f"attention f64 parity T'={Tp} p={p} ctx={ctx}: error / derived bound: fwd {r_fwd:.3f}, bwd {worst}")
    # the fused bias sums = column sums of the stored gradient
    cs = _f64(dqkv).sum(0)
    assert np.abs(_f64(bias) - cs).max() <= 1e-3 * max(1.0, np.abs(cs).max())
    # rows of padded queries still attend to themselves (ndt1.py:436) and get a finite gradient
    assert np.isfinite(got).all() and np.isfinite(got_ad).all()
