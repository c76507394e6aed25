"""GPU, kernel level: the streaming attention of attn_flash.hip (nbci_attention_flash_fwd / _bwd: the 32 x 32-tile kernels at heads 32 / 64 / 96,
bf16, any length; unmasked - torch.nn.TransformerEncoderLayer under models/itransformer.py:158-173 and the PatchTST encoder, models/patchtst.py:176)
against an f64 numpy attention on the SAME bf16-rounded inputs, with the dropout masks of oracle/rng.py. As in test_attention_f64_gpu.py the
tolerance is DERIVED per element from the reference's own magnitudes: the kernels round the (kept) probabilities and dS to bf16 before the second
products (2^-9 relative each), the outputs to bf16 (2^-9), and the backward takes delta = dO . O from the stored bf16 forward output; every bound
is those sums with a factor 2 of head room, and the test prints the worst error / bound. Shapes: the iTransformer's (1501 tokens, head 96), PatchTST's
(205 patches, head 32), head 64, lengths below / at / just above one 32-key step, dropout off and on (the keep words the dq kernel hands to the
dk/dv kernel are exercised whenever p > 0)."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_attention_f64_gpu import EPS, _bf16, _f64, _keep

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _reference(x, dout, NS, nh, S, hd, p, seed, site):
    """x (NS, S, 3, nh, hd), dout (NS, S, nh, hd) f64 -> forward output, its bound, log-sum-exp and what the backward needs"""
    scale = 1.0 / np.sqrt(hd)
    ar = np.arange(S)
    O = np.zeros((NS, S, nh, hd)); tolO = np.zeros_like(O); lse = np.zeros((NS, nh, S)); keep = {}
    for b in range(NS):
        for h in range(nh):
            q, k, v = x[b, :, 0, h], x[b, :, 1, h], x[b, :, 2, h]
            sc = q @ k.T * scale
            mx = sc.max(1, keepdims=True)
            E = np.exp(sc - mx)
            P = E / E.sum(1, keepdims=True)
            lse[b, h] = mx[:, 0] + np.log(E.sum(1))
            kp = _keep(seed, site, ((b * nh + h) * S + ar[:, None]) * S + ar[None, :], p)
            keep[b, h] = (P, kp)
            Pd = P * kp
            O[b, :, h] = Pd @ v
            tolO[b, :, h] = EPS * (np.abs(Pd) @ np.abs(v)) + EPS * np.abs(O[b, :, h]) + 1e-6
    return O, tolO, lse, keep


def _backward(x, dout, out_stored, keep, NS, nh, S, hd):
    scale = 1.0 / np.sqrt(hd)
    g = np.zeros((NS, S, 3, nh, hd)); tol = np.zeros_like(g)
    for b in range(NS):
        for h in range(nh):
            q, k, v = x[b, :, 0, h], x[b, :, 1, h], x[b, :, 2, h]
            P, kp = keep[b, h]
            Pd = P * kp
            da, ad = dout[b, :, h], out_stored[b, :, h]
            delta = (da * ad).sum(1)
            d_abs = np.abs(da * ad).sum(1)
            dP = (da @ v.T) * kp
            dS = P * (dP - delta[:, None]) * scale
            dS_err = EPS * np.abs(dS) + EPS * scale * P * d_abs[:, None] + EPS * scale * P * (np.abs(da) @ np.abs(v).T) * kp
            g[b, :, 0, h] = dS @ k
            g[b, :, 1, h] = dS.T @ q
            g[b, :, 2, h] = Pd.T @ da
            tol[b, :, 0, h] = dS_err @ np.abs(k) + EPS * np.abs(g[b, :, 0, h]) + 1e-6
            tol[b, :, 1, h] = dS_err.T @ np.abs(q) + EPS * np.abs(g[b, :, 1, h]) + 1e-6
            tol[b, :, 2, h] = EPS * (np.abs(Pd).T @ np.abs(da)) + EPS * np.abs(g[b, :, 2, h]) + 1e-6
    return g, tol


CASES = [  # sequences, heads, length, head size, dropout
    (1, 2, 1501, 96, 0.4),    # the iTransformer's channel-token encoder (models/itransformer.py:158-173 at 1500 channels + CLS)
    (1, 2, 1501, 96, 0.0),
    (3, 4, 205, 32, 0.2),     # PatchTST: 205 patches, head 32
    (2, 3, 97, 64, 0.3),
    (2, 2, 333, 96, 0.0),
    (2, 2, 25, 96, 0.3),      # one ragged step
    (2, 2, 64, 32, 0.4),      # exactly two full steps, no ragged one
    (1, 3, 33, 64, 0.0),      # one full step + one key
    (2, 4, 101, 32, 0.0),     # the iTransformer's per-channel bin-token embedder (itransformer.py:40-93: CLS + 100 bins, 128 x 4 heads)
]


@pytest.mark.parametrize("NS,nh,S,hd,p", CASES)
def test_streaming_attention_forward_and_backward_against_f64(NS, nh, S, hd, p):
    from llm_bci_amd._lib import check, lib
    l = lib()
    H = nh * hd
    g = np.random.default_rng(S * 11 + hd)
    qkv_t = _bf16(g.standard_normal((NS * S, 3 * H)) * 1.2)
    dout_t = _bf16(g.standard_normal((NS * S, H)))
    seed, site = 4321, 6
    x = _f64(qkv_t).reshape(NS, S, 3, nh, hd)
    O, tolO, lse_ref, keep = _reference(x, None, NS, nh, S, hd, p, seed, site)

    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.zeros(NS * S, H, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(NS * nh * S, dtype=torch.float32, device=DEV)
    dsum = torch.zeros_like(lse)
    dqkv = torch.zeros(NS * S, 3 * H, dtype=torch.bfloat16, device=DEV)
    check(l.nbci_attention_flash_fwd(P(qkv_t), P(out), P(lse), NS, nh, S, H, p, seed, site, st), "flash fwd")
    torch.cuda.synchronize()
    got_o = _f64(out).reshape(NS, S, nh, hd)
    r_fwd = float((np.abs(got_o - O) / tolO).max())
    assert r_fwd <= 1.0, f"forward output: error / bound = {r_fwd:.3f}"
    assert np.abs(_f64(lse).reshape(NS, nh, S) - lse_ref).max() <= 2e-4

    check(l.nbci_attention_flash_bwd(P(qkv_t), P(out), P(dout_t), P(lse), P(dsum), P(dqkv), NS, nh, S, H, p, seed, site, st), "flash bwd")
    torch.cuda.synchronize()
    want, tol = _backward(x, _f64(dout_t).reshape(NS, S, nh, hd), got_o, keep, NS, nh, S, hd)
    got = _f64(dqkv).reshape(NS, S, 3, nh, hd)
    worst = {}
    for i, nm in enumerate(("dq", "dk", "dv")):
        r = float((np.abs(got[:, :, i] - want[:, :, i]) / tol[:, :, i]).max())
        worst[nm] = round(r, 3)
        assert r <= 1.0, f"{nm}: error / bound = {r:.3f}"
        assert np.abs(want[:, :, i]).max() > 0
    # dO . O as the dq kernel leaves it for the dk / dv kernel
    delta = (_f64(dout_t).reshape(NS, S, nh, hd) * got_o).sum(-1).transpose(0, 2, 1)
    assert np.abs(_f64(dsum).reshape(NS, nh, S) - delta).max() <= 1e-3 * max(1.0, np.abs(delta).max())
    assert np.isfinite(got).all() and np.isfinite(got_o).all()
    print(f"streaming attention f64 parity S={S} head {hd} p={p}: error / derived bound: fwd {r_fwd:.3f}, bwd {worst}")
