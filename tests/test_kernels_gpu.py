"""Single-kernel GPU parity through the C-ABI: CTC vs torch.nn.CTCLoss fixtures, AdamW vs oracle,
LayerNorm / softmax vs float64 numpy, smoothing vs oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ndt1 as O
from oracle import optim as OO
from oracle import rng as R
from test_oracle_golden import load

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _l():
    from llm_bci_amd import ndt1  # noqa: F401  (registers ctypes signatures)
    from llm_bci_amd._lib import check, lib
    return lib(), check


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def d(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("nm", ["basic", "repeat_infeasible", "too_short", "empty_target", "long"])
def test_ctc_cases(nm):
    l, check = _l()
    fx = load("ctc_cases")
    lp = fx[nm + "_lp"].astype(np.float32)
    B, T, V = lp.shape
    tg = fx[nm + "_targets"].astype(np.int64); S = tg.shape[1]
    il = fx[nm + "_il"].astype(np.int32); tl = fx[nm + "_tl"].astype(np.int64)
    loss = torch.zeros(B, device=DEV)
    nws = l.nbci_ctc_workspace_floats(B, T, S)
    ws = torch.zeros(max(int(nws), 1), device=DEV)
    ldd = (V + 7) // 8 * 8
    dl = torch.full((B, T, ldd), 7.0, device=DEV)
    l.nbci_ctc.restype = C.c_int
    lpd, tgd, ild, tld = d(lp), d(tg), d(il), d(tl)   # keep the borrowed buffers alive across the call
    check(l.nbci_ctc(vp(lpd), vp(tgd), vp(ild), vp(tld), B, T, V, S, 0, 1, vp(loss), vp(ws), vp(dl), 0, ldd,
                     C.c_float(1.0), st()), "nbci_ctc")
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss.cpu().numpy(), fx[nm + "_loss"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dl[:, :, :V].cpu().numpy(), fx[nm + "_grad"], atol=2e-5)
    assert torch.all(dl[:, :, V:] == 0)


@pytest.mark.parametrize("T,S,lens,tls", [(600, 100, [600, 377, 16], [100, 40, 9]), (1024, 150, [1024, 1000], [150, 120])])
def test_ctc_long_sequences_chunked_gradient_pass(T, S, lens, tls):
    """up to the reference's max_F = 1024 tokens (configs/ndt1.yaml:39): frames x vocab no longer fits the LDS of the gradient pass,
    which then runs in chunks of frames; against the numpy oracle (itself pinned to torch.nn.CTCLoss by ctc_cases.npz)."""
    from oracle.ctc import ctc_loss_and_grad
    l, check = _l()
    g = np.random.default_rng(8)
    B, V = len(lens), 41
    lp = torch.log_softmax(torch.from_numpy(g.standard_normal((B, T, V)).astype(np.float32)), -1).numpy()
    tg = g.integers(1, V, (B, S)).astype(np.int64)
    il, tl = np.array(lens, np.int32), np.array(tls, np.int64)
    ref_loss, ref_grad = ctc_loss_and_grad(lp, tg, il.astype(np.int64), tl, blank=0, zero_infinity=True)
    loss = torch.zeros(B, device=DEV)
    ws = torch.zeros(int(l.nbci_ctc_workspace_floats(B, T, S)), device=DEV)
    ldd = (V + 7) // 8 * 8
    dl = torch.full((B, T, ldd), 7.0, device=DEV)
    lpd, tgd, ild, tld = d(lp), d(tg), d(il), d(tl)
    check(l.nbci_ctc(vp(lpd), vp(tgd), vp(ild), vp(tld), B, T, V, S, 0, 1, vp(loss), vp(ws), vp(dl), 0, ldd, C.c_float(1.0), st()), "nbci_ctc")
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=2e-5, atol=1e-4)
    # alpha + beta - lp + nll cancels numbers of magnitude nll ~ 2 000 - 3 800 down to O(1) in f32 (1.2e-4 - 2.4e-4 per ulp there):
    # the posteriors carry a relative error of ~1e-3 at these lengths whatever the kernel does (a chunk-boundary slip would be O(1))
    err = np.abs(dl[:, :, :V].cpu().numpy() - ref_grad)
    assert err.max() < 4e-3 and err.mean() < 5e-5, (err.max(), err.mean())
    assert torch.all(dl[:, :, V:] == 0)


def test_adamw_matches_oracle_and_refreshes_bf16_shadow():
    l, check = _l()
    g0 = np.random.default_rng(0)
    n = 4096 + 8
    p = g0.standard_normal(n).astype(np.float32); g = g0.standard_normal(n).astype(np.float32) * 0.1
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    pd, gd, md, vd = d(p), d(g), d(m), d(v)
    plp = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    for t in (1, 2, 3):
        lr, b1 = OO.onecycle(t - 1, 100, 1e-3, 0.0, 25)
        OO.adamw_step(p, g * 0.5, m, v, t, lr, b1, 0.999, 1e-8, 5e-5)
        check(l.nbci_adamw(vp(pd), vp(gd), vp(md), vp(vd), vp(plp), n, lr, b1, 0.999, 1e-8, 5e-5, 1 - b1 ** t, 1 - 0.999 ** t,
                           0.5, st()), "nbci_adamw")
    torch.cuda.synchronize()
    np.testing.assert_allclose(pd.cpu().numpy(), p, atol=2e-6)
    np.testing.assert_allclose(md.cpu().numpy(), m, atol=1e-7)
    np.testing.assert_allclose(vd.cpu().numpy(), v, atol=1e-8)
    assert torch.equal(plp, pd.bfloat16())


def test_adamw_variants_zero_grad_and_bf16_gradients():
    """nbci_adamw_zero = nbci_adamw + a cleared gradient; nbci_adamw_lp on bf16 gradients = nbci_adamw on the same values in f32."""
    l, check = _l()
    g0 = np.random.default_rng(3)
    n = 8192 + 16
    p = g0.standard_normal(n).astype(np.float32)
    g = torch.from_numpy(g0.standard_normal(n).astype(np.float32) * 0.1).to(DEV).bfloat16()      # bf16-representable gradients
    args = (n, 1e-3, 0.9, 0.999, 1e-8, 5e-5, 0.1, 0.001, 0.5)
    res = {}
    for kind in ("plain", "zero", "lp"):
        pd, md, vd = d(p), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        plp = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        gf = g.float().clone()
        if kind == "plain":
            check(l.nbci_adamw(vp(pd), vp(gf), vp(md), vp(vd), vp(plp), *args, st()), "nbci_adamw")
        elif kind == "zero":
            check(l.nbci_adamw_zero(vp(pd), vp(gf), vp(md), vp(vd), vp(plp), *args, 256, st()), "nbci_adamw_zero")
            assert float(gf.abs().max()) == 0.0
        else:
            check(l.nbci_adamw_lp(vp(pd), vp(g), vp(md), vp(vd), vp(plp), *args, st()), "nbci_adamw_lp")
        torch.cuda.synchronize()
        res[kind] = (pd.clone(), md.clone(), vd.clone(), plp.clone())
    for kind in ("zero", "lp"):
        for a, b in zip(res["plain"], res[kind]):
            assert torch.equal(a, b), kind


@pytest.mark.parametrize("H", [32, 1024, 768])
def test_layernorm_fwd_bwd(H):
    l, check = _l()
    g0 = np.random.default_rng(1)
    M = 77
    x = g0.standard_normal((M, H)).astype(np.float32) * 2 + 0.5
    w = g0.standard_normal(H).astype(np.float32); b = g0.standard_normal(H).astype(np.float32)
    dy = g0.standard_normal((M, H)).astype(np.float32)
    y, xhat, rstd = O.layer_norm(x.astype(np.float64), w, b)
    dx, dw, db = O.layer_norm_bwd(dy.astype(np.float64), xhat, rstd, w)
    xd, wd, bd, dyd = d(x), d(w), d(b), d(dy)
    yd = torch.zeros(M, H, device=DEV); mean = torch.zeros(M, device=DEV); rs = torch.zeros(M, device=DEV)
    for fn in (l.nbci_layernorm_fwd, l.nbci_layernorm_bwd):
        fn.restype = C.c_int
    check(l.nbci_layernorm_fwd(vp(xd), vp(wd), vp(bd), vp(yd), 0, vp(mean), vp(rs), M, H, st()), "ln_fwd")
    dxd = torch.ones(M, H, device=DEV); dwd = torch.zeros(H, device=DEV); dbd = torch.zeros(H, device=DEV)
    check(l.nbci_layernorm_bwd(vp(dyd), vp(xd), vp(wd), vp(mean), vp(rs), vp(dxd), vp(dwd), vp(dbd), M, H, 1, st()), "ln_bwd")
    torch.cuda.synchronize()
    np.testing.assert_allclose(yd.cpu().numpy(), y, atol=2e-5)
    np.testing.assert_allclose(dxd.cpu().numpy(), dx + 1, atol=5e-5)
    np.testing.assert_allclose(dwd.cpu().numpy(), dw, atol=2e-4)
    np.testing.assert_allclose(dbd.cpu().numpy(), db, atol=2e-4)


def _bf(a):
    """float64 value of the bf16 rounding of a (round to nearest even, as the kernels store)"""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().double().numpy()


@pytest.mark.parametrize("H,M,streams", [(1024, 77, "bf16"), (1024, 77, "f32"), (768, 41, "bf16"), (32, 9, "bf16"), (1024, 2300, "bf16")])
def test_layernorm_general_forms_against_f64(H, M, streams):
    """nbci_layernorm_fwd_ex / _bwd_ex (what the NDT1 step launches; residual_dtype f32 / bf16): x, dy, dx_in as STORED (bf16-rounded
    where the stream is bf16) against float64 numpy. Every output is computed in f32 and rounded once, so the bound per element is
    half a bf16 ulp of the exact value (2^-9 relative) for the bf16 outputs plus f32 accumulation noise; the dropout mask of the cast
    copy is the oracle's keep_mask bit for bit; M = 2300 runs several rows per wave (the prefetch loop) and more than one block."""
    l, check = _l()
    g0 = np.random.default_rng(5)
    lp = streams == "bf16"
    x = g0.standard_normal((M, H)) * 2 + 0.5
    dy = g0.standard_normal((M, H)); dxi = g0.standard_normal((M, H))
    w = g0.standard_normal(H).astype(np.float32); b = g0.standard_normal(H).astype(np.float32)
    xs = _bf(x) if lp else x.astype(np.float32).astype(np.float64)
    dys, dxs = _bf(dy), (_bf(dxi) if lp else dxi.astype(np.float32).astype(np.float64))
    y, xhat, rstd = O.layer_norm(xs, w, b)
    dx, dw, db = O.layer_norm_bwd(dys, xhat, rstd, w)
    dx = dx + dxs
    tdt = torch.bfloat16 if lp else torch.float32
    xd, dyd, dxd = d(xs.astype(np.float32)).to(tdt), d(dys.astype(np.float32)).bfloat16(), d(dxs.astype(np.float32)).to(tdt)
    wd, bd = d(w), d(b)
    yd = torch.zeros(M, H, device=DEV, dtype=torch.bfloat16); mean = torch.zeros(M, device=DEV); rs = torch.zeros(M, device=DEV)
    check(l.nbci_layernorm_fwd_ex(vp(xd), int(lp), vp(wd), vp(bd), vp(yd), 1, vp(mean), vp(rs), M, H, st()), "ln_fwd_ex")
    torch.cuda.synchronize()
    yy = yd.double().cpu().numpy()
    assert np.all(np.abs(yy - y) <= 2.0 ** -8 * np.abs(y) + 1e-5)
    np.testing.assert_allclose(rs.cpu().numpy(), rstd.reshape(-1), rtol=2e-5)
    p, seed, site = 0.4, 11, 18
    keep = R.keep_mask(seed, site, M * H, p).reshape(M, H).astype(np.float64)   # multipliers: 1 / (1 - p) or 0
    for dx_in in (dxd, None):
        out = torch.zeros(M, H, device=DEV, dtype=tdt); cast = torch.zeros(M, H, device=DEV, dtype=torch.bfloat16)
        dwd = torch.zeros(H, device=DEV); dbd = torch.zeros(H, device=DEV); cs = torch.zeros(H, device=DEV)
        check(l.nbci_layernorm_bwd_ex(vp(dyd), 1, vp(xd), int(lp), vp(wd), vp(mean), vp(rs), vp(dx_in), vp(out), vp(dwd), vp(dbd), M, H,
                                      vp(cast), 1, p, seed, site, vp(cs), st()), "ln_bwd_ex")
        torch.cuda.synchronize()
        ref = dx if dx_in is not None else dx - dxs
        tol = (2.0 ** -8 * np.abs(ref) if lp else 0) + 2e-4   # (mean / rstd come from the f32 kernel: a few 1e-5 on values of order 1 - 10)
        assert np.all(np.abs(out.double().cpu().numpy() - ref) <= tol)
        masked = ref * keep
        assert np.all(np.abs(cast.double().cpu().numpy() - masked) <= 2.0 ** -8 * np.abs(masked) + 4e-4)
        assert np.array_equal(cast.cpu().float().numpy() != 0, (keep > 0) & (np.abs(out.double().cpu().numpy()) > 1e-30))
        scale = np.sqrt(M)
        np.testing.assert_allclose(dwd.cpu().numpy(), dw, atol=3e-4 * scale)
        np.testing.assert_allclose(dbd.cpu().numpy(), db, atol=3e-4 * scale)
        np.testing.assert_allclose(cs.cpu().numpy(), masked.sum(0), atol=3e-4 * scale)


@pytest.mark.parametrize("Tp,ctx", [(143, (-2, -2)), (37, (3, 2)), (300, (0, -2)), (18, (-1, -1))])
def test_softmax_fwd_bwd_mask_and_dropout(Tp, ctx):
    l, check = _l()
    g0 = np.random.default_rng(2)
    B, nh = 2, 3
    ldS, ldP = (Tp + 3) // 4 * 4, (Tp + 7) // 8 * 8
    S = np.zeros((B, nh, Tp, ldS), np.float32); S[..., :Tp] = g0.standard_normal((B, nh, Tp, Tp))
    tm = (g0.random((B, Tp)) > 0.3).astype(np.int32)
    cm = O.context_mask(ctx[0], ctx[1], max(Tp, 4))[:Tp, :Tp]
    am = (np.eye(Tp, dtype=np.int64)[None] | (cm[None] & tm[:, None, :])).astype(bool)
    s = np.where(am[:, None], S[..., :Tp].astype(np.float64), -np.inf)
    e = np.exp(s - s.max(-1, keepdims=True)); P = e / e.sum(-1, keepdims=True)
    p_drop, seed, site = 0.4, 99, 16
    keep = R.keep_mask(seed, site, B * nh * Tp * Tp, p_drop).reshape(B, nh, Tp, Tp)
    Pd_ = torch.zeros(B, nh, Tp, ldP, device=DEV); Pdd = torch.zeros(B, nh, Tp, ldP, device=DEV)
    for fn in (l.nbci_softmax_fwd, l.nbci_softmax_bwd):
        fn.restype = C.c_int
    Sd, tmd = d(S), d(tm)
    check(l.nbci_softmax_fwd(vp(Sd), vp(Pd_), vp(Pdd), 0, vp(tmd), B, nh, Tp, ldS, ldP, ctx[0], ctx[1], C.c_float(p_drop),
                             seed, site, st()), "softmax_fwd")
    dPd = np.zeros((B, nh, Tp, ldS), np.float32); dPd[..., :Tp] = g0.standard_normal((B, nh, Tp, Tp))
    dS = torch.zeros(B, nh, Tp, ldP, device=DEV)
    dPdd = d(dPd)
    check(l.nbci_softmax_bwd(vp(dPdd), vp(Pd_), vp(dS), 0, B, nh, Tp, ldS, ldP, C.c_float(p_drop), seed, site, st()), "softmax_bwd")
    torch.cuda.synchronize()
    np.testing.assert_allclose(Pd_[..., :Tp].cpu().numpy(), P, atol=2e-6)
    np.testing.assert_allclose(Pdd[..., :Tp].cpu().numpy(), P * keep, atol=5e-6)
    dP = dPd[..., :Tp].astype(np.float64) * keep
    ref = P * (dP - (dP * P).sum(-1, keepdims=True))
    np.testing.assert_allclose(dS[..., :Tp].cpu().numpy(), ref, atol=2e-5)


def test_smooth_noise_matches_oracle():
    l, check = _l()
    g0 = np.random.default_rng(3)
    B, T, N = 3, 70, 24
    x = g0.standard_normal((B, T, N)).astype(np.float32)
    taps = O.gaussian_taps(2).astype(np.float32)
    out = torch.zeros(B, T, N, device=DEV)
    l.nbci_smooth_noise.restype = C.c_int
    xd, tapd = d(x), d(taps)
    check(l.nbci_smooth_noise(vp(xd), vp(out), 0, B, T, N, vp(tapd), len(taps), C.c_float(0.0), C.c_float(0.0), 0, st()), "smooth")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), O.smooth(x, O.gaussian_taps(2)), atol=2e-6)
    check(l.nbci_smooth_noise(vp(xd), vp(out), 0, B, T, N, vp(tapd), len(taps), C.c_float(1.0), C.c_float(0.2), 5, st()), "smooth")
    torch.cuda.synchronize()
    ref = O.smooth(x, O.gaussian_taps(2)) + R.white_noise(5, B, T, N) + 0.2 * R.normal(5, 2, B * N).reshape(B, 1, N)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-4)
    noise = out.cpu().numpy() - O.smooth(x, O.gaussian_taps(2))
    assert abs(noise.std() - np.sqrt(1 + 0.04)) < 0.05 and abs(noise.mean()) < 0.05
