"""GPU parity of nbci_gemm (include/nbci.h) against float64 matmul on the same inputs.

Integer-valued, ASYMMETRIC operands make the bf16 products exact, so layout mistakes
(transposed fragments, swapped C rows/cols) show up as O(1) errors, not rounding noise.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from llm_bci_amd import ops
    return ops


def _ints(shape, lo=-4, hi=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float()


def _mk(Mx, Kx, kmajor, dtype, seed):
    """logical (rows x K) matrix stored k-major [rows][K] or row-major-in-k [K][rows]."""
    logical = _ints((Mx, Kx), seed=seed)
    # pad leading dim so it is 16-byte aligned but larger than the logical extent
    if kmajor:
        ld = ((Kx + 7) // 8) * 8 + 8
        st = torch.zeros(Mx, ld)
        st[:, :Kx] = logical
    else:
        ld = ((Mx + 7) // 8) * 8 + 8
        st = torch.zeros(Kx, ld)
        st[:, :Mx] = logical.t()
    # poison the padding: the kernel must mask it, not multiply it
    if kmajor:
        st[:, Kx:] = 1000.0
    else:
        st[:, Mx:] = 1000.0
    return logical.double(), st.to(DEV, dtype), ld


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("ak", [True, False])
@pytest.mark.parametrize("bk", [True, False])
@pytest.mark.parametrize("shape", [(143, 41, 143), (300, 200, 136), (128, 128, 64), (257, 129, 70)])
def test_layouts_exact(dtype, ak, bk, shape):
    ops = _ops()
    M, N, K = shape
    a_log, a_st, lda = _mk(M, K, ak, dtype, 1)
    b_log, b_st, ldb = _mk(N, K, bk, dtype, 2)
    ldc = ((N + 3) // 4) * 4 + 4
    out = torch.full((M, ldc), -7.0, device=DEV)
    ops.gemm(M, N, K, ops.operand(a_st, lda, ak), ops.operand(b_st, ldb, bk), out, ldc,
             in_dtype=ops._dt(a_st), c_dtype=ops.NBCI_F32)
    torch.cuda.synchronize()
    ref = (a_log @ b_log.t())
    got = out[:, :N].double().cpu()
    assert torch.equal(got, ref), f"max err {(got - ref).abs().max()}"
    assert torch.all(out[:, N:] == -7.0), "wrote outside N"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_unaligned_scalar_path(dtype):
    """ld not a multiple of 16 bytes -> scalar loader + scalar epilogue."""
    ops = _ops()
    M, N, K = 70, 41, 45
    a = _ints((M, K), seed=3)
    b = _ints((N, K), seed=4)
    out = torch.zeros(M, N, device=DEV)
    ad, bd = a.to(DEV, dtype), b.to(DEV, dtype)
    ops.gemm(M, N, K, ops.operand(ad, K, True), ops.operand(bd, K, True), out, N,
             in_dtype=ops._dt(ad), c_dtype=ops.NBCI_F32)
    torch.cuda.synchronize()
    assert torch.equal(out.double().cpu(), a.double() @ b.double().t())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_window_view_forward_and_wgrad(dtype):
    """nn.Unfold((S,D),stride) + Linear == GEMM over an overlapping-row view (ndt1.py:138,180)."""
    ops = _ops()
    Bn, T, D, S, stride, H = 3, 50, 16, 8, 4, 40
    Tp = 1 + (T - S) // stride
    y = _ints((Bn, T, D), seed=5)
    w = _ints((H, S * D), seed=6)
    yd, wd = y.to(DEV, dtype), w.to(DEV, dtype)
    out = torch.zeros(Bn * Tp, H, device=DEV)
    A = ops.operand(yd, stride * D, True, rpb=Tp, gstride=T * D)
    ops.gemm(Bn * Tp, H, S * D, A, ops.operand(wd, S * D, True), out, H,
             in_dtype=ops._dt(yd), c_dtype=ops.NBCI_F32)
    torch.cuda.synchronize()
    win = torch.stack([y[:, j * stride:j * stride + S, :].reshape(Bn, S * D) for j in range(Tp)], 1)  # B,Tp,S*D
    ref = win.double().reshape(Bn * Tp, S * D) @ w.double().t()
    assert torch.equal(out.double().cpu(), ref)
    # weight grad: dW[h][k] = sum_rows dx[row][h] * win[row][k]  (both operands row-major-in-k)
    dx = _ints((Bn * Tp, H), seed=7)
    dxd = dx.to(DEV, dtype)
    dW = torch.zeros(H, S * D, device=DEV)
    Bop = ops.operand(yd, stride * D, False, rpb=Tp, gstride=T * D)
    ops.gemm(H, S * D, Bn * Tp, ops.operand(dxd, H, False), Bop, dW, S * D,
             in_dtype=ops._dt(yd), c_dtype=ops.NBCI_F32)
    torch.cuda.synchronize()
    refw = dx.double().t() @ win.double().reshape(Bn * Tp, S * D)
    assert torch.equal(dW.double().cpu(), refw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_batched_heads(dtype):
    """two-level batch (b, head) over a packed qkv buffer, as attention uses it."""
    ops = _ops()
    Bn, Tq, nh, hd = 2, 37, 3, 32
    H = nh * hd
    qkv = _ints((Bn, Tq, 3 * H), lo=-2, hi=3, seed=8)
    qd = qkv.to(DEV, dtype)
    ldS = 40
    S = torch.zeros(Bn, nh, Tq, ldS, device=DEV)
    A = ops.operand(qd, 3 * H, True, zs1=Tq * 3 * H, zs2=hd)
    Bo = ops.operand(qd, 3 * H, True, zs1=Tq * 3 * H, zs2=hd, offset=H)
    ops.gemm(Tq, Tq, hd, A, Bo, S, ldS, in_dtype=ops._dt(qd), c_dtype=ops.NBCI_F32,
             batch=Bn * nh, zdiv=nh, czs1=nh * Tq * ldS, czs2=Tq * ldS, alpha=0.5)
    torch.cuda.synchronize()
    q = qkv[..., :H].reshape(Bn, Tq, nh, hd).permute(0, 2, 1, 3).double()
    k = qkv[..., H:2 * H].reshape(Bn, Tq, nh, hd).permute(0, 2, 1, 3).double()
    ref = 0.5 * q @ k.transpose(-1, -2)
    assert torch.equal(S[..., :Tq].double().cpu(), ref)


def test_splitk_matches():
    ops = _ops()
    M, N, K = 130, 96, 1000
    a = _ints((K, M), seed=9)   # stored [K][M]
    b = _ints((K, N), seed=10)
    for dtype in (torch.bfloat16, torch.float32):
        ad, bd = a.to(DEV, dtype), b.to(DEV, dtype)
        out = torch.zeros(M, N, device=DEV)
        ops.gemm(M, N, K, ops.operand(ad, M, False), ops.operand(bd, N, False), out, N,
                 in_dtype=ops._dt(ad), c_dtype=ops.NBCI_F32, splitk=5)
        torch.cuda.synchronize()
        assert torch.equal(out.double().cpu(), a.double().t() @ b.double())


def test_epilogue_bias_act_residual_c2_beta_bf16out():
    ops = _ops()
    torch.manual_seed(0)
    M, N, K = 100, 72, 64
    x = torch.randn(M, K)
    w = torch.randn(N, K) * 0.2
    bias = torch.randn(N)
    res = torch.randn(M, N)
    xd, wd, bd, rd = x.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV)
    pre = x.double() @ w.double().t() + bias.double()
    for act, fn in ((1, lambda v: v / (1 + v.abs())), (2, lambda v: torch.nn.functional.gelu(v)), (3, torch.relu)):
        out = torch.zeros(M, N, device=DEV)
        c2 = torch.zeros(M, N, device=DEV)
        ops.gemm(M, N, K, ops.operand(xd, K, True), ops.operand(wd, K, True), out, N, in_dtype=ops.NBCI_F32,
                 c_dtype=ops.NBCI_F32, bias=bd, act=act, residual=rd, ldr=N, C2=c2)
        torch.cuda.synchronize()
        assert torch.allclose(c2.double().cpu(), pre, atol=1e-4)
        assert torch.allclose(out.double().cpu(), fn(pre) + res.double(), atol=1e-4)
    # beta accumulate
    out = torch.ones(M, N, device=DEV)
    ops.gemm(M, N, K, ops.operand(xd, K, True), ops.operand(wd, K, True), out, N, in_dtype=ops.NBCI_F32,
             c_dtype=ops.NBCI_F32, beta=1.0, alpha=2.0)
    torch.cuda.synchronize()
    assert torch.allclose(out.double().cpu(), 2 * (x.double() @ w.double().t()) + 1, atol=1e-4)
    # bf16 in / bf16 out
    xb, wb = xd.bfloat16(), wd.bfloat16()
    outb = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(M, N, K, ops.operand(xb, K, True), ops.operand(wb, K, True), outb, N, in_dtype=ops.NBCI_BF16,
             c_dtype=ops.NBCI_BF16, bias=bd)
    torch.cuda.synchronize()
    refb = xb.double().cpu() @ wb.double().cpu().t() + bias.double()
    assert torch.allclose(outb.double().cpu(), refb, atol=0.05, rtol=0.01)


def test_epilogue_dropout_statistics_and_determinism():
    ops = _ops()
    M, N, K = 512, 256, 64
    xd = torch.ones(M, K, device=DEV)
    wd = torch.ones(N, K, device=DEV) / K
    outs = []
    for _ in range(2):
        out = torch.zeros(M, N, device=DEV)
        ops.gemm(M, N, K, ops.operand(xd, K, True), ops.operand(wd, K, True), out, N, in_dtype=ops.NBCI_F32,
                 c_dtype=ops.NBCI_F32, drop_p=0.4, seed=123, site=7)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    kept = (outs[0] != 0)
    assert abs(kept.float().mean().item() - 0.6) < 0.01
    assert torch.allclose(outs[0][kept], torch.full_like(outs[0][kept], 1 / 0.6), atol=1e-5)
    out2 = torch.zeros(M, N, device=DEV)
    ops.gemm(M, N, K, ops.operand(xd, K, True), ops.operand(wd, K, True), out2, N, in_dtype=ops.NBCI_F32,
             c_dtype=ops.NBCI_F32, drop_p=0.4, seed=124, site=7)
    torch.cuda.synchronize()
    assert not torch.equal(out2.cpu(), outs[0])


def test_error_reporting():
    ops = _ops()
    from llm_bci_amd._lib import NbciError
    x = torch.zeros(4, 4, device=DEV)
    with pytest.raises(NbciError):
        ops.gemm(0, 4, 4, ops.operand(x, 4, True), ops.operand(x, 4, True), x, 4, in_dtype=0, c_dtype=0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_big_random_tolerance(dtype):
    """C2-sized projection with random data: f32 path <= 1e-3 abs, bf16 path within bf16 rounding."""
    ops = _ops()
    torch.manual_seed(1)
    M, N, K = 1144, 1024, 1024
    x = torch.randn(M, K)
    w = torch.randn(N, K) / 32
    xd, wd = x.to(DEV, dtype), w.to(DEV, dtype)
    out = torch.zeros(M, N, device=DEV)
    ops.gemm(M, N, K, ops.operand(xd, K, True), ops.operand(wd, K, True), out, N, in_dtype=ops._dt(xd),
             c_dtype=ops.NBCI_F32)
    torch.cuda.synchronize()
    ref = xd.double().cpu() @ wd.double().cpu().t()
    err = (out.double().cpu() - ref).abs().max().item()
    assert err < (1e-3 if dtype == torch.float32 else 1e-2), err


@pytest.mark.parametrize("bk", [True, False])
@pytest.mark.parametrize("shape", [(9152, 1024, 1024), (9000, 640, 192), (7000, 1024, 256)])
def test_three_stage_288_tile_kernel_exact(bk, shape):
    """shapes that select the 3-stage 288 x 128 kernel (counted vmcnt pipeline): exact integer data,
    repeated 3 times to catch staging races."""
    ops = _ops()
    M, N, K = shape
    a = _ints((M, K), lo=-2, hi=3, seed=11)
    b = _ints((N, K), lo=-2, hi=3, seed=12)
    ad = a.to(DEV, torch.bfloat16)
    bd = (b if bk else b.t().contiguous()).to(DEV, torch.bfloat16)
    ref = (a.double() @ b.double().t())
    for _ in range(3):
        out = torch.zeros(M, N, device=DEV)
        ops.gemm(M, N, K, ops.operand(ad, K, True), ops.operand(bd, bd.stride(0), bk), out, N, in_dtype=ops.NBCI_BF16,
                 c_dtype=ops.NBCI_F32)
        torch.cuda.synchronize()
        assert torch.equal(out.double().cpu(), ref)


def test_grouped_gemm_matches_individual():
    ops = _ops()
    import ctypes as C
    from llm_bci_amd._lib import GemmDesc, check, lib
    K = 1000
    probs = [(256, 128), (128, 384), (130, 128)]
    descs = (GemmDesc * len(probs))()
    keep, refs, outs = [], [], []
    for i, (M, N) in enumerate(probs):
        a = _ints((K, M), lo=-2, hi=3, seed=20 + i); b = _ints((K, N), lo=-2, hi=3, seed=30 + i)
        ad, bd = a.to(DEV, torch.bfloat16), b.to(DEV, torch.bfloat16)
        out = torch.ones(M, N, device=DEV)
        d = descs[i]
        d.M, d.N, d.K, d.in_dtype = M, N, K, ops.NBCI_BF16
        d.A, d.B = ops.operand(ad, M, False), ops.operand(bd, N, False)
        d.C, d.ldc, d.c_dtype, d.batch, d.zdiv, d.splitk, d.alpha, d.beta = out.data_ptr(), N, ops.NBCI_F32, 1, 1, 1, 1.0, 1.0
        keep += [ad, bd]; outs.append(out); refs.append(a.double().t() @ b.double() + 1)
    check(lib().nbci_gemm_grouped(descs, len(probs), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "grouped")
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert torch.equal(o.double().cpu(), r)


@pytest.mark.parametrize("ak", [True, False])
@pytest.mark.parametrize("bk", [True, False])
@pytest.mark.parametrize("shape", [(300, 256, 512), (1144, 1024, 1024), (128, 128, 256), (700, 384, 4096)])
def test_multistage_small_grid_kernel_exact(ak, bk, shape):
    """<= 256 workgroups and K % 64 == 0 selects the 4-stage counted-vmcnt kernel: exact integer data,
    all four layouts, repeated to catch staging races."""
    ops = _ops()
    M, N, K = shape
    a = _ints((M, K), lo=-2, hi=3, seed=41)
    b = _ints((N, K), lo=-2, hi=3, seed=42)
    ad = (a if ak else a.t().contiguous()).to(DEV, torch.bfloat16)
    bd = (b if bk else b.t().contiguous()).to(DEV, torch.bfloat16)
    ref = a.double() @ b.double().t()
    for _ in range(3):
        out = torch.zeros(M, N, device=DEV)
        ops.gemm(M, N, K, ops.operand(ad, ad.stride(0), ak), ops.operand(bd, bd.stride(0), bk), out, N,
                 in_dtype=ops.NBCI_BF16, c_dtype=ops.NBCI_F32)
        torch.cuda.synchronize()
        assert torch.equal(out.double().cpu(), ref)


@pytest.mark.parametrize("shape", [(3, 21, 64, 16), (64, 143, 1024, 256)])   # (B, T', H, D): small -> 128-row tiles, real -> 160-row tiles
def test_phase_gemm_views_match_col2im(shape):
    """The embedder backward as ONE GEMM: overlapping-row k-major view of zero-padded sample blocks
    x reversed weight slices (row-major-in-k view with a NEGATIVE group stride), gate = softsign' from the stored
    OUTPUT. Exact integer data vs the explicit window gradient + col2im it replaces (reference ndt1.py:138-140,180)."""
    ops = _ops()
    import ctypes as C
    from llm_bci_amd._lib import GemmDesc, check, lib
    B, Tp, H, D = shape
    st, nwin = 4, 8
    size = st * nwin
    T = st * (Tp - 1) + size
    Q, npad = T // st, nwin - 1
    P = Q + npad
    dx0 = _ints((B, Tp, H), lo=-3, hi=4, seed=51).to(DEV)
    W = _ints((H, size * D), lo=-2, hi=3, seed=52).to(DEV)
    y = (_ints((B, T, D), lo=-7, hi=8, seed=53) / 8.0).to(DEV)          # activation outputs in (-1, 1), exact in bf16
    # reference: dwin = dx0 W, fold the windows back onto the bins, times (1 - |y|)^2
    dwin = (dx0.reshape(B * Tp, H) @ W).reshape(B, Tp, size * D)
    dpre = torch.zeros(B, T, D, device=DEV)
    for r in range(size):
        dpre[:, r: r + st * Tp: st, :] += dwin[:, :, r * D:(r + 1) * D]
    ref = dpre * (1.0 - y.abs()) ** 2
    pad = torch.zeros(B, P, H, device=DEV, dtype=torch.bfloat16)
    pad[:, npad:npad + Tp] = dx0.bfloat16()
    Wb, yb = W.bfloat16().contiguous(), y.bfloat16().contiguous()
    out = torch.full((B * Q, st * D), -7.0, device=DEV)
    d = GemmDesc()
    d.M, d.N, d.K, d.in_dtype = B * Q, st * D, nwin * H, ops.NBCI_BF16
    d.A = ops.operand(pad, H, True, rpb=Q, gstride=P * H)
    d.B = ops.operand(Wb, size * D, False, rpb=H, gstride=-D * st, offset=D * st * (nwin - 1))
    d.C, d.ldc, d.c_dtype, d.batch, d.zdiv, d.splitk, d.alpha, d.beta = out.data_ptr(), st * D, ops.NBCI_F32, 1, 1, 1, 1.0, 0.0
    d.gate, d.ldg, d.gate_act = yb.data_ptr(), st * D, 64 + 1   # 64 + NBCI_ACT_SOFTSIGN: derivative from the output
    for _ in range(2):
        check(lib().nbci_gemm(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "phase gemm")
    torch.cuda.synchronize()
    assert torch.equal(out.reshape(B, T, D).double().cpu(), ref.double().cpu())


@pytest.mark.parametrize("shape", [(3, 21, 64, 16), (16, 143, 1024, 256)])   # (B, T', H, D)
def test_phase_layout_plain_operands_match_the_views(shape):
    """The step's embedder backward since round 3 (ndt1.hip, carve): dx0 as npad zero rows + per sample Q rows (T' tokens + npad zero
    rows), every row one H apart -> the phase GEMM's A is a PLAIN k-major operand whose rows overlap (ld < K), and the stack-projection
    weight gradient runs over ALL B * Q rows with two plain operands (the window rows of y overlap: rpb = -1; the zero rows of dx0 meet
    window rows that run into the next sample / the zeroed slack behind y). Exact integer data against the explicit windows."""
    ops = _ops()
    import ctypes as C
    from llm_bci_amd._lib import GemmDesc, check, lib
    B, Tp, H, D = shape
    st, nwin = 4, 8
    size = st * nwin
    T = st * (Tp - 1) + size
    Q, npad = T // st, nwin - 1
    assert Q - Tp == npad
    dx0 = _ints((B, Tp, H), lo=-3, hi=4, seed=61).to(DEV)
    W = _ints((H, size * D), lo=-2, hi=3, seed=62).to(DEV)
    y = (_ints((B, T, D), lo=-7, hi=8, seed=63) / 8.0).to(DEV)
    lay = torch.zeros(npad + B * Q, H, device=DEV, dtype=torch.bfloat16)
    lay[npad:].view(B, Q, H)[:, :Tp] = dx0.bfloat16()
    ybuf = torch.zeros(B * T * D + (size - st) * D, device=DEV, dtype=torch.bfloat16)   # y + its zeroed slack
    ybuf[:B * T * D] = y.bfloat16().reshape(-1)
    Wb = W.bfloat16().contiguous()
    # (1) phase GEMM: d pre-activation rows (b, q) = bins st*q .. of sample b
    dwin = (dx0.reshape(B * Tp, H) @ W).reshape(B, Tp, size * D)
    dpre = torch.zeros(B, T, D, device=DEV)
    for r in range(size):
        dpre[:, r: r + st * Tp: st, :] += dwin[:, :, r * D:(r + 1) * D]
    out = torch.full((B * Q, st * D), -7.0, device=DEV)
    d = GemmDesc()
    d.M, d.N, d.K, d.in_dtype = B * Q, st * D, nwin * H, ops.NBCI_BF16
    d.A = ops.operand(lay, H, True)
    d.B = ops.operand(Wb, size * D, False, rpb=H, gstride=-D * st, offset=D * st * (nwin - 1))
    d.C, d.ldc, d.c_dtype, d.batch, d.zdiv, d.splitk, d.alpha, d.beta = out.data_ptr(), st * D, ops.NBCI_F32, 1, 1, 1, 1.0, 0.0
    check(lib().nbci_gemm(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "phase gemm")
    torch.cuda.synchronize()
    assert torch.equal(out.reshape(B, T, D).double().cpu(), dpre.double().cpu())
    # (2) stack-projection weight gradient: dW[h][k] = sum over tokens dx0[b, j, h] * window(b, j)[k]
    win = torch.stack([y[:, j * st:j * st + size, :].reshape(B, size * D) for j in range(Tp)], 1)
    refw = dx0.reshape(B * Tp, H).double().t() @ win.reshape(B * Tp, size * D).double()
    dW = torch.zeros(H, size * D, device=DEV)
    ops.gemm(H, size * D, B * Q, ops.operand(lay, H, False, offset=npad * H), ops.operand(ybuf, st * D, False, rpb=-1), dW, size * D,
             in_dtype=ops.NBCI_BF16, c_dtype=ops.NBCI_F32, beta=1.0)
    torch.cuda.synchronize()
    assert torch.equal(dW.double().cpu(), refw.cpu())


@pytest.mark.parametrize("M,residual", [(9152, torch.bfloat16), (2000, torch.float32), (37, torch.bfloat16)])
def test_mlp_strip_prototype_equals_the_two_gemm_launches(M, residual):
    """csrc/mlp_strip.hip (round 4 prototype, nbci_debug_mlp_strip): up projection + GELU (+ act' copy) and down projection + dropout + residual of
    models/ndt1.py:224-227,328 in ONE launch with the row strip resident on its CU - bit-equal to the two nbci_gemm launches it would replace
    (same K order, same rounding of g, same dropout counters). It is slower than they are (profiles/r04_mlp_strip_*.txt) and not on the step's path."""
    import ctypes as C
    from llm_bci_amd._lib import check, lib
    ops = _ops()
    H = I = 1024
    torch.manual_seed(3)
    h = torch.randn(M, H, device=DEV).bfloat16(); x = torch.randn(M, H, device=DEV).to(residual)
    Wu = (torch.randn(I, H, device=DEV) / 32).bfloat16(); bu = torch.randn(I, device=DEV) * 0.1
    Wd = (torch.randn(H, I, device=DEV) / 32).bfloat16(); bd = torch.randn(H, device=DEV) * 0.1
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for fused in (False, True):
        g = torch.zeros(M, I, device=DEV, dtype=torch.bfloat16); da = torch.zeros_like(g); y = torch.zeros(M, H, device=DEV, dtype=residual)
        up = ops.gemm_desc(M, I, H, ops.operand(h, H, True), ops.operand(Wu, H, True), g, I, in_dtype=1, c_dtype=1, bias=bu, act=2, C2=da, c2_grad=1)
        dn = ops.gemm_desc(M, H, I, ops.operand(g, I, True), ops.operand(Wd, I, True), y, H, in_dtype=1, c_dtype=0 if residual == torch.float32 else 1,
                           bias=bd, drop_p=0.4, seed=7, site=19, residual=x, ldr=H)
        if fused:
            check(lib().nbci_debug_mlp_strip(C.byref(up), C.byref(dn), st), "mlp_strip")
        else:
            check(lib().nbci_gemm(C.byref(up), st), "up"); check(lib().nbci_gemm(C.byref(dn), st), "down")
        torch.cuda.synchronize()
        outs.append((g, da, y))
    for a, b, nm in zip(outs[0], outs[1], ("g", "act'", "y")):
        assert torch.equal(a, b), nm
    assert outs[0][2].abs().sum() > 0
