"""GPU parity of the producer / consumer bf16 GEMM kernel (llm_bci_amd/csrc/gemm_pc.hip: 144 x 256 tiles, 4 consumer + 4 producer
waves, 3 LDS stages) through nbci_gemm, forced with nbci_debug_gemm_pc(2). Integer-valued operands make every bf16 product and
f32 sum exact, so a wrong fragment, a stale LDS stage (a synchronisation slip) or a swapped tile shows up as an O(1) error.
Shapes cover: ragged M / N edges, K = 2 .. 64 tiles (pipeline prologue / steady state / drain), both B layouts, batches, and the
fused epilogues the train step uses; every case is also compared with the two-workgroup-per-CU kernels on the same inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def pc():
    from llm_bci_amd._lib import lib
    l = lib()

    def set_mode(m):
        assert l.nbci_debug_gemm_pc(m) == 0
    yield set_mode
    set_mode(1)


def _ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float()


def _operands(M, N, K, bk, seed):
    a = _ints((M, K), -3, 4, seed)
    b = _ints((N, K), -3, 4, seed + 1)
    a_st = a.to(DEV, torch.bfloat16)
    b_st = (b if bk else b.t().contiguous()).to(DEV, torch.bfloat16)
    return a, b, a_st, b_st


@pytest.mark.parametrize("bk", [True, False])
@pytest.mark.parametrize("M,N,K", [(144, 256, 128), (9152, 1024, 1024), (1000, 520, 192), (143, 256, 64 * 7), (300, 1032, 4096), (2891, 768, 320)])
def test_pc_kernel_exact_and_equal_to_reference_kernels(pc, bk, M, N, K):
    from llm_bci_amd import ops
    a, b, a_st, b_st = _operands(M, N, K, bk, 3)
    ref = (a.double() @ b.double().t())
    outs = []
    for mode in (2, 0):
        pc(mode)
        out = torch.full((M, N), -7.0, device=DEV)
        ops.gemm(M, N, K, ops.operand(a_st, K, True), ops.operand(b_st, K if bk else N, bk), out, N, in_dtype=ops.NBCI_BF16, c_dtype=ops.NBCI_F32)
        torch.cuda.synchronize()
        outs.append(out.double().cpu())
    assert torch.equal(outs[0], ref), (outs[0] - ref).abs().max()      # exact
    assert torch.equal(outs[0], outs[1])


def test_pc_kernel_repeated_launches_are_race_free(pc):
    """the same problem 40 times with different data each time: a stale stage or an early read would make some tile wrong once in a while"""
    from llm_bci_amd import ops
    pc(2)
    M, N, K = 9152, 1024, 1024
    for it in range(40):
        a, b, a_st, b_st = _operands(M, N, K, it % 2 == 0, 100 + it)
        out = torch.empty(M, N, device=DEV)
        ops.gemm(M, N, K, ops.operand(a_st, K, True), ops.operand(b_st, K if it % 2 == 0 else N, it % 2 == 0), out, N,
                 in_dtype=ops.NBCI_BF16, c_dtype=ops.NBCI_F32)
        ref = a.to(DEV) @ b.to(DEV).t()          # integer data: the f32 library matmul is exact too
        assert torch.equal(out, ref), it


def test_pc_kernel_epilogues_match_reference_kernels(pc):
    """bias + GELU with stored act', bias + dropout + residual (f32 stream), gate multiply + column sums, bf16 / f32 outputs, batch"""
    from llm_bci_amd import ops
    M, N, K = 1300, 512, 256
    g = torch.Generator().manual_seed(5)
    a = (torch.randn(M, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.1).to(DEV, torch.bfloat16)
    wt = w.t().contiguous()
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    gate = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)

    def run(mode):
        pc(mode)
        o = {}
        y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16); d = torch.empty_like(y)
        ops.gemm(M, N, K, ops.operand(a, K, True), ops.operand(w, K, True), y, N, in_dtype=1, c_dtype=1, bias=bias, act=2, C2=d, c2_grad=1)
        o["gelu"], o["dgelu"] = y, d
        r = torch.empty(M, N, device=DEV)
        ops.gemm(M, N, K, ops.operand(a, K, True), ops.operand(w, K, True), r, N, in_dtype=1, c_dtype=0, bias=bias, drop_p=0.4, seed=11, site=3,
                 residual=res, ldr=N)
        o["drop_res"] = r
        gg = torch.empty(M, N, device=DEV, dtype=torch.bfloat16); cs = torch.zeros(N, device=DEV)
        ops.gemm(M, N, K, ops.operand(a, K, True), ops.operand(wt, N, False), gg, N, in_dtype=1, c_dtype=1, gate=gate, ldg=N, gate_act=-1, colsum=cs)
        o["gated"], o["colsum"] = gg, cs
        ab = (torch.randn(3, 200, K, generator=torch.Generator().manual_seed(9)) * 0.5).to(DEV, torch.bfloat16)
        yb = torch.empty(3, 200, N, device=DEV)
        ops.gemm(200, N, K, ops.operand(ab, K, True, zs1=200 * K), ops.operand(w, K, True), yb, N, in_dtype=1, c_dtype=0, batch=3, czs1=200 * N)
        o["batched"] = yb
        torch.cuda.synchronize()
        return o
    new, old = run(2), run(0)
    for k in new:
        if k == "colsum":   # atomics: order differs
            assert torch.allclose(new[k], old[k], rtol=1e-4, atol=1e-3), k
        else:
            assert torch.equal(new[k], old[k]), k
    ref = torch.nn.functional.gelu(a.float() @ w.float().t() + bias)
    assert (new["gelu"].float() - ref).abs().max() < 0.02
