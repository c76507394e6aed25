"""Thin Python wrappers over the C-ABI: torch tensors in, raw pointers + current stream out.

PyTorch is used for device memory and streams only; all arithmetic happens in libnbci.so.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import NBCI_BF16, NBCI_F32, GemmDesc, Operand, check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dt(t):
    if t.dtype == torch.float32:
        return NBCI_F32
    if t.dtype == torch.bfloat16:
        return NBCI_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.NbciUnavailable("nbci ops need CUDA/HIP tensors; there is no CPU fallback")


def operand(t, ld, kmajor, rpb=0, gstride=0, zs1=0, zs2=0, offset=0):
    """Describe a GEMM operand living inside tensor `t` (element offset `offset`)."""
    return Operand(t.data_ptr() + offset * t.element_size(), ld, 1 if kmajor else 0, rpb, gstride, zs1, zs2)


def gemm_desc(M, N, K, A, B, Cout, ldc, *, in_dtype, c_dtype, C2=None, czs1=0, czs2=0, batch=1, zdiv=1,
              splitk=1, alpha=1.0, beta=0.0, bias=None, act=0, drop_p=0.0, seed=0, site=0,
              residual=None, ldr=0, c_offset=0, c2_grad=0, gate=None, ldg=0, gate_act=0, colsum=None):
    """the nbci_gemm_desc of C = alpha * A . B^T with the fused epilogue of include/nbci.h"""
    d = GemmDesc()
    d.M, d.N, d.K, d.in_dtype = M, N, K, in_dtype
    d.A, d.B = A, B
    d.C = Cout.data_ptr() + c_offset * Cout.element_size()
    d.C2 = C2.data_ptr() if C2 is not None else None
    d.ldc, d.czs1, d.czs2, d.c_dtype = ldc, czs1, czs2, c_dtype
    d.batch, d.zdiv, d.splitk = batch, zdiv, splitk
    d.alpha, d.beta = alpha, beta
    d.bias = bias.data_ptr() if bias is not None else None
    d.act, d.drop_p, d.seed, d.site = act, drop_p, seed, site
    d.residual = residual.data_ptr() if residual is not None else None
    d.ldr = ldr
    d.residual_dtype = 1 if (residual is not None and residual.dtype == torch.bfloat16) else 0   # NBCI_BF16 / NBCI_F32
    d.c2_grad = c2_grad
    if gate is not None:    # result *= act'(gate); gate_act < 0: `gate` already holds act' (the forward's C2 with c2_grad=1)
        d.gate, d.ldg, d.gate_act = gate.data_ptr(), ldg, gate_act
    if colsum is not None:  # f32 [N] += column sums of the stored C (bias gradient fused into the GEMM producing the activation gradient)
        d.colsum = colsum.data_ptr()
    return d


def gemm(M, N, K, A, B, Cout, ldc, **kw):
    """C = alpha * A . B^T with the fused epilogue of include/nbci.h (nbci_gemm)."""
    d = gemm_desc(M, N, K, A, B, Cout, ldc, **kw)
    check(lib().nbci_gemm(C.byref(d), _stream()), "nbci_gemm")


def linear_nt(x, w, out=None, **kw):
    """y[M,N] = x[M,K] . w[N,K]^T for contiguous 2-D tensors (test/helper convenience)."""
    _need_cuda(x, w)
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=kw.pop("out_dtype", torch.float32))
    gemm(M, N, K, operand(x, x.stride(0), True), operand(w, w.stride(0), True), out, out.stride(0),
         in_dtype=_dt(x), c_dtype=_dt(out), **kw)
    return out
