"""The embedder's stack-projection weight gradient alone (dW[H][size*D] = dx0^T . windows, K = B*T' tokens): plain operands against the step's
views (dx0 in zero-padded sample blocks; the windows as an overlapping-row view of y).   python tools/time_stack_wgrad.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops
dev = "cuda"
Bn, T, D, size, st, H = 64, 600, 256, 32, 4, 1024
Tk = 1 + (T - size) // st; Q = T // st; npad = size // st - 1; P = Q + npad
K, N = Bn * Tk, size * D
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
a_plain = (torch.randn(K, H, device=dev) / 8).bfloat16()
b_plain = (torch.randn(K, N, device=dev) / 8).bfloat16()
a_pad = (torch.randn(Bn * P, H, device=dev) / 8).bfloat16()
y = (torch.randn(Bn * T, D, device=dev) / 8).bfloat16()
c = torch.zeros(H, N, device=dev)
A_plain = ops.operand(a_plain, H, False)
A_view = ops.operand(a_pad, H, False, rpb=Tk, gstride=P * H, offset=npad * H)
B_plain = ops.operand(b_plain, N, False)
B_view = ops.operand(y, st * D, False, rpb=Tk, gstride=T * D)
fl = 2.0 * H * N * K
for name, A, B in (("plain A, plain B", A_plain, B_plain), ("padded-block A, plain B", A_view, B_plain), ("plain A, window-view B", A_plain, B_view),
                   ("padded-block A, window-view B (the step)", A_view, B_view)):
    us = timeit(lambda: ops.gemm(H, N, K, A, B, c, N, in_dtype=1, c_dtype=0, beta=1.0))
    print(f"{name:45s} {us:7.1f} us  {fl / us / 1e6:7.1f} TF")
