"""The iTransformer MLP data gradient alone: du = (c W_2) * relu'(g) * keep with the linear1 bias gradient as column sums, 24 016 x 3072 x 768
(six such launches per step, 220 us each): which part of the epilogue costs what.   python tools/time_itr_du.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops
M, N, K = 24016, 3072, 768
dev = "cuda"
a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(K, N, device=dev) / 32).bfloat16()   # W_2 as [K][N]: row-major-in-k B
g = torch.relu(torch.randn(M, N, device=dev)).bfloat16()
cb = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); cs = torch.zeros(N, device=dev)
A, B = ops.operand(a, K, True), ops.operand(w, N, False)
cases = {
    "plain bf16 out": lambda: ops.gemm(M, N, K, A, B, cb, N, in_dtype=1, c_dtype=1),
    "+ gate relu (from g)": lambda: ops.gemm(M, N, K, A, B, cb, N, in_dtype=1, c_dtype=1, gate=g, ldg=N, gate_act=3),
    "+ gate + dropout": lambda: ops.gemm(M, N, K, A, B, cb, N, in_dtype=1, c_dtype=1, gate=g, ldg=N, gate_act=3, drop_p=0.4, seed=1, site=2),
    "+ gate + dropout + colsum (the step's)": lambda: ops.gemm(M, N, K, A, B, cb, N, in_dtype=1, c_dtype=1, gate=g, ldg=N, gate_act=3, drop_p=0.4, seed=1, site=2, colsum=cs),
}
fl = 2.0 * M * N * K
for name, f in cases.items():
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:42s} {us:7.1f} us  {fl / us / 1e6:6.1f} TF")
