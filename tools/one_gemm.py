"""Run ONE GEMM shape repeatedly (for rocprofv3 --pmc runs). args: M N K ak bk iters"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops
M, N, K, ak, bk, it = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
a = (torch.randn(M, K, device="cuda") if ak else torch.randn(K, M, device="cuda")).bfloat16()
b = (torch.randn(N, K, device="cuda") if bk else torch.randn(K, N, device="cuda")).bfloat16()
c = torch.zeros(M, N, device="cuda")
for _ in range(it):
    ops.gemm(M, N, K, ops.operand(a, a.stride(0), ak), ops.operand(b, b.stride(0), bk), c, N, in_dtype=1, c_dtype=0)
torch.cuda.synchronize()
