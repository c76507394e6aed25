"""time one GEMM shape: args M N K ak bk"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops
M, N, K, ak, bk = [int(x) for x in sys.argv[1:6]]
a = (torch.randn(M, K, device="cuda") if ak else torch.randn(K, M, device="cuda")).bfloat16()
b = (torch.randn(N, K, device="cuda") if bk else torch.randn(K, N, device="cuda")).bfloat16()
c = torch.zeros(M, N, device="cuda")
f = lambda: ops.gemm(M, N, K, ops.operand(a, a.stride(0), ak), ops.operand(b, b.stride(0), bk), c, N, in_dtype=1, c_dtype=0)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('NBCI_LIB','default').split('/')[-1]:>14} M{M} N{N} K{K}: {e0.elapsed_time(e1)/20*1e3:8.1f} us")
