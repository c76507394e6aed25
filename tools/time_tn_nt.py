"""weight-gradient shape, operands row-major-in-k (TN) vs k-major (NT): dW[M][N] = A^T B, K = tokens."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops
K = 9152
dev = "cuda"
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
for M, N in ((1024, 1024), (3072, 1024), (1024, 8192), (8192, 1024)):
    a = (torch.randn(K, M, device=dev) / 8).bfloat16(); b = (torch.randn(K, N, device=dev) / 8).bfloat16()
    at, bt = a.t().contiguous(), b.t().contiguous()
    c = torch.zeros(M, N, device=dev)
    tn = timeit(lambda: ops.gemm(M, N, K, ops.operand(a, M, False), ops.operand(b, N, False), c, N, in_dtype=1, c_dtype=0, beta=1.0))
    nt = timeit(lambda: ops.gemm(M, N, K, ops.operand(at, K, True), ops.operand(bt, K, True), c, N, in_dtype=1, c_dtype=0, beta=1.0))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: TN {tn:7.1f} us ({fl/tn/1e6:6.1f} TF)   NT {nt:7.1f} us ({fl/nt/1e6:6.1f} TF)")
