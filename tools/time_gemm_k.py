"""fixed vs per-K-tile cost: time the 9152 x N x K NT GEMM (bf16 out, no epilogue extras) over K."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 9152
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = "cuda"
for K in (256, 512, 1024, 2048, 4096, 8192):
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) / 32).bfloat16()
    cb = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    A, B = ops.operand(a, K, True), ops.operand(b, K, True)
    f = lambda: ops.gemm(M, N, K, A, B, cb, N, in_dtype=1, c_dtype=1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"M={M} N={N} K={K:5d} {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")
