"""the step's forward NT shapes (bf16 out), one line per shape: for in-box A/B of env switches (NBCI_GEMM_STAGGER, ...)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops
M = 9152
dev = "cuda"
out = []
for N, K in ((1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096), (1024, 8192)):
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) / 32).bfloat16()
    cb = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    A, B = ops.operand(a, K, True), ops.operand(b, K, True)
    f = lambda: ops.gemm(M, N, K, A, B, cb, N, in_dtype=1, c_dtype=1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{N}x{K}: {e0.elapsed_time(e1) / 30 * 1e3:6.1f}")
print(os.environ.get("NBCI_GEMM_STAGGER", "-"), " | ".join(out), flush=True)
