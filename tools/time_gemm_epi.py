"""epilogue cost: time the M x 1024 x 1024 NT GEMM with the fused epilogues the model uses."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 9152
N = K = 1024
dev = "cuda"
a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) / 32).bfloat16()
bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev)
cf = torch.zeros(M, N, device=dev); cb = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); c2 = torch.zeros_like(cb)
A, B = ops.operand(a, K, True), ops.operand(b, K, True)
Bn = ops.operand(b, K, False)
cases = {
    "plain f32 out": lambda: ops.gemm(M, N, K, A, B, cf, N, in_dtype=1, c_dtype=0),
    "plain bf16 out": lambda: ops.gemm(M, N, K, A, B, cb, N, in_dtype=1, c_dtype=1),
    "bias+residual f32 (out_proj)": lambda: ops.gemm(M, N, K, A, B, cf, N, in_dtype=1, c_dtype=0, bias=bias, residual=res, ldr=N),
    "bias+dropout+residual f32 (down)": lambda: ops.gemm(M, N, K, A, B, cf, N, in_dtype=1, c_dtype=0, bias=bias, residual=res, ldr=N, drop_p=0.4, seed=1, site=2),
    "bias+gelu+C2grad bf16 (up)": lambda: ops.gemm(M, N, K, A, B, cb, N, in_dtype=1, c_dtype=1, bias=bias, act=2, C2=c2, c2_grad=1),
    "dgrad plain f32 (dh)": lambda: ops.gemm(M, N, K, A, Bn, cf, N, in_dtype=1, c_dtype=0),
    "dgrad dropout bf16 (da)": lambda: ops.gemm(M, N, K, A, Bn, cb, N, in_dtype=1, c_dtype=1, drop_p=0.4, seed=1, site=2),
}
for name, f in cases.items():
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"M={M} {name:36s} {e0.elapsed_time(e1)/20*1e3:7.1f} us")
