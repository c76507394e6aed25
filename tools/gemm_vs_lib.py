"""The step's plain GEMM shapes on this library's kernels beside torch.matmul (hipBLASLt / rocBLAS) on the same box:
a yardstick for what these SHAPES allow (M = 64 x 143 = 9152 token rows is 2-3 rounds of tiles on 256 CUs), not a
dependency -- the product path never calls a library GEMM. bf16 operands, f32 accumulate; outputs bf16 (fwd / dgrad)
or f32 accumulate-into (wgrad, beta = 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops  # noqa: E402

dev = "cuda"
T = 9152


def timeit(f, n=20):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def row(kind, M, N, K, ours, lib):
    fl = 2.0 * M * N * K / 1e6
    print(f"{kind:6s} M={M:5d} N={N:5d} K={K:5d}   ours {ours:7.1f} us {fl / ours:7.1f} TF   torch.matmul {lib:7.1f} us {fl / lib:7.1f} TF"
          f"   ours/lib {ours / lib:5.2f}", flush=True)


# forward NT: y[T][N] = x[T][K] . w[N][K]^T
for N, K in ((1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)):
    x = torch.randn(T, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / 32).bfloat16()
    y = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    ours = timeit(lambda: ops.gemm(T, N, K, ops.operand(x, K, True), ops.operand(w, K, True), y, N, in_dtype=1, c_dtype=1))
    wt = w.t()
    lib = timeit(lambda: torch.matmul(x, wt, out=y))
    row("fwd NT", T, N, K, ours, lib)

# dgrad NN: dx[T][K] = dy[T][N] . w[N][K]   (contraction over N; w is row-major-in-k)
for N, K in ((1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)):
    dy = torch.randn(T, N, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / 32).bfloat16()
    dx = torch.empty(T, K, device=dev, dtype=torch.bfloat16)
    ours = timeit(lambda: ops.gemm(T, K, N, ops.operand(dy, N, True), ops.operand(w, K, False), dx, K, in_dtype=1, c_dtype=1))
    lib = timeit(lambda: torch.matmul(dy, w, out=dx))
    row("dgrad", T, K, N, ours, lib)

# wgrad TN: dw[N][K] += dy[T][N]^T . x[T][K]   (contraction over the T token rows; f32 accumulate-into)
for N, K in ((1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)):
    dy = (torch.randn(T, N, device=dev) / 8).bfloat16()
    x = (torch.randn(T, K, device=dev) / 8).bfloat16()
    dw = torch.zeros(N, K, device=dev)
    ours = timeit(lambda: ops.gemm(N, K, T, ops.operand(dy, N, False), ops.operand(x, K, False), dw, K, in_dtype=1, c_dtype=0, beta=1.0))
    dyt = dy.t()
    dwb = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
    lib = timeit(lambda: torch.matmul(dyt, x, out=dwb))    # library: bf16 out, no accumulate (less work than ours)
    row("wgrad", N, K, T, ours, lib)
