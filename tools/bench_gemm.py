"""Micro-benchmark of nbci_gemm on the NDT1 C2 shapes (run on the GPU box)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops  # noqa: E402


def bench(M, N, K, ak, bk, dtype, splitk=1, iters=20, cdt=torch.float32):
    dev = "cuda"
    a = (torch.randn(M, K, device=dev) if ak else torch.randn(K, M, device=dev)).to(dtype)
    b = (torch.randn(N, K, device=dev) if bk else torch.randn(K, N, device=dev)).to(dtype)
    c = torch.zeros(M, N, device=dev, dtype=cdt)
    A = ops.operand(a, a.stride(0), ak)
    B = ops.operand(b, b.stride(0), bk)

    def run():
        ops.gemm(M, N, K, A, B, c, N, in_dtype=ops._dt(a), c_dtype=ops._dt(c), splitk=splitk)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9


if __name__ == "__main__":
    Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    Mtok = Bn * 143
    rows = []
    for dtype in (torch.bfloat16, torch.float32):
        for name, (M, N, K, ak, bk, sk) in {
            "fwd_qkv   NT": (Mtok, 3072, 1024, True, True, 1),
            "fwd_proj  NT": (Mtok, 1024, 1024, True, True, 1),
            "fwd_stack NT": (Mtok, 1024, 8192, True, True, 1),
            "dgrad     NN": (Mtok, 1024, 1024, True, False, 1),
            "dgrad_qkv NN": (Mtok, 1024, 3072, True, False, 1),
            "wgrad     TN": (1024, 1024, Mtok, False, False, 4),
            "wgrad_qkv TN": (3072, 1024, Mtok, False, False, 2),
            "wgrad_stk TN": (1024, 8192, Mtok, False, False, 1),
        }.items():
            ms, tf = bench(M, N, K, ak, bk, dtype, sk)
            rows.append({"case": name, "dtype": str(dtype), "M": M, "N": N, "K": K, "splitk": sk,
                         "ms": round(ms, 4), "TFLOPs": round(tf, 1)})
            print(rows[-1], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/bench_gemm.json", "w"), indent=1)
