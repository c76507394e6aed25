"""In-box A/B of the bf16 GEMM kernel families on the train step's shapes: the two-workgroup-per-CU kernels (mode 0) against the
producer / consumer kernel (mode 2, gemm_pc.hip), interleaved rounds in ONE process (median and min of the rounds), with
torch.matmul (hipBLASLt) beside them as the yardstick. bf16 in / bf16 out, random data."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops  # noqa: E402
from llm_bci_amd._lib import lib  # noqa: E402

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 9152
ZERO = os.environ.get("ZERO") == "1"   # zero-filled operands: the same instruction stream at the clock the chip holds without data toggling
if ZERO:
    torch.randn = lambda *a, **k: torch.zeros(*a, **k)
l = lib()


def timeit(f, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = []
for kind, N, K in (("fwd", 1024, 1024), ("fwd", 3072, 1024), ("fwd", 4096, 1024), ("fwd", 1024, 4096), ("fwd", 1024, 8192),
                   ("dgrad", 1024, 1024), ("dgrad", 1024, 3072), ("dgrad", 1024, 4096), ("dgrad", 4096, 1024)):
    x = torch.randn(M, K, device=dev).bfloat16()
    if kind == "fwd":
        w = (torch.randn(N, K, device=dev) / 32).bfloat16()
        A, B = ops.operand(x, K, True), ops.operand(w, K, True)
        wl = w.t()
    else:   # dx[M][N] = dy[M][K] . w[K][N]
        w = (torch.randn(K, N, device=dev) / 32).bfloat16()
        A, B = ops.operand(x, K, True), ops.operand(w, N, False)
        wl = w
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm(M, N, K, A, B, y, N, in_dtype=1, c_dtype=1)
    fl = lambda: torch.matmul(x, wl, out=y)
    res = {0: [], 2: [], "lib": []}
    for mode in (0, 2):
        l.nbci_debug_gemm_pc(mode); f(); f()
    fl(); fl()
    for rnd in range(5):
        for mode in (0, 2):
            l.nbci_debug_gemm_pc(mode)
            res[mode].append(timeit(f))
        res["lib"].append(timeit(fl))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    mn = {k: min(v) for k, v in res.items()}
    flop = 2.0 * M * N * K / 1e6
    print(f"{kind:5s} M={M} N={N:5d} K={K:5d}  2wg/CU {med[0]:7.1f} us ({flop / med[0]:6.1f} TF, min {mn[0]:6.1f})   prod/cons {med[2]:7.1f} us "
          f"({flop / med[2]:6.1f} TF, min {mn[2]:6.1f})   torch.matmul {med['lib']:7.1f} us ({flop / med['lib']:6.1f} TF)   pc/2wg {med[2] / med[0]:5.2f}  pc/lib {med[2] / med['lib']:5.2f}",
          flush=True)
l.nbci_debug_gemm_pc(1)
