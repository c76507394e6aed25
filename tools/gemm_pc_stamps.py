"""Where the producer / consumer GEMM (gemm_pc.hip) spends a K tile, from a MEASUREMENT build (-DNBCI_STAMPS):
    tools/build_variant.sh stamps -DNBCI_STAMPS
    NBCI_LIB=build/stamps/libnbci.so python tools/gemm_pc_stamps.py 9152 1024 1024
Producer wave 4 of every workgroup stamps (shader cycles) per K tile k: tile k landed -> past barrier B_k -> FREE polled and
tile k+2 issued -> tile k+1 landed -> past B_{k+1}. A long "landed -> past barrier" = the producers wait for the CONSUMERS
(consumer-bound: good); a long "wait until landed" = the consumers wait for the DMA. Wall clock (100 MHz): entry, B_0, loop end, epilogue begin / end."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops  # noqa: E402
from llm_bci_amd._lib import lib  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (9152, 1024, 1024)
bk = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
dev = "cuda"
l = lib()
l.nbci_debug_gemm_pc(2)
a = torch.randn(M, K, device=dev).bfloat16()
b = (torch.randn(N, K, device=dev) / 32).bfloat16() if bk else (torch.randn(K, N, device=dev) / 32).bfloat16()
c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
A, B = ops.operand(a, K, True), ops.operand(b, K if bk else N, bk)
f = lambda: ops.gemm(M, N, K, A, B, c, N, in_dtype=1, c_dtype=1)  # noqa: E731
for _ in range(5):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record()
torch.cuda.synchronize()
nblk = min(1024, ((M + 143) // 144) * ((N + 255) // 256))
tile = np.zeros((nblk, 64, 4), dtype=np.uint64)
wall = np.zeros((nblk, 8), dtype=np.uint64)
rd = l.nbci_debug_read_pc_stamps
rd.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert rd(tile.ctypes.data, wall.ctypes.data, nblk) == 0
nt = K // 64
t = tile[:, :nt].astype(np.int64)
w = wall.astype(np.int64)
w0 = w[:, 0].min()
us = (w - w0) / 100.0


def q(x):
    x = np.asarray(x).reshape(-1)
    return "  ".join(f"{np.percentile(x, p):8.2f}" for p in (5, 25, 50, 75, 95))


print(f"M={M} N={N} K={K} B {'k-major' if bk else 'row-major-in-k'}: {nblk} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us")
print("wall clock per workgroup (us)       p5       p25      p50      p75      p95")
print("entry after first                ", q(us[:, 0]))
print("fill: entry -> B_0               ", q(us[:, 1] - us[:, 0]))
print("K loop: B_0 -> last tile landed  ", q(us[:, 2] - us[:, 1]))
print("consumers' tail + acc -> LDS     ", q(us[:, 3] - us[:, 2]))
print("row loop (stores issued)         ", q(us[:, 4] - us[:, 3]))
print("whole workgroup                  ", q(us[:, 4] - us[:, 0]))
# per tile k (producer wave 4): [2] tile k landed -> B_k -> [0] past B_k -> poll FREE, issue tile k+2 -> [1] -> wait -> [2] of tile k+1
k0, k1 = 2, nt - 3
a = t[:, k0:k1]
nx = t[:, k0 + 1:k1 + 1]
print(f"per K tile, producer wave 4 (shader cycles), tiles {k0} .. {k1 - 1}:")
print("poll FREE + issue tile k+2       ", q(a[:, :, 1] - a[:, :, 0]))
print("wait until tile k+1 has landed   ", q(nx[:, :, 2] - a[:, :, 1]))
print("landed -> past barrier B_k+1     ", q(nx[:, :, 0] - nx[:, :, 2]))
print("barrier to barrier               ", q(nx[:, :, 0] - a[:, :, 0]))
