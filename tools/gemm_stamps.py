"""Where a GEMM launch spends its time, per workgroup: in-kernel wall-clock stamps (100 MHz) from a MEASUREMENT build of the
library (gemm_glds.hip compiled with -DNBCI_STAMPS, loaded through NBCI_LIB; the product build has no stamps).

  NBCI_LIB=build/ab/libnbci_stamps.so python tools/gemm_stamps.py 9152 4096 1024

Stamps per workgroup (wave 0): 0 entry, 1 first K tile landed, 2 K loop done, 6 accumulators in the LDS tile, 3 epilogue stores
issued, 4 stores acknowledged.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops  # noqa: E402
from llm_bci_amd._lib import lib  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (9152, 4096, 1024)
dev = "cuda"
a = torch.randn(M, K, device=dev).bfloat16()
b = (torch.randn(N, K, device=dev) / 32).bfloat16()
c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
A, B = ops.operand(a, K, True), ops.operand(b, K, True)
f = lambda: ops.gemm(M, N, K, A, B, c, N, in_dtype=1, c_dtype=1)  # noqa: E731
for _ in range(5):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record()
torch.cuda.synchronize()
nblk = min(8192, max(((M + bm - 1) // bm) * (N // 128) for bm in (128, 144, 160)))
buf = np.zeros((nblk, 8), dtype=np.uint64)
rd = lib().nbci_debug_read_stamps
rd.argtypes = [C.c_void_p, C.c_int]
assert rd(buf.ctypes.data, nblk) == 0
buf = buf[buf[:, 0] > 0]
t = buf[:, :5].astype(np.int64)
t6 = buf[:, 6].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
print(f"M={M} N={N} K={K}: {len(buf)} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us, "
      f"first entry -> last store acknowledged {us[:, 4].max():.1f} us")


def q(x):
    return "  ".join(f"{np.percentile(x, p):6.2f}" for p in (5, 25, 50, 75, 95))


print("phase (us)                 p5     p25    p50    p75    p95")
print("entry after first      ", q(us[:, 0]))
print("fill   (0->1)          ", q(us[:, 1] - us[:, 0]))
print("K loop (1->2)          ", q(us[:, 2] - us[:, 1]))
print("epilogue issue (2->3)  ", q(us[:, 3] - us[:, 2]))
if t6.max() > 0:
    u6 = (t6 - t0) / 100.0
    print("  accumulators -> LDS  ", q(u6 - us[:, 2]))
    print("  row loop             ", q(us[:, 3] - u6))
print("store drain (3->4)     ", q(us[:, 4] - us[:, 3]))
print("whole workgroup (0->4) ", q(us[:, 4] - us[:, 0]))
hw = buf[:, 5]
cu = ((hw >> 32) & 0xF) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xF)   # xcc, se, sh, cu
ids, counts = np.unique(cu, return_counts=True)
print(f"{len(ids)} distinct CUs; workgroups per CU min/median/max {counts.min()}/{int(np.median(counts))}/{counts.max()}")
busy = np.zeros(len(ids))
for i, k in enumerate(ids):
    w = us[cu == k]
    busy[i] = w[:, 4].max()
print("per-CU finish time (us)", q(busy))
one = ids[len(ids) // 2]
print(f"timeline of CU {one}: (entry, K loop start, K loop end, stores issued, acknowledged)")
for row in sorted(us[cu == one].tolist()):
    print("   " + "  ".join(f"{x:7.2f}" for x in row))

# ---- K-loop stamps (shader cycles), wave 0 of each workgroup, K tiles 4..11
try:
    rk = lib().nbci_debug_read_kstamps
except AttributeError:
    rk = None
if rk is not None:
    rk.argtypes = [C.c_void_p, C.c_int]
    nb = min(1024, nblk)
    kb = np.zeros((nb, 8, 8), dtype=np.uint64)
    assert rk(kb.ctypes.data, nb) == 0
    k = kb[kb[:, :, 0].min(axis=1) > 0].astype(np.int64)
    if len(k):
        print(f"K loop, per K tile (shader cycles; wave 0 of {len(k)} workgroups x 8 tiles; MFMA work of the wave per tile = "
              f"36 x 16 = 576 cycles at the 144-row tile):")
        names = ("LDS-DMA issue (0->1)", "reads + first k-step issued, second landed (1->2)", "second k-step MFMAs issued (2->3)",
                 "barrier incl. next tile's vmcnt (3->4)", "loop top -> loop top")
        for i, nm in enumerate(names[:4]):
            print(f"  {nm:52s}", q((k[:, :, i + 1] - k[:, :, i]).ravel()))
        print(f"  {names[4]:52s}", q((k[:, 1:, 0] - k[:, :-1, 0]).ravel()))
