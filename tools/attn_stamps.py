"""Phases of the attention forward kernel per workgroup (wave 0): in-kernel wall-clock stamps from a MEASUREMENT build
(attention.hip compiled with -DNBCI_STAMPS, loaded through NBCI_LIB). Stamps: 0 entry, 1 K/V images in LDS, 2 scores,
3 softmax, 4 dropout, 5 P.V, 6 end (stores issued)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd._lib import check, lib  # noqa: E402

B, nh, Tp, H = 64, 8, 143, 1024
dev = "cuda"
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
qkv = torch.randn(B * Tp, 3 * H, device=dev).bfloat16()
tm = torch.ones(B, Tp, dtype=torch.int32, device=dev)
out = torch.zeros(B * Tp, H, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(B * nh * Tp, device=dev)
l = lib()
f = lambda: check(l.nbci_attention_fwd(vp(qkv), vp(tm), vp(out), vp(lse), B, nh, Tp, H, -2, -2, 0.4, 1, 16, 17, st()), "fwd")  # noqa: E731
for _ in range(5):
    f()
torch.cuda.synchronize()
f()
torch.cuda.synchronize()
n = B * nh
buf = np.zeros((n, 8), dtype=np.uint64)
rd = l.nbci_debug_read_attn_stamps
rd.argtypes = [C.c_void_p, C.c_int]
assert rd(buf.ctypes.data, n) == 0
t = buf[:, :7].astype(np.int64)
us = (t - t[:, 0].min()) / 100.0
q = lambda x: "  ".join(f"{np.percentile(x, p):6.2f}" for p in (5, 25, 50, 75, 95))  # noqa: E731
print(f"{n} workgroups; first entry -> last end {us[:, 6].max():.1f} us")
print("phase (us)              p5     p25    p50    p75    p95")
print("entry after first   ", q(us[:, 0]))
for name, a, b in (("K/V images -> LDS  ", 0, 1), ("Q load + scores    ", 1, 2), ("softmax            ", 2, 3), ("dropout            ", 3, 4),
                   ("P.V                ", 4, 5), ("output stores      ", 5, 6), ("whole workgroup    ", 0, 6)):
    print(name, q(us[:, b] - us[:, a]))
