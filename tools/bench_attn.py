"""Micro-benchmark of the fused attention kernels (run on the GPU box)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd._lib import check, lib  # noqa: E402

B, nh, Tp, H = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 8, 143, 1024
dev = "cuda"
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
qkv = torch.randn(B * Tp, 3 * H, device=dev).bfloat16()
tm = torch.ones(B, Tp, dtype=torch.int32, device=dev)
out = torch.zeros(B * Tp, H, device=dev, dtype=torch.bfloat16)
da = torch.randn(B * Tp, H, device=dev).bfloat16()
ldP = (Tp + 7) // 8 * 8
dS = torch.zeros(B * nh * Tp * ldP, device=dev, dtype=torch.bfloat16)
Pd = torch.zeros_like(dS)
dqkv = torch.zeros_like(qkv)
bg = torch.zeros(3 * H, device=dev)
lse = torch.zeros(B * nh * Tp, device=dev)
l = lib()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for p in (0.0, 0.4):
    f = lambda: check(l.nbci_attention_fwd(vp(qkv), vp(tm), vp(out), vp(lse), B, nh, Tp, H, -2, -2, p, 1, 16, 17, st()), "fwd")
    b1 = lambda: check(l.nbci_attention_bwd(vp(qkv), vp(tm), vp(out), vp(lse), vp(da), vp(dS), vp(Pd), ldP, vp(dqkv), vp(bg), B, nh, Tp, H, -2, -2, p, 1, 16, st()), "bwd")
    b0 = lambda: check(l.nbci_attention_bwd(vp(qkv), vp(tm), vp(out), vp(lse), vp(da), vp(dS), vp(Pd), ldP, vp(dqkv), None, B, nh, Tp, H, -2, -2, p, 1, 16, st()), "bwd")
    print(f"p={p}: fwd {timeit(f):.1f} us   bwd(bias) {timeit(b1):.1f} us   bwd(no bias) {timeit(b0):.1f} us", flush=True)
