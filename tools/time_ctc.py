"""nbci_ctc alone (the step's shape: B = 64, V = 41, S = 60 labels, T' = 143 frames), and how its time moves with the frame count and the batch:
what is the per-frame cost of the alpha / beta recursion and what is fixed (staging, gradient pass, launch)?   python tools/time_ctc.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ndt1  # noqa: F401,E402
from llm_bci_amd._lib import check, lib  # noqa: E402

l = lib()
dev = "cuda"


def vp(t):
    return C.c_void_p(t.data_ptr())


def run(B, Tp, V, S, reps=50):
    torch.manual_seed(0)
    lp = torch.log_softmax(torch.randn(B, Tp, V, device=dev), -1).contiguous()
    tg = torch.randint(1, V, (B, S), device=dev)
    il = torch.full((B,), Tp, device=dev, dtype=torch.int32)
    tl = torch.full((B,), min(S, (Tp - 1) // 2), device=dev, dtype=torch.int64)
    loss = torch.zeros(B, device=dev)
    ws = torch.zeros(max(1, int(l.nbci_ctc_workspace_floats(B, Tp, S))), device=dev)
    ldd = (V + 7) // 8 * 8
    dl = torch.zeros(B, Tp, ldd, device=dev, dtype=torch.bfloat16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def f():
        check(l.nbci_ctc(vp(lp), vp(tg), vp(il), vp(tl), B, Tp, V, S, 0, 1, vp(loss), vp(ws), vp(dl), 1, ldd, C.c_float(1.0), st), "ctc")
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for B, Tp, S in ((64, 143, 60), (64, 72, 60), (64, 36, 60), (64, 143, 20), (8, 143, 60), (256, 143, 60)):
    print(f"B={B:4d} T'={Tp:4d} S={S:3d}: {run(B, Tp, 41, S):7.1f} us", flush=True)
