"""The one-launch MLP half (csrc/mlp_strip.hip: up projection + GELU + down projection + residual, row strip resident on its CU) against the two GEMM
launches of the step, same operands: results (g, act', y) and time (hot loop of 20 and per-launch events with operands rotated through > 256 MiB).
    python tools/ab_mlp_strip.py [--M 9152] [--residual bf16|fp32]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops            # noqa: E402
from llm_bci_amd._lib import check, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=9152)
ap.add_argument("--residual", default="bf16")
ap.add_argument("--sets", type=int, default=6)
a = ap.parse_args()
dev, M, H, I = "cuda", a.M, 1024, 1024
rdt = torch.bfloat16 if a.residual == "bf16" else torch.float32
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
Wu = (torch.randn(I, H, device=dev) / 32).bfloat16(); bu = torch.randn(I, device=dev) * 0.1
Wd = (torch.randn(H, I, device=dev) / 32).bfloat16(); bd = torch.randn(H, device=dev) * 0.1
sets = []
for _ in range(a.sets):
    h = torch.randn(M, H, device=dev).bfloat16(); x = torch.randn(M, H, device=dev).to(rdt)
    sets.append(dict(h=h, x=x, g=torch.zeros(M, I, device=dev, dtype=torch.bfloat16), da=torch.zeros(M, I, device=dev, dtype=torch.bfloat16),
                     y=torch.zeros(M, H, device=dev, dtype=rdt)))


def descs(s):
    up = ops.gemm_desc(M, I, H, ops.operand(s["h"], H, True), ops.operand(Wu, H, True), s["g"], I, in_dtype=1, c_dtype=1, bias=bu, act=2, C2=s["da"], c2_grad=1)
    dn = ops.gemm_desc(M, H, I, ops.operand(s["g"], I, True), ops.operand(Wd, I, True), s["y"], H, in_dtype=1, c_dtype=0 if rdt == torch.float32 else 1,
                       bias=bd, drop_p=0.4, seed=7, site=19, residual=s["x"], ldr=H)
    return up, dn


def two(s):
    up, dn = descs(s)
    check(lib().nbci_gemm(C.byref(up), st()), "up"); check(lib().nbci_gemm(C.byref(dn), st()), "down")


def one(s):
    up, dn = descs(s)
    check(lib().nbci_debug_mlp_strip(C.byref(up), C.byref(dn), st()), "mlp_strip")


two(sets[0]); torch.cuda.synchronize()
ref = {k: sets[0][k].float().clone() for k in ("g", "da", "y")}
for k in ("g", "da", "y"):
    sets[0][k].zero_()
one(sets[0]); torch.cuda.synchronize()
for k in ("g", "da", "y"):
    d = (sets[0][k].float() - ref[k]).abs()
    print(f"{k:3s}: max |diff| {d.max().item():.3e}  mismatching elements {(d > 0).sum().item()} / {d.numel()}  (ref abs max {ref[k].abs().max().item():.2f})")


def hot(fn, n=20):
    for _ in range(3):
        fn(sets[0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(sets[0])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rotated(fn, n=24):
    evs = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(sets[i % len(sets)]); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
    return ts[len(ts) // 2]


for r in range(3):
    print(f"round {r}: two launches hot {hot(two):7.1f} us  rotated {rotated(two):7.1f} us   |   one launch hot {hot(one):7.1f} us  rotated {rotated(one):7.1f} us", flush=True)
