"""Why is a step's K = 1024 GEMM launch slower inside the train step than in a timing loop? (34.8 vs 25.3 us.)
Times the step's shapes with per-launch events under four operand states, same process, interleaved rounds:
  hot        : the same buffers every launch (what tools/time_gemm*.py measure: A, W, residual, C all served from L2 / Infinity Cache);
  rot_all    : operand SETS rotated so that the footprint between two uses of a buffer exceeds the 256 MiB Infinity Cache (everything from HBM);
  rot_act    : activations (A, residual, C) rotated, ONE weight (the step's state for a layer's weight is in between: read once per step);
  after_ln   : like the step: a LayerNorm-forward-like producer writes A just before the GEMM reads it (A hot, everything else cold).
Prints the median launch duration (us) per case and shape.

    python tools/gemm_cold_ab.py [--rounds 7] [--sets 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--sets", type=int, default=10)
ap.add_argument("--M", type=int, default=9152)
a = ap.parse_args()
dev = "cuda"
M, S = a.M, a.sets


def make(N, K, res_dtype):
    sets = []
    for _ in range(S):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) / 32).bfloat16()
        R = torch.randn(M, N, device=dev).to(res_dtype) if res_dtype is not None else None
        Cb = torch.zeros(M, N, device=dev, dtype=res_dtype if res_dtype is not None else torch.bfloat16)
        sets.append((A, W, R, Cb))
    return sets


def launch(N, K, st, W=None, epi="res"):
    A, W0, R, Cb = st
    W = W0 if W is None else W
    kw = {}
    if epi == "res":
        kw = dict(bias=bias[N], residual=R, ldr=N, drop_p=0.4, seed=1, site=2)
    elif epi == "bias":
        kw = dict(bias=bias[N])
    ops.gemm(M, N, K, ops.operand(A, K, True), ops.operand(W, K, True), Cb, N, in_dtype=1, c_dtype=0 if Cb.dtype == torch.float32 else 1, **kw)


bias = {n: torch.randn(n, device=dev) for n in (1024, 3072)}
shapes = [("out/down N=1024 K=1024, bias+dropout+bf16 residual", 1024, 1024, torch.bfloat16, "res"),
          ("out/down N=1024 K=1024, bias+dropout+f32 residual", 1024, 1024, torch.float32, "res"),
          ("qkv N=3072 K=1024, bias, bf16 out", 3072, 1024, None, "bias")]


def timed(fn, n):
    """per-launch events; returns the median launch duration in us"""
    evs = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(i); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
    return ts[len(ts) // 2]


for name, N, K, rdt, epi in shapes:
    sets = make(N, K, rdt)
    src = torch.randn(M, K, device=dev)
    res = {k: [] for k in ("hot", "rot_all", "rot_act", "after_ln")}
    for r in range(a.rounds):
        res["hot"].append(timed(lambda i: launch(N, K, sets[0], epi=epi), 20))
        res["rot_all"].append(timed(lambda i: launch(N, K, sets[i % S], epi=epi), 2 * S))
        res["rot_act"].append(timed(lambda i: launch(N, K, sets[i % S], W=sets[0][1], epi=epi), 2 * S))

        def after_ln(i):
            st = sets[i % S]
            st[0].copy_(src)       # a producer kernel writes A (f32 -> bf16 cast: 37 MB read, 18.7 MB written) right before the GEMM reads it
            launch(N, K, st, epi=epi)
        # (the event pair brackets producer + GEMM here; the producer alone is timed and subtracted)
        both = timed(after_ln, 2 * S)
        prod = timed(lambda i: sets[i % S][0].copy_(src), 2 * S)
        res["after_ln"].append(both - prod)
    print(name)
    for k, v in res.items():
        v = sorted(v)
        print(f"    {k:9s} median {v[len(v) // 2]:7.1f} us   min {v[0]:7.1f}   max {v[-1]:7.1f}")
    del sets
    torch.cuda.empty_cache()
