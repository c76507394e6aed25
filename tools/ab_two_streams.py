"""Hypothesis test (round 4): every launch of the B = 64 step fills the chip's two workgroup slots per CU with ONE kernel whose workgroups run in
lockstep - all in their K loops (memory idle), then all in their epilogues (MFMA idle), then a launch boundary. Would two half-batch steps on two
streams, each kernel half as large, fill each other's bubbles? Measured here with what exists: TWO independent trainers (own weights, gradients,
AdamW) at B = 32 each, their steps enqueued on two streams, against one trainer at B = 64 (same FLOPs; the pair pays AdamW twice and runs its
weight gradients at half the K).

    python tools/ab_two_streams.py [--steps 20] [--windows 5] [--residual-dtype bf16]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_batch  # noqa: E402
from llm_bci_amd.ndt1 import NDT1  # noqa: E402
from llm_bci_amd.trainer import NativeTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--windows", type=int, default=5)
ap.add_argument("--residual-dtype", default="bf16")
ap.add_argument("--total", type=int, nargs="*", default=[64, 16, 8])
a = ap.parse_args()
dev = torch.device("cuda", 0)


def build(side):
    torch.manual_seed(1)
    m = NDT1({}, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype="bf16", residual_dtype=a.residual_dtype).to(dev)
    return NativeTrainer(m, total_steps=1_000_000, side_stream=side)


def window(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps * 1e3


for total in a.total:
    half = total // 2
    _, bf = make_batch(total, 600, 256, 60, 41, dev, seed=0)
    _, b1 = make_batch(half, 600, 256, 60, 41, dev, seed=1)
    _, b2 = make_batch(half, 600, 256, 60, 41, dev, seed=2)
    one, one_side = build(False), build("auto")
    t1, t2 = build(False), build(False)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def pair(i):
        with torch.cuda.stream(s1):
            t1.train_step(b1, seed=i)
        with torch.cuda.stream(s2):
            t2.train_step(b2, seed=i)

    cases = {f"one trainer B={total}, one stream": lambda i: one.train_step(bf, seed=i),
             f"one trainer B={total}, side stream (shipped)": lambda i: one_side.train_step(bf, seed=i),
             f"one trainer B={half} alone": lambda i: t1.train_step(b1, seed=i),
             f"two trainers B={half} + B={half} on two streams": pair}
    res = {k: [] for k in cases}
    for k, fn in cases.items():
        for i in range(3):
            fn(i)
    for w in range(a.windows):
        for k, fn in cases.items():
            res[k].append(window(fn))
    for k, v in res.items():
        v = sorted(v)
        print(f"{k:55s} median {v[len(v) // 2]:.3f} ms  min {v[0]:.3f}  max {v[-1]:.3f}", flush=True)
    del one, one_side, t1, t2
    torch.cuda.empty_cache()
