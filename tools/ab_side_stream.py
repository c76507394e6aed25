"""In-box A/B of NativeTrainer's side stream (weight gradients + fold + per-segment AdamW beside the backward chain) over batch sizes:
one process, modes interleaved, median of windows.   python tools/ab_side_stream.py [--batches 8 16 32 64] [--steps 20] [--windows 7]"""
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
ap.add_argument("--batches", type=int, nargs="*", default=[8, 16, 32, 64])
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--windows", type=int, default=7)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = NDT1({}, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype="bf16").to(dev)
tr = NativeTrainer(model, total_steps=1_000_000, side_stream=False)
for B in args.batches:
    _, b = make_batch(B, 600, 256, 60, 41, dev, seed=0)
    res = {False: [], True: []}
    for mode in (False, True):
        tr.side_stream = mode
        for i in range(3):
            tr.train_step(b, seed=i)
    for w in range(args.windows):
        for mode in (False, True):
            tr.side_stream = mode
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                tr.train_step(b, seed=100 * w + i)
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) / args.steps * 1e3)
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    print(f"B={B:3d} rows={B * 143:5d}  one stream {med[False]:.3f} ms (min {min(res[False]):.3f})   side stream {med[True]:.3f} ms (min {min(res[True]):.3f})"
          f"   ratio {med[True] / med[False]:.3f}", flush=True)
