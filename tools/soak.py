"""Soak of the native train step over batch sizes and ragged lengths (each a different set of kernel families and launch schemes:
balanced grouped launches with and without a partial last K tile, the two CTC kernels, fused / streaming attention). Prints the loss
trajectory ends; exits non-zero on a non-finite loss. Usage: python tools/soak.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from llm_bci_amd.ndt1 import NDT1  # noqa: E402
from llm_bci_amd.trainer import NativeTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
for B, bins, ragged in ((64, 600, False), (64, 600, True), (32, 600, True), (16, 600, False), (128, 600, True), (8, 600, True), (16, 1200, True), (5, 332, True)):
    torch.manual_seed(1)
    model = NDT1({"encoder": {"embedder": {"n_channels": 256}}}, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True,
                 compute_dtype="bf16").to(dev)
    tr = NativeTrainer(model, lr=1e-3, wd=5e-5, eps=1e-8, scheduler="cosine", total_steps=steps + 8, warmup_pct=0.0, div_factor=25)
    batches = [bench.make_batch(B, bins, 256, 60, 41, dev, seed=s, ragged=ragged)[1] for s in range(4)]
    first = last = None
    for i in range(steps):
        loss, _ = tr.train_step(batches[i % 4], seed=i)
        if i == 0 or i == steps - 1:
            v = float(loss.sum().item()) / B
            first = v if i == 0 else first
            last = v
    torch.cuda.synchronize()
    st = tr.read_stats()
    ok = first == first and last == last and abs(last) < 1e9
    print(f"B={B:3d} bins={bins} ragged={ragged}: loss/sample {first:.2f} -> {last:.2f}  PER {st['PER']:.3f}  {'ok' if ok else 'NOT FINITE'}", flush=True)
    if not ok:
        sys.exit(1)
print("soak ok")
