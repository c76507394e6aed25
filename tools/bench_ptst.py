"""PatchTST train-step timing on one GPU (BASELINE.json configs[4] shapes: 1024 ch x 2048(->2050) steps, patch 10 -> 205 patches).
    python tools/bench_ptst.py [--batch 2] [--channels 1024] [--steps 5] [--dtype bf16] [--method ctc]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fwd_flops(B, C, P=205, D=256, F=1024, L=4, pl=10, V=41):
    M = B * C * P
    layer = 2 * M * (4 * D * D + 2 * D * F) + 4 * B * C * P * P * D
    return 2 * M * pl * D + L * layer + 2 * B * P * D * V


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--channels", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--residual-dtype", default="bf16", help="fp32 | bf16 (the models default to fp32 = parity; the benches measure the bf16 streams unless told otherwise)")
    ap.add_argument("--method", default="ctc")
    a = ap.parse_args()
    from llm_bci_amd.patchtst import PatchTSTForSpikingActivity
    from llm_bci_amd.trainer import NativeTrainer
    torch.manual_seed(1)
    T = 2050
    over = {"encoder": {"num_input_channels": a.channels, "context_length": T, "do_mask_input": a.method == "mlm"}}
    kw = dict(method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True) if a.method == "ctc" else dict(method_name="mlm", loss="poisson_nll", log_input=True)
    m = PatchTSTForSpikingActivity(over, compute_dtype=a.dtype, residual_dtype=("fp32" if a.dtype == "fp32" else a.residual_dtype), **kw).to("cuda")
    tr = NativeTrainer(m, lr=1e-3, wd=5e-5, eps=1e-8, scheduler="cosine", total_steps=1000, warmup_pct=0.0, div_factor=25, compute_per=a.method == "ctc")
    g = np.random.default_rng(0)
    B, C = a.batch, a.channels
    batch = {"spikes": torch.from_numpy(g.standard_normal((B, T, C)).astype(np.float32)).cuda(), "spikes_mask": torch.ones(B, T, dtype=torch.int64, device="cuda"),
             "spikes_lengths": torch.full((B,), T, dtype=torch.int64, device="cuda")}
    if a.method == "ctc":
        batch["targets"] = torch.from_numpy(g.integers(1, 41, (B, 60))).cuda()
        batch["targets_lengths"] = torch.full((B,), 60, dtype=torch.int64, device="cuda")
    else:
        batch["spikes"] = torch.from_numpy(g.poisson(0.5, (B, T, C)).astype(np.float32)).cuda()
    for i in range(a.warmup):
        tr.train_step(batch, seed=10 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        tr.train_step(batch, seed=100 + i)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.steps
    st = tr.read_stats()
    fl = 3 * fwd_flops(B, C)
    print(f"PatchTST {a.method}  B={B} C={C} T={T} P=205 {a.dtype}: {el * 1e3:.2f} ms/step  {B / el:.2f} samples/s  "
          f"{fl / el / 1e12:.1f} model TFLOP/s  loss {st['loss']:.4f}  ws {m._ws.numel() / 2**30:.2f} GiB", flush=True)


if __name__ == "__main__":
    main()
