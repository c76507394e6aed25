"""iTransformer SSL train-step timing on one GPU (BASELINE.json configs[2]: trainer_ssl_itransformer.yaml shapes).
    python tools/bench_itr.py [--batch 16] [--channels 668] [--steps 10] [--dtype bf16]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fwd_flops(B, N, T=100, H=768, L=5, use_cls=1):
    S = N + use_cls
    emb = 2 * B * N * (T * H + H * H)
    layer = 2 * B * S * (3 * H * H + H * H + 8 * H * H) + 4 * B * S * S * H
    dec = 2 * B * S * (H * H + H * T)
    return emb + L * layer + dec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--channels", type=int, default=668)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--residual-dtype", default="bf16", help="fp32 | bf16 (the models default to fp32 = parity; the benches measure the bf16 streams unless told otherwise)")
    a = ap.parse_args()
    from llm_bci_amd.itransformer import iTransformer
    from llm_bci_amd.trainer import NativeTrainer
    torch.manual_seed(1)
    over = {"encoder": {"embed_region": False}, "masker": {"main": {"active": True}}}
    m = iTransformer(over, method_name="mlm", loss="poisson_nll", log_input=True, compute_dtype=a.dtype, residual_dtype=("fp32" if a.dtype == "fp32" else a.residual_dtype)).to("cuda")
    tr = NativeTrainer(m, lr=1e-4, wd=0.01, eps=1e-8, scheduler="cosine", total_steps=1000, warmup_pct=0.15, div_factor=25, compute_per=False)
    g = np.random.default_rng(0)
    B, T, N = a.batch, 100, a.channels
    batch = {"spikes": torch.from_numpy(g.poisson(0.5, (B, T, N)).astype(np.float32)).cuda(),
             "spikes_mask": torch.ones(B, T, dtype=torch.int64, device="cuda"),
             "spikes_timestamp": torch.arange(T, device="cuda").repeat(B, 1)}
    for i in range(a.warmup):
        tr.train_step(batch, seed=10 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        tr.train_step(batch, seed=100 + i)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.steps
    st = tr.read_stats()
    fl = 3 * fwd_flops(B, N)
    print(f"iTransformer mlm  B={B} N={N} T={T} {a.dtype}: {el * 1e3:.2f} ms/step  {B / el:.1f} samples/s  "
          f"{fl / el / 1e12:.1f} model TFLOP/s  loss/bin {st['loss']:.4f}  ws {m._ws.numel() / 2**30:.2f} GiB", flush=True)


if __name__ == "__main__":
    main()
