"""bench.py — train-step samples/sec of the HIP NDT1-CTC path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full train step of the reference's loop (trainer.py:332-362 semantics) over one
synthetic batch already resident in HBM: forward (recipe dropout + noise ON) -> CTC sum-loss ->
backward -> (N>1: overlapped RCCL all-reduce) -> fused AdamW + OneCycle -> zero_grad, plus the
on-device PER metric the reference computes every step. Workload = BASELINE.json configs[1]:
default configs/ndt1.yaml (5 layers x 1024, 41.06 M params), 256 ch x 600 bins, bf16 operands.
Weak scaling: --batch is PER GPU (default 64 = the recipe's train_batch_size).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3     # f32-input MFMA = f32 vector peak
KIND_NAMES = {0: "f32 TN(A^T B^T)", 1: "f32 A^T.B", 2: "f32 A.B(kn)", 3: "f32 NT",
              4: "bf16 wgrad (A,B row-major-in-k)", 5: "bf16 (A row-major-in-k, B k-major)",
              6: "bf16 dgrad (A k-major, B row-major-in-k)", 7: "bf16 fwd NT (both k-major)"}


def fwd_flops_per_sample(T, N, D=256, S=32, st=4, H=1024, I=1024, L=5, V=41):
    Tp = 1 + (T - S) // st
    return 2.0 * (13 * T * N + T * N * D + Tp * (S * D) * H + L * (4 * Tp * H * H + 2 * Tp * H * I + 4 * Tp * Tp * H) + Tp * H * V), Tp


def make_batch(B, T, N, S, vocab, dev, seed):
    g = np.random.default_rng(seed)
    b = dict(spikes=g.standard_normal((B, T, N)).astype(np.float32), spikes_mask=np.ones((B, T), np.int64),
             spikes_timestamp=np.tile(np.arange(T), (B, 1)), spikes_lengths=np.full(B, T, np.int64),
             targets=g.integers(1, vocab, (B, S)).astype(np.int64), targets_lengths=np.full(B, S, np.int64))
    return b, {k: torch.from_numpy(v).to(dev) for k, v in b.items()}


def cpu_baseline(budget_s=20.0, B=4, T=600, N=256, S=60):
    """oracle ('port') train step on the host cores, bounded sample of the same workload."""
    from oracle import ndt1 as O
    from oracle.step import CpuTrainer
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    cfg = O.make_config()
    p = O.init_params(cfg, 1)
    tr = CpuTrainer(cfg, p, total_steps=1000)
    batch, _ = make_batch(B, T, N, S, 41, "cpu", 0)
    tr.step(batch, train=True, seed=1)  # warm-up (BLAS threads, page faults)
    n, t0 = 0, time.perf_counter()
    while True:
        tr.step(batch, train=True, seed=2 + n)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 8:
            break
    return {"value": round(B * n / el, 3), "unit": "samples/s", "cores": int(cores), "kind": "port",
            "sample": f"{n} train steps (fwd+CTC+bwd+AdamW, dropout/noise on) of the numpy oracle, batch {B} x {T} bins x {N} ch, "
                      f"5-layer NDT1, fp32, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (weak scaling)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--bins", type=int, default=600)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--target-len", type=int, default=60)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local = local % max(1, ndev)          # (rehearsal: several ranks may share one GPU with NBCI_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("NBCI_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    from llm_bci_amd._lib import check, lib
    from llm_bci_amd.ndt1 import NDT1
    from llm_bci_amd.trainer import NativeTrainer

    torch.manual_seed(1)  # trainer.py:122
    over = {"encoder": {"embedder": {"n_channels": args.channels}}}
    model = NDT1(over, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype=args.dtype).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    total_steps = args.steps + args.warmup + 16   # OneCycle horizon covers warm-up + timed + the 3 profiling steps
    tr = NativeTrainer(model, lr=1e-3, wd=5e-5, eps=1e-8, scheduler="cosine", total_steps=total_steps, warmup_pct=0.0,
                       div_factor=25)
    _, batch = make_batch(args.batch, args.bins, args.channels, args.target_len, 41, dev, seed=rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        tr.train_step(batch, seed=100 + i + 100003 * rank)   # per-rank dropout / noise streams, as DDP ranks have
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tr.train_step(batch, seed=1000 + i + 100003 * rank)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    stats = tr.read_stats()

    roof = None
    if not args.no_roofline:
        # per-launch HIP-event timing of every GEMM over 3 more steps (own pass: events perturb the step).
        # Every rank runs the steps (they contain collectives); only rank 0 records and reports.
        l = lib()
        if rank == 0:
            check(l.nbci_profile_enable(1), "profile_enable")
        nprof = 3
        for i in range(nprof):
            tr.train_step(batch, seed=5000 + i + 100003 * rank)
        torch.cuda.synchronize()
    if rank == 0 and not args.no_roofline:
        out = (C.c_double * 24)()
        check(l.nbci_profile_collect(out), "profile_collect")
        check(l.nbci_profile_enable(0), "profile_enable")
        kinds = [(out[k * 3], out[k * 3 + 1], int(out[k * 3 + 2]), k) for k in range(8) if out[k * 3 + 2] > 0]
        tot_ms = sum(k[0] for k in kinds)
        ms, fl, cnt, kid = max(kinds)
        peak = PEAK_BF16_TFLOPS if kid >= 4 else PEAK_F32_TFLOPS
        ach = fl / ms / 1e9
        traffic = None   # HBM-side bytes per launch of that kernel from the committed PMC passes (profiles/), if present
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_gemm.json")))
            traffic = pmc["kind_avg_hbm_bytes_per_launch"].get(str(kid))
        except Exception:
            pass
        roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": traffic, "kernel": f"gemm_kernel<{KIND_NAMES[kid]}>", "launches_per_step": cnt // nprof,
                "avg_launch_us": round(1e3 * ms / cnt, 2), "flop_per_launch": round(fl / cnt / 1e9, 3),
                "share_of_gemm_time": round(ms / tot_ms, 3), "gemm_ms_per_step": round(tot_ms / nprof, 3),
                "all_gemm_tflops": round(sum(k[1] for k in kinds) / tot_ms / 1e9, 1)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        fps, Tp = fwd_flops_per_sample(args.bins, args.channels)
        gb = args.batch * world
        value = gb * args.steps / el
        res = {
            "metric": "train-step samples/sec (spike windows), NDT1-CTC", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "NDT1 CTC, default configs/ndt1.yaml (5 layers x 1024 hidden, 8 heads, stack 32/4), "
                                   f"{args.channels} ch x {args.bins} bins -> {Tp} tokens, target len {args.target_len}, "
                                   "recipe trainer_ctc_ndt1.yaml (dropout 0.2/0.4 + noise on, AdamW lr 1e-3 wd 5e-5, OneCycle cosine)",
                       "global_batch": gb, "per_gpu_batch": args.batch, "params": n_params,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "step": "fwd + CTC + bwd + grad all-reduce(mean) + fused AdamW + on-device PER"},
            "model_tflops_per_s": round(3 * fps * value / 1e12, 1),
            "train_loss_per_example": round(stats["loss"], 4), "train_PER": stats["PER"],
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline()
            res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
