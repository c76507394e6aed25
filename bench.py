"""bench.py — train-step samples/sec of the HIP NDT1-CTC path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full train step of the reference's loop (trainer.py:332-362 semantics) over one
synthetic batch already resident in HBM: forward (recipe dropout + noise ON) -> CTC sum-loss ->
backward -> (N>1: overlapped RCCL all-reduce) -> fused AdamW + OneCycle -> zero_grad, plus the
on-device PER metric the reference computes every step. Workload = BASELINE.json configs[1]:
default configs/ndt1.yaml (5 layers x 1024, 41.06 M params), 256 ch x 600 bins, bf16 operands.
Run as `python bench.py --gpus N` with N > 1 and no torchrun around it, the script starts its own N ranks
(`python -m torch.distributed.run`, one per GPU) BEFORE touching the GPU and relays rank 0's JSON line; it exits non-zero
when fewer than N GPUs are visible or WORLD_SIZE disagrees with --gpus (never a silent 1-GPU number).
Scaling modes: `--scaling weak` (default): --batch is PER GPU (64 = the recipe's train_batch_size on every rank);
`--scaling strong`: the reference's own semantics (trainer.py:77-80 `split_batches=True`: configs/trainer_ctc_ndt1.yaml's
train_batch_size 64 is the GLOBAL batch, each of W ranks gets 64 / W). At N > 1 BOTH are measured in the one run: `value` is the
mode asked for, the other is in `other_scaling`. `dp` reports the ranks, all-reduce bytes per step and the exposed communication
(step time with the exchange minus the same step with it skipped).
Timing: W warm-up steps, then --repeats windows of EXACTLY K steps, each bracketed by barrier + device sync (max over ranks);
`value` / `ms_per_step` are the MEDIAN window's, `ms_per_step_min/max` the spread. Extra keys at N=1: `extra_points` (B = 8 and a
ragged-length batch), `roofline` (dominant GEMM, HIP events live), `cpu_baseline` (PyTorch-CPU restatement on the host cores).
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
    # 2 FLOP per multiply-add of every contraction; the attention core is two T' x T' x H matmuls per layer
    # (2 * T'^2 * H FLOP each), counted once - NOT inside the doubling (round 1 double-counted it: 12.33 instead of 11.909 GFLOP).
    return 2.0 * (13 * T * N + T * N * D + Tp * (S * D) * H + L * (4 * Tp * H * H + 2 * Tp * H * I) + Tp * H * V) + L * 4.0 * Tp * Tp * H, Tp


def make_batch(B, T, N, S, vocab, dev, seed, ragged=False):
    """synthetic batch in pad_collate_fn's layout (datasets.py:236-272). ragged: lengths uniform in [T/2, T], right-padded with
    zeros, masks / timestamps / target lengths to match (the longest sample keeps T so the padded shape is unchanged)."""
    g = np.random.default_rng(seed)
    b = dict(spikes=g.standard_normal((B, T, N)).astype(np.float32), spikes_mask=np.ones((B, T), np.int64),
             spikes_timestamp=np.tile(np.arange(T), (B, 1)), spikes_lengths=np.full(B, T, np.int64),
             targets=g.integers(1, vocab, (B, S)).astype(np.int64), targets_lengths=np.full(B, S, np.int64))
    if ragged:
        lens = g.integers(T // 2, T + 1, B); lens[0] = T
        for i, L in enumerate(lens):
            b["spikes"][i, L:] = 0; b["spikes_mask"][i, L:] = 0; b["spikes_timestamp"][i, L:] = 0
            b["targets_lengths"][i] = max(1, int(S * L / T)); b["targets"][i, b["targets_lengths"][i]:] = 0
        b["spikes_lengths"] = lens.astype(np.int64)
    return b, {k: torch.from_numpy(v).to(dev) for k, v in b.items()}


def cpu_baseline(budget_s=12.0, T=600, N=256, S=60):
    """The reference's train step on the host cores: the reference itself cannot travel to this box, so the timed
    thing is oracle/torch_step.py - a PyTorch-CPU restatement of the identical step (forward -> CTC sum -> autograd backward ->
    torch.optim.AdamW + OneCycleLR, fp32, dropout / noise ON as in the recipe), pinned to the reference's outputs by
    tests/test_oracle_torch_step.py - with torch.set_num_threads(physical cores), at B = 8 and at the recipe's B = 64, each for a
    bounded sample (about `budget_s` of CPU work). `value` is the better of the two. The numpy oracle's rate (round 1's
    baseline) is kept as a second field."""
    from oracle import torch_step as TS
    from llm_bci_amd.ndt1 import NDT1
    host = TS.host_cpu_description()
    cores = int(host["physical_cores"])
    torch.set_num_threads(cores)
    torch.manual_seed(1)
    m = NDT1({}, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype="fp32")   # CPU construction only: the reference-order init
    p0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    points = {}
    for B in (8, 64):
        tr = TS.TorchCpuTrainer(p0, total_steps=1000)
        _, batch = make_batch(B, T, N, S, 41, "cpu", 0)
        tr.step(batch, train=True)                       # warm-up (thread pool, allocator, oneDNN primitives)
        n, t0 = 0, time.perf_counter()
        while True:
            tr.step(batch, train=True)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 12:
                break
        points[f"B{B}"] = {"samples_per_s": round(B * n / el, 3), "steps": n, "seconds": round(el, 2)}
        del tr
    best = max(points, key=lambda k: points[k]["samples_per_s"])
    numpy_port = None
    try:   # round 1's figure, for continuity: the numpy oracle's step (BLAS threads as configured by the environment)
        from oracle import ndt1 as O
        from oracle.step import CpuTrainer
        cfg = O.make_config()
        ct = CpuTrainer(cfg, O.init_params(cfg, 1), total_steps=1000)
        bnp, _ = make_batch(4, T, N, S, 41, "cpu", 0)
        ct.step(bnp, train=True, seed=1)
        t0 = time.perf_counter(); ct.step(bnp, train=True, seed=2); ct.step(bnp, train=True, seed=3)
        numpy_port = round(8 / (time.perf_counter() - t0), 3)
    except Exception:
        pass
    return {"value": points[best]["samples_per_s"], "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"PyTorch-CPU restatement of the reference step (oracle/torch_step.py: fwd + CTC sum + autograd bwd + AdamW/OneCycle, fp32, "
                      f"dropout/noise on), {T} bins x {N} ch, 5-layer NDT1, torch {torch.__version__} with {cores} threads; best of {points}",
            "host_cpu": host, "points": points, "numpy_oracle_samples_per_s": numpy_port,
            "reference_eager_8vcpu_build_container": 5.2}


def feed_from_host(tr, args, dev, n_rows=256, windows=3):
    """K train steps per window, every batch collated on the host and uploaded while the previous step runs."""
    from llm_bci_amd.collate import DeviceFeeder, HostCollator, PinnedPool, item_from_row
    g = np.random.default_rng(11)
    T, N, S, B = args.bins, args.channels, args.target_len, args.batch
    lens = g.integers(T // 2, T + 1, n_rows)
    items = [item_from_row({"spikes": g.standard_normal((int(L), N)).astype(np.float32),
                            "targets": g.integers(1, 41, (max(1, int(S * L / T)),)).astype(np.int64)}) for L in lens]
    full = item_from_row({"spikes": g.standard_normal((T, N)).astype(np.float32), "targets": g.integers(1, 41, (S,)).astype(np.int64)})
    pad = {k: dict(dim=0, side="right", value=0, truncate=None, min_length=None)      # trainer_ctc_ndt1.yaml pad_dict
           for k in ("spikes", "spikes_mask", "spikes_timestamp", "targets", "targets_mask")}
    names = ["spikes", "spikes_mask", "spikes_timestamp", "spikes_lengths", "targets", "targets_lengths"]
    n_batches = 2 + windows * args.steps

    def sampler():
        for _ in range(n_batches):
            idx = g.integers(0, n_rows, B - 1)
            yield [full] + [items[i] for i in idx]      # one full-length row: the padded shape equals the resident point's

    pool = PinnedPool()
    feeder = DeviceFeeder(HostCollator(sampler(), names, pad, workers=4, depth=4, pool=pool), dev, pool=pool)
    for _ in range(2):
        b, _u = next(feeder)
        tr.train_step(b, seed=1)
    ws = []
    for w in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            b, _u = next(feeder)
            tr.train_step(b, seed=7000 + 100 * w + i)
        torch.cuda.synchronize()
        ws.append(time.perf_counter() - t0)
    w = sorted(ws)[len(ws) // 2]
    return {"ms_per_step": round(1e3 * w / args.steps, 3), "samples_per_s": round(B * args.steps / w, 1),
            "h2d_mbytes_per_step": round(B * T * N * 4 / 1e6, 1), "pinned_buffers_allocated": pool.allocated}


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks ourselves, before anything touches the GPU in this
    process (torch.cuda.device_count() does not initialise it), and leave with the launcher's exit code. The reference gets its
    ranks the same way, from `accelerate launch` (models/trainer.py:77-80,258-262)."""
    import socket
    import subprocess
    backend = os.environ.get("NBCI_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if backend == "nccl" and ndev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} asked for but only {ndev} GPU(s) are visible; refusing to report a {ndev}-GPU number as "
              f"{args.gpus} (rehearse the multi-rank path on one GPU with NBCI_DIST_BACKEND=gloo)", file=sys.stderr, flush=True)
        raise SystemExit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=64, help="global batch split over the ranks (strong scaling; the recipe's train_batch_size)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="which mode `value` reports (N > 1 measures both)")
    ap.add_argument("--comm-dtype", default="fp32", choices=["fp32", "bf16"], help="gradient all-reduce dtype (fp32 = the reference's DDP)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--bins", type=int, default=600)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--target-len", type=int, default=60)
    ap.add_argument("--repeats", type=int, default=0, help="timed windows of --steps steps each; the median window is reported "
                    "(0 = as many as fill about 2 s of GPU time, 5..40)")
    ap.add_argument("--side-stream", default="auto", choices=["auto", "on", "off"], help="NativeTrainer(side_stream=...)")
    ap.add_argument("--no-extra-points", action="store_true", help="skip the B=8 / ragged / other-model points (N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks (or run `python bench.py --gpus {args.gpus}` "
              "alone and let it start them)", file=sys.stderr, flush=True)
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("NBCI_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, {ndev} visible")
    local = local % max(1, ndev)          # (rehearsal: several ranks may share one GPU with NBCI_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    from llm_bci_amd._lib import check, lib
    from llm_bci_amd.ndt1 import NDT1
    from llm_bci_amd.trainer import NativeTrainer

    torch.manual_seed(1)  # trainer.py:122
    over = {"encoder": {"embedder": {"n_channels": args.channels}}}
    model = NDT1(over, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype=args.dtype).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    # OneCycle horizon: far beyond anything this script runs (the schedule only sets lr / beta1 scalars of the fused AdamW)
    tr = NativeTrainer(model, lr=1e-3, wd=5e-5, eps=1e-8, scheduler="cosine", total_steps=1_000_000, warmup_pct=0.0,
                       div_factor=25, comm_dtype=args.comm_dtype, side_stream={"auto": "auto", "on": True, "off": False}[args.side_stream])
    if args.global_batch % world:
        raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} ranks")
    per_gpu = {"weak": args.batch, "strong": args.global_batch // world}
    batches = {m: make_batch(per_gpu[m], args.bins, args.channels, args.target_len, 41, dev, seed=rank)[1] for m in per_gpu}
    batch = batches[args.scaling]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        tr.train_step(batch, seed=100 + i + 100003 * rank)   # per-rank dropout / noise streams, as DDP ranks have

    def timed_window(b, k, seed0):
        """EXACTLY k steps between two (barrier + device sync) brackets; max over ranks."""
        sync()
        t0 = time.perf_counter()
        for i in range(k):
            tr.train_step(b, seed=seed0 + i + 100003 * rank)
        sync()
        e = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([e], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        return e

    def measure(b, seed0, repeats):
        """sorted windows of EXACTLY --steps steps each. repeats = 0: as many as fill about 2 s of GPU time (so that a once-a-second
        utilisation sampler sees the GPU busy), between 5 and 40; every rank takes the count from rank 0's first window."""
        ws = [timed_window(b, args.steps, seed0)]
        n = repeats if repeats > 0 else int(min(40, max(5, 2.0 / max(ws[0], 1e-6))))
        ws += [timed_window(b, args.steps, seed0 + 1000 * (r + 1)) for r in range(n - 1)]
        return sorted(ws)

    windows = measure(batch, 1000, args.repeats)
    el = windows[len(windows) // 2]                          # the median window is the reported one
    stats = tr.read_stats()

    other = dp = None
    if world > 1:
        # the other scaling mode, same binary, same run
        om = "strong" if args.scaling == "weak" else "weak"
        for i in range(2):
            tr.train_step(batches[om], seed=300 + i + 100003 * rank)
        wo = measure(batches[om], 50000, 5)
        eo = wo[len(wo) // 2]
        gbo = per_gpu[om] * world
        other = {"scaling": om, "global_batch": gbo, "per_gpu_batch": per_gpu[om], "value": round(gbo * args.steps / eo, 2),
                 "unit": "samples/s", "ms_per_step": round(1e3 * eo / args.steps, 3), "ms_per_step_min": round(1e3 * wo[0] / args.steps, 3),
                 "ms_per_step_max": round(1e3 * wo[-1] / args.steps, 3), "repeats": len(wo)}
        # exposed communication: the same step with the gradient exchange skipped (every rank steps on its local gradients;
        # weights diverge across ranks from here on, which no later measurement depends on)
        exposed = {}
        tr.reducer.enabled = False
        for m2 in (args.scaling, om):
            tr.train_step(batches[m2], seed=400 + 100003 * rank)
            wl = measure(batches[m2], 60000, 5)
            t_comm = (el if m2 == args.scaling else eo) / args.steps
            exposed[m2] = {"ms_per_step_no_exchange": round(1e3 * wl[len(wl) // 2] / args.steps, 3),
                           "exposed_comm_ms": round(1e3 * (t_comm - wl[len(wl) // 2] / args.steps), 3)}
        tr.reducer.enabled = True
        esz = 2 if args.comm_dtype == "bf16" else 4
        dp = {"ranks": dist.get_world_size(), "backend": "rccl" if backend == "nccl" else backend, "comm_dtype": args.comm_dtype,
              "allreduce_bytes_per_step": int(model._total) * esz, "buckets_per_step": tr.reducer.last_buckets,
              "algorithm": "bucketed all-reduce(SUM) per backward segment, overlapped with backward; 1/W folded into AdamW", **exposed}

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
            for nm in ("r02_pmc_gemm.json", "r01_pmc_gemm.json"):
                f = os.path.join(ROOT, "profiles", nm)
                if os.path.exists(f):
                    traffic = json.load(open(f))["kind_avg_hbm_bytes_per_launch"].get(str(kid))
                    break
        except Exception:
            pass
        roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": traffic, "kernel": f"gemm_kernel<{KIND_NAMES[kid]}>", "launches_per_step": cnt // nprof,
                "avg_launch_us": round(1e3 * ms / cnt, 2), "flop_per_launch": round(fl / cnt / 1e9, 3),
                "share_of_gemm_time": round(ms / tot_ms, 3), "gemm_ms_per_step": round(tot_ms / nprof, 3),
                "all_gemm_tflops": round(sum(k[1] for k in kinds) / tot_ms / 1e9, 1)}
    extra = None
    if world == 1 and not args.no_extra_points:
        # Two more points of the same binary (not bench lines): SURVEY's small batch B = 8 (launch-latency bound) and the recipe
        # batch with RAGGED lengths (uniform in [T/2, T], right-padded), each the median of 3 windows.
        extra = {}
        for name, (B2, rg) in {"B8_full_length": (8, False), f"B{args.batch}_ragged": (args.batch, True)}.items():
            _, b2 = make_batch(B2, args.bins, args.channels, args.target_len, 41, dev, seed=7, ragged=rg)
            tr.train_step(b2, seed=3)
            tr.train_step(b2, seed=4)
            w = measure(b2, 9000, 5)[2]
            extra[name] = {"ms_per_step": round(1e3 * w / args.steps, 3), "samples_per_s": round(B2 * args.steps / w, 1)}
            if rg:
                extra[name]["mean_valid_fraction"] = round(float(b2["spikes_lengths"].float().mean().item()) / args.bins, 3)
        # the same ragged workload FED FROM THE HOST every step: a pool of rows in host memory -> background
        # collation into recycled pinned buffers (llm_bci_amd.collate.HostCollator) -> asynchronous H2D on a copy stream
        # (DeviceFeeder) -> train step; a fresh batch of the recipe's size each step, nothing resident.
        fed = feed_from_host(tr, args, dev)
        rs = extra[f"B{args.batch}_ragged"]["ms_per_step"]
        fed["vs_resident_ragged"] = round(fed["ms_per_step"] / rs, 3)
        extra[f"B{args.batch}_ragged_fed_from_host"] = fed
        tr.read_stats()
    if world > 1:
        dist.barrier()

    if rank == 0:
        fps, Tp = fwd_flops_per_sample(args.bins, args.channels)
        gb = per_gpu[args.scaling] * world
        value = gb * args.steps / el
        res = {
            "metric": "train-step samples/sec (spike windows), NDT1-CTC", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
            "repeats": len(windows), "ms_per_step_min": round(1e3 * windows[0] / args.steps, 3), "ms_per_step_max": round(1e3 * windows[-1] / args.steps, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "NDT1 CTC, default configs/ndt1.yaml (5 layers x 1024 hidden, 8 heads, stack 32/4), "
                                   f"{args.channels} ch x {args.bins} bins -> {Tp} tokens, target len {args.target_len}, "
                                   "recipe trainer_ctc_ndt1.yaml (dropout 0.2/0.4 + noise on, AdamW lr 1e-3 wd 5e-5, OneCycle cosine)",
                       "global_batch": gb, "per_gpu_batch": per_gpu[args.scaling], "params": n_params, "params_padded": int(model._total),
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "step": "fwd + CTC + bwd + grad all-reduce(mean) + fused AdamW + on-device PER"},
            "model_tflops_per_s": round(3 * fps * value / 1e12, 1),
            "train_loss_per_example": round(stats["loss"], 4), "train_PER": stats["PER"],
            "roofline": roof, "extra_points": extra, "other_scaling": other, "dp": dp,
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline()
            res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)   # a reported ratio, not a quality measure (see roofline)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
