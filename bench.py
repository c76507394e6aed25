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


RIDGE = PEAK_BF16_TFLOPS * 1e12 / 8e12     # FLOP per byte where the bf16 MFMA roof meets the 8 TB/s HBM roof


def kernel_table(l, nsteps, peak_tflops=PEAK_BF16_TFLOPS):
    """nbci_profile_collect_text -> one dict per kernel symbol, by descending time: launches per step, average launch duration (HIP
    events on the launch's own stream), ALGORITHMIC flops / bytes per launch, achieved rates, the bound its arithmetic intensity
    puts it under (MFMA above the ridge of 312 FLOP/B, else HBM) and the fraction of that roof; GEMMs carry both fractions."""
    from llm_bci_amd._lib import check
    buf = C.create_string_buffer(1 << 16)
    check(l.nbci_profile_collect_text(buf, 1 << 16), "profile_collect_text")
    rows = []
    for line in buf.value.decode().splitlines():
        sym, n, ms, fl, by = line.split("\t")
        n, ms, fl, by = int(n), float(ms), float(fl), float(by)
        if n == 0 or ms <= 0:
            continue
        us = 1e3 * ms / n
        r = {"kernel": sym, "launches_per_step": round(n / nsteps, 2), "avg_launch_us": round(us, 2), "ms_per_step": round(ms / nsteps, 4)}
        tf = fl / ms / 1e9 if fl > 0 else None           # TFLOP/s
        gb = by / ms / 1e6 if by > 0 else None           # GB/s
        if fl > 0:
            r["gflop_per_launch"] = round(fl / n / 1e9, 3); r["tflops"] = round(tf, 1); r["frac_mfma"] = round(tf / peak_tflops, 4)
        if by > 0:
            r["mbytes_per_launch"] = round(by / n / 1e6, 2); r["gbytes_per_s"] = round(gb, 1); r["frac_hbm"] = round(gb / 8000.0, 4)
        if fl > 0 and by > 0:
            r["flop_per_byte"] = round(fl / by, 1)
            r["bound"] = "mfma" if fl / by >= RIDGE else "hbm"
        elif fl > 0:
            r["bound"] = "mfma"
        elif by > 0:
            r["bound"] = "hbm"
        else:
            r["bound"] = "latency"
        r["frac"] = r.get("frac_mfma") if r["bound"] == "mfma" else r.get("frac_hbm")
        rows.append(r)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def pmc_traffic(symbol, run_cfg):
    """(bytes, source): HBM-side bytes per launch of `symbol` from the newest committed counter passes (profiles/r*_pmc_kernels.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH doubled per the gfx950 correction, KiB -> bytes) - OFFLINE-profiled,
    not measured in this run - and only when those passes ran THIS run's configuration (batch, shapes, dtypes); otherwise (None, why)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_kernels.json")))
    for f in reversed(files):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        cfg = j.get("config")
        if not cfg:
            continue
        if any(str(cfg.get(k)) != str(v) for k, v in run_cfg.items() if k in cfg):
            return None, f"offline counters ({os.path.basename(f)}) were taken for {cfg}, this run is {run_cfg}: not reported"
        for k, v in j.get("hbm_bytes_per_launch", {}).items():
            if k.replace(" ", "") == symbol.replace(" ", ""):
                return v, f"offline-profiled: {os.path.basename(f)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same configuration, side stream off)"
        return None, f"{os.path.basename(f)} has no entry for this symbol"
    return None, "no committed counter pass carries its configuration"


def roofline_from_profile(l, nsteps, run_cfg=None):
    rows = kernel_table(l, nsteps)
    if not rows:
        return None
    top = rows[0]
    tot = sum(r["ms_per_step"] for r in rows)
    gemm = [r for r in rows if r["kernel"].startswith("gemm")]
    mf = top["bound"] == "mfma" or "tflops" in top
    roof = {"bound": "mfma" if mf else "hbm", "achieved": top["tflops"] if mf else top["gbytes_per_s"],
            "peak": PEAK_BF16_TFLOPS if mf else 8000.0, "unit": "TFLOP/s" if mf else "GB/s",
            "frac": top["frac_mfma"] if mf else top["frac_hbm"], "traffic": None, "kernel": top["kernel"],
            "launches_per_step": top["launches_per_step"], "avg_launch_us": top["avg_launch_us"],
            "flop_per_launch": top.get("gflop_per_launch"), "algorithmic_mbytes_per_launch": top.get("mbytes_per_launch"),
            "frac_hbm_of_the_same_kernel": top.get("frac_hbm"), "intensity_flop_per_byte": top.get("flop_per_byte"),
            "ridge_flop_per_byte": round(RIDGE, 1), "share_of_kernel_time": round(top["ms_per_step"] / tot, 3),
            "kernel_ms_per_step": round(tot, 3), "gemm_ms_per_step": round(sum(r["ms_per_step"] for r in gemm), 3),
            "all_gemm_tflops": round(sum(r.get("gflop_per_launch", 0) * r["launches_per_step"] for r in gemm) /
                                     max(1e-9, sum(r["ms_per_step"] for r in gemm)), 1),
            "timing": "HIP events around every launch on its own stream, 3 steps after the timed windows, everything on ONE stream "
                      "(the timed windows overlap the weight gradients / AdamW on a second stream: NativeTrainer side_stream)", "per_kernel": rows}
    roof["traffic"], roof["traffic_source"] = pmc_traffic(top["kernel"], run_cfg or {})
    return roof


def other_model_points(l, dev, steps, residual_dtype="bf16"):
    """The other rows of BASELINE.json's configs on the same binary (not bench lines): iTransformer SSL at 668 / 1500 channels
    (configs[2], recipe trainer_ssl_itransformer.yaml: B = 16, T = 100), PatchTST at 1024 channels x 2050 bins with fp8 q/k/v
    (configs[4], B = 2 per GPU) and the BCI coupler at its real widths (configs[3]: projector 1024 -> 2048 -> 4096 + splice, forward
    and backward over B = 64 x 143 encoder tokens; the LLM itself is stock and not timed here). Each: full train step (coupler:
    forward + backward), median-free mean over `steps` steps after 2 warm-up steps, plus the launch-time share and roofline
    fraction of its top kernel from a 2-step HIP-event pass."""
    import gc
    from llm_bci_amd._lib import check
    from llm_bci_amd.trainer import NativeTrainer
    out = {}
    g = np.random.default_rng(0)

    def timed(step_fn, n):
        for i in range(2):
            step_fn(10 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step_fn(100 + i)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n
        l.nbci_profile_collect_text(C.create_string_buffer(1 << 16), 1 << 16)
        check(l.nbci_profile_enable(1), "profile_enable")
        for i in range(2):
            step_fn(200 + i)
        torch.cuda.synchronize()
        check(l.nbci_profile_enable(0), "profile_enable")
        rows = kernel_table(l, 2)
        tot = sum(r["ms_per_step"] for r in rows) or 1.0
        top = [{k: r.get(k) for k in ("kernel", "launches_per_step", "avg_launch_us", "bound", "frac", "frac_mfma", "frac_hbm")} |
               {"share_of_timed_kernels": round(r["ms_per_step"] / tot, 3)} for r in rows[:3]]
        return el, top

    def point(name, fn):
        try:
            out[name] = fn()
        except Exception as e:   # an extra point never takes the bench line down
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        gc.collect(); torch.cuda.empty_cache()

    def itr(N):
        from llm_bci_amd.itransformer import iTransformer
        torch.manual_seed(1)
        m = iTransformer({"encoder": {"embed_region": False}, "masker": {"main": {"active": True}}}, method_name="mlm", loss="poisson_nll",
                         log_input=True, compute_dtype="bf16", residual_dtype=residual_dtype).to(dev)
        tr = NativeTrainer(m, lr=1e-4, wd=0.01, eps=1e-8, scheduler="cosine", total_steps=100000, warmup_pct=0.15, div_factor=25, compute_per=False)
        B, T = 16, 100
        b = {"spikes": torch.from_numpy(g.poisson(0.5, (B, T, N)).astype(np.float32)).to(dev), "spikes_mask": torch.ones(B, T, dtype=torch.int64, device=dev),
             "spikes_timestamp": torch.arange(T, device=dev).repeat(B, 1)}
        el, top = timed(lambda sd: tr.train_step(b, seed=sd), steps)
        S, H, L = N + 1, 768, 5
        fwd = 2 * B * N * (T * H + H * H) + L * (2 * B * S * 12 * H * H + 4 * B * S * S * H) + 2 * B * S * (H * H + H * T)
        return {"workload": f"iTransformer mlm, B={B}, T={T}, {N} channels -> {S} tokens, 768 x 8 heads x 5 layers, bf16, masker + AdamW in the step",
                "ms_per_step": round(1e3 * el, 3), "samples_per_s": round(B / el, 1), "model_tflops_per_s": round(3 * fwd / el / 1e12, 1), "top_kernels": top}

    def ptst():
        from llm_bci_amd.patchtst import PatchTSTForSpikingActivity
        torch.manual_seed(1)
        B, Cn, T = 2, 1024, 2050
        m = PatchTSTForSpikingActivity({"encoder": {"num_input_channels": Cn, "context_length": T, "do_mask_input": False}}, compute_dtype="fp8",
                                       residual_dtype=residual_dtype, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True).to(dev)
        tr = NativeTrainer(m, lr=1e-3, wd=5e-5, eps=1e-8, scheduler="cosine", total_steps=100000, warmup_pct=0.0, div_factor=25)
        b = {"spikes": torch.from_numpy(g.standard_normal((B, T, Cn)).astype(np.float32)).to(dev), "spikes_mask": torch.ones(B, T, dtype=torch.int64, device=dev),
             "spikes_lengths": torch.full((B,), T, dtype=torch.int64, device=dev), "targets": torch.from_numpy(g.integers(1, 41, (B, 60))).to(dev),
             "targets_lengths": torch.full((B,), 60, dtype=torch.int64, device=dev)}
        el, top = timed(lambda sd: tr.train_step(b, seed=sd), max(3, steps // 2))
        P, D, F, L = 205, 256, 1024, 4
        M = B * Cn * P
        fwd = 2 * M * 10 * D + L * (2 * M * (4 * D * D + 2 * D * F) + 4 * B * Cn * P * P * D) + 2 * B * P * D * 41
        return {"workload": f"PatchTST ctc, B={B}, {Cn} ch x {T} bins -> 205 patches, d_model 256 x 8 heads x 4 layers, fp8 (block-scaled e4m3) q/k/v, rest bf16",
                "ms_per_step": round(1e3 * el, 3), "samples_per_s": round(B / el, 2), "model_tflops_per_s": round(3 * fwd / el / 1e12, 1), "top_kernels": top}

    def coupler():
        from llm_bci_amd import bci as BC
        torch.manual_seed(1)
        B, Tk, Hn, I, Hl, Lt = 64, 143, 1024, 2048, 4096, 24
        pj = BC.Projector(Hn, I, Hl, True, "relu").to(dev)
        w1, b1, w2, b2 = (t.detach() for t in pj.tensors())
        w1l, w2l = w1.bfloat16(), w2.bfloat16()
        x = torch.from_numpy(g.standard_normal((B * Tk, Hn)).astype(np.float32)).to(dev).bfloat16()
        text = torch.from_numpy(g.standard_normal((B, Lt, Hl)).astype(np.float32)).to(dev).bfloat16()
        amask = torch.ones(B, Lt, dtype=torch.int64, device=dev); valid = torch.ones(B, Tk, dtype=torch.int64, device=dev)
        split = torch.full((B,), 8, dtype=torch.int64, device=dev)
        gw1, gb1, gw2, gb2 = (torch.zeros_like(t, dtype=torch.float32) for t in (w1, b1, w2, b2))

        def step(_sd):
            y, h, dact = BC._proj_forward(x, w1l, b1, w2l, b2, pj.act, True)
            emb, _m, _t, sp = BC._splice_forward(text, y.view(B, Tk, Hl), amask, valid, None, split)
            _dt, dsp = BC._splice_backward(emb, sp, B, Lt, Tk, Hl, False)          # (the embeddings stand in for their gradient: same bytes)
            BC._proj_backward(dsp.view(B * Tk, Hl), x, h, dact, w1l, w2l, gw1, gb1, gw2, gb2, need_dx=True)
        el, top = timed(step, steps)
        fl = 3 * 2.0 * B * Tk * (Hn * I + I * Hl)
        return {"workload": f"BCI coupler alone: projector {Hn} -> {I} -> {Hl} (relu) + splice into {Lt} text embeddings, forward + backward, "
                            f"B={B} x {Tk} encoder tokens, bf16", "ms_per_step": round(1e3 * el, 3), "samples_per_s": round(B / el, 1),
                "model_tflops_per_s": round(fl / el / 1e12, 1), "top_kernels": top}

    point("iTransformer_N668_B16", lambda: itr(668))
    point("iTransformer_N1500_B16", lambda: itr(1500))
    point("PatchTST_C5_fp8_B2", ptst)
    point("BCI_coupler_B64", coupler)
    return out


def cpu_child(threads, B, timed, T=600, N=256, S=60):
    """ONE point of the CPU baseline, run in a fresh CPU-only process (`python bench.py --cpu-child threads B steps`, started by
    cpu_baseline() with OMP_NUM_THREADS / OMP_PROC_BIND=close / OMP_PLACES=cores in its environment): prints one JSON line."""
    from oracle import torch_step as TS
    from llm_bci_amd.ndt1 import NDT1
    torch.set_num_threads(threads)
    torch.manual_seed(1)
    m = NDT1({}, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype="fp32")   # CPU construction only: the reference-order init
    p0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    tr = TS.TorchCpuTrainer(p0, total_steps=1000)
    _, batch = make_batch(B, T, N, S, 41, "cpu", 0)
    tr.step(batch, train=True)                       # warm-up (thread pool, allocator, oneDNN primitives)
    t0 = time.perf_counter()
    for _ in range(timed):
        tr.step(batch, train=True)
    el = time.perf_counter() - t0
    fps, _ = fwd_flops_per_sample(T, N)
    print(json.dumps({"samples_per_s": round(B * timed / el, 3), "steps": timed, "seconds": round(el, 2), "threads": threads, "batch": B,
                      "gflops_per_s": round(3 * fps * B * timed / el / 1e9, 1), "torch_threads": torch.get_num_threads()}), flush=True)


def cpu_baseline(T=600, N=256, S=60):
    """The reference's train step on the host cores: the reference itself cannot travel to this box, so the timed
    thing is oracle/torch_step.py - a PyTorch-CPU restatement of the identical step (forward -> CTC sum -> autograd backward ->
    torch.optim.AdamW + OneCycleLR, fp32, dropout / noise ON as in the recipe), pinned to the reference's outputs by
    tests/test_oracle_torch_step.py. Every point runs in a FRESH CPU-only child process (subprocess, never exec) with its OpenMP threads
    pinned (OMP_PROC_BIND=close, OMP_PLACES=cores, OMP_NUM_THREADS = the count): thread counts {16, 32, one socket's cores} at B = 8
    and at the recipe's B = 64 (1 warm-up + 2 timed steps each). (Measured on the 2 x 64-core EPYC 9575F box: 16 and 32 pinned threads tie,
    a whole socket is SLOWER - these GEMMs have 1 144 .. 9 152 rows and the step is full of small ops; more threads only add synchronisation.) `value` = the best samples/s seen,
    `cores` = the threads that gave it; every point carries the GFLOP/s it reached. Bounded: about a minute of CPU work."""
    import subprocess
    from oracle import torch_step as TS
    host = TS.host_cpu_description()
    cores = int(host["physical_cores"])
    per_socket = max(1, cores // max(1, int(host["sockets"])))

    def run(B, threads, timed):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores",
                   HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", str(threads), str(B), str(timed)], env=env,
                             capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            return {"error": (out.stderr or out.stdout)[-300:], "threads": threads, "samples_per_s": 0.0}
        return json.loads(line[-1])

    cands = sorted({t for t in (16, 32, per_socket) if 1 <= t <= cores} or {cores})
    sweep = {f"B8_t{t}": run(8, t, 2) for t in cands}
    sweep.update({f"B64_t{t}": run(64, t, 2) for t in cands})      # the recipe's batch at every count too: more rows per GEMM scale further
    best_t = max((v for k, v in sweep.items() if k.startswith("B64")), key=lambda r: r["samples_per_s"])["threads"]
    b64 = sweep[f"B64_t{best_t}"]
    points = sweep
    best = max(points.values(), key=lambda r: r["samples_per_s"])
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "kind": "port", "gflops_per_s": best.get("gflops_per_s"),
            "sample": f"PyTorch-CPU restatement of the reference step (oracle/torch_step.py: fwd + CTC sum + autograd bwd + AdamW/OneCycle, fp32, "
                      f"dropout/noise on), {T} bins x {N} ch, 5-layer NDT1, torch {torch.__version__}; one fresh CPU-only process per point, OpenMP threads "
                      f"pinned (OMP_PROC_BIND=close, OMP_PLACES=cores); threads swept over {cands} at B=8 and B=64 (2 timed steps each); best of all points",
            "host_cpu": host, "points": points, "recipe_batch_64": b64,
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
    ap.add_argument("--residual-dtype", default="bf16", choices=["fp32", "bf16"], help="NDT1(residual_dtype=...): storage of the residual / gradient streams between kernels")
    ap.add_argument("--bins", type=int, default=600)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--target-len", type=int, default=60)
    ap.add_argument("--repeats", type=int, default=0, help="timed windows of --steps steps each; the median window is reported "
                    "(0 = as many as fill about 2 s of GPU time, 5..40)")
    ap.add_argument("--side-stream", default="auto", choices=["auto", "on", "off"], help="NativeTrainer(side_stream=...)")
    ap.add_argument("--no-extra-points", action="store_true", help="skip the B=8 / ragged / other-model points (N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-child", nargs=3, type=int, metavar=("THREADS", "BATCH", "STEPS"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_child:          # a point of cpu_baseline(), in its own CPU-only process (nothing below runs)
        cpu_child(*args.cpu_child)
        return

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
    model = NDT1(over, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype=args.dtype, residual_dtype=args.residual_dtype).to(dev)
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
    stats = tr.read_stats()                                  # (raises if a balanced grouped weight-gradient launch timed out: NativeTrainer.check_kernels)

    # The same step with the OTHER storage of the residual / gradient streams (NDT1(residual_dtype=...)), same run, every rank: "fp32" is
    # what the reference's bf16 autocast keeps in f32 (ndt1.py:325,328) - the reference-equivalent precision -, "bf16" the opt-in narrower
    # storage. Whichever `value` is, the other one is a top-level key of the line (value_reference_precision / value_bf16_streams).
    other_streams = None
    if args.dtype == "bf16":
        other_rd = "fp32" if args.residual_dtype == "bf16" else "bf16"
        torch.manual_seed(1)
        m2 = NDT1(over, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype="bf16", residual_dtype=other_rd).to(dev)
        tr2 = NativeTrainer(m2, lr=1e-3, wd=5e-5, eps=1e-8, scheduler="cosine", total_steps=1_000_000, warmup_pct=0.0, div_factor=25,
                            comm_dtype=args.comm_dtype, side_stream={"auto": "auto", "on": True, "off": False}[args.side_stream])
        tr_main, tr = tr, tr2
        for i in range(3):
            tr.train_step(batch, seed=i + 100003 * rank)
        w2 = measure(batch, 12000, 5)
        tr.read_stats()
        tr = tr_main
        e2 = w2[len(w2) // 2]
        other_streams = {"residual_dtype": other_rd, "ms_per_step": round(1e3 * e2 / args.steps, 3),
                         "value": round(per_gpu[args.scaling] * world * args.steps / e2, 2)}
        del m2, tr2

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
        # per-launch HIP-event timing of every kernel of the step over 3 more steps (own pass: events perturb the step).
        # Every rank runs the steps (they contain collectives); only rank 0 records and reports.
        l = lib()
        if rank == 0:
            l.nbci_profile_collect_text(C.create_string_buffer(1 << 16), 1 << 16)   # (drain anything recorded earlier)
            check(l.nbci_profile_enable(1), "profile_enable")
        nprof = 3
        side = tr.side_stream
        tr.side_stream = False     # one stream: a per-kernel roofline wants the kernel alone on the chip, not beside the side stream's
        for i in range(nprof):
            tr.train_step(batch, seed=5000 + i + 100003 * rank)
        torch.cuda.synchronize()
        tr.side_stream = side
    if rank == 0 and not args.no_roofline:
        check(l.nbci_profile_enable(0), "profile_enable")
        roof = roofline_from_profile(l, nprof, {"batch": per_gpu[args.scaling], "bins": args.bins, "channels": args.channels, "dtype": args.dtype,
                                                "residual_dtype": args.residual_dtype})
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
        extra["other_models"] = other_model_points(lib(), dev, max(5, args.steps // 2), args.residual_dtype)
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
                       "parallelism": f"dp{world}" if world > 1 else "single", "residual_dtype": args.residual_dtype,
                       "step": "fwd + CTC + bwd + grad all-reduce(mean) + fused AdamW + on-device PER"},
            "model_tflops_per_s": round(3 * fps * value / 1e12, 1),
            "train_loss_per_example": round(stats["loss"], 4), "train_PER": stats["PER"],
            "roofline": roof, "extra_points": extra, "other_scaling": other, "dp": dp,
        }
        # both stream storages at the top level: the reference-equivalent precision (f32 streams) and the opt-in bf16 streams
        if args.dtype == "bf16":
            mine = {"residual_dtype": args.residual_dtype, "ms_per_step": res["ms_per_step"], "value": res["value"]}
            f32s, b16s = (mine, other_streams) if args.residual_dtype == "fp32" else (other_streams, mine)
            res["value_reference_precision"] = f32s["value"]; res["ms_per_step_reference_precision"] = f32s["ms_per_step"]
            res["value_bf16_streams"] = b16s["value"]; res["ms_per_step_bf16_streams"] = b16s["ms_per_step"]
            res["precision_note"] = ("value = residual_dtype " + args.residual_dtype + "; value_reference_precision = the f32 residual / gradient streams "
                                     "the reference's bf16 autocast keeps (ndt1.py:325,328); both measured in this run, same batch and seeds")
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
