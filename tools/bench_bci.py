"""BASELINE configs[3]: BCI = NDT1 encoder (HIP) + coupler (HIP) + Llama-shaped causal LM with LoRA adapters (stock HF / PyTorch-ROCm,
NOT ours), one NativeTrainer step = forward + LLM autograd + coupler / encoder backward + (N > 1: bucketed all-reduce of encoder +
coupler + adapter gradients) + ONE fused AdamW over the joint flat buffer.

    python tools/bench_bci.py --llm 7b --batch 4 --steps 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_bci.py --llm 7b

Random-init weights of the named architecture (no checkpoints in this image), synthetic spikes / token ids. Besides the step time
it splits a step into encoder+coupler (our kernels) and the LLM (hipBLASLt / SDPA through PyTorch) with device events, because the
LLM dominates and is not what this repository accelerates.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LLMS = {  # LlamaConfig keyword arguments
    "tiny": dict(vocab_size=32000, hidden_size=256, intermediate_size=688, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4),
    "1b": dict(vocab_size=32000, hidden_size=2048, intermediate_size=5504, num_hidden_layers=16, num_attention_heads=16, num_key_value_heads=16),
    "7b": dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=32),
}
LORA = dict(r=8, alpha=32, dropout=0.2, modules_to_save=[],     # trainer_bci.yaml:54-58
            target_modules=["q_proj", "v_proj", "k_proj", "o_proj", "gate_proj", "up_proj", "down_proj"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--llm", default="7b", choices=sorted(LLMS))
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--text", type=int, default=24, help="prompt + sentence tokens per sample")
    ap.add_argument("--bins", type=int, default=600)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-lora", action="store_true", help="freeze the LLM entirely (freeze_llm: true)")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("NBCI_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from transformers import AutoModelForCausalLM, LlamaConfig
    from llm_bci_amd.bci import BCI
    from llm_bci_amd.trainer import NativeTrainer

    torch.manual_seed(1)
    t0 = time.perf_counter()
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float16)
    with torch.device(dev):
        llm = AutoModelForCausalLM.from_config(LlamaConfig(**LLMS[args.llm]))
    torch.set_default_dtype(old)
    if args.no_lora:
        for p in llm.parameters():
            p.requires_grad = False
    else:
        llm = BCI._add_lora(llm, LORA)
    m = BCI({"projector": {"stacking": 1, "inter_size": 2048, "bias": True, "act": "relu"}}, llm=llm, method_name="ctc", vocab_size=41,
            blank_id=0, zero_infinity=True, compute_dtype=args.dtype).to(dev)
    build_s = time.perf_counter() - t0
    tr = NativeTrainer(m, lr=1e-4, wd=0.0, total_steps=args.steps + args.warmup + 8, compute_per=False)
    g = np.random.default_rng(rank)
    B, T, Lt = args.batch, args.bins, args.text
    d = lambda a: torch.from_numpy(a).to(dev)
    batch = dict(input_ids=d(g.integers(0, 32000, (B, Lt)).astype(np.int64)), attention_mask=d(np.ones((B, Lt), np.int64)),
                 input_split=d(np.full(B, 8, np.int64)), spikes=d(g.standard_normal((B, T, 256)).astype(np.float32)),
                 spikes_mask=d(np.ones((B, T), np.int64)), spikes_timestamp=d(np.tile(np.arange(T), (B, 1))),
                 spikes_lengths=d(np.full(B, T, np.int64)))
    tg = g.integers(0, 32000, (B, Lt)).astype(np.int64); tg[:, :8] = -100
    batch["targets"] = d(tg)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        tr.train_step(batch, seed=i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tr.train_step(batch, seed=100 + i)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    # phase split (own pass): encoder + coupler forward | LLM forward + CE | LLM backward | coupler + encoder backward | AdamW
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    llm_fwd = m.llm.forward
    marks = {}

    def timed_llm(*a, **k):
        marks["a"] = torch.cuda.Event(enable_timing=True); marks["a"].record()
        return llm_fwd(*a, **k)

    m.llm.forward = timed_llm
    ev[0].record()
    m.train()
    m._run_forward(batch, want_grad=True, seed=7)
    ev[1].record()
    grads = tr.grads
    nseg = len(m._segments)
    top = nseg - 1
    m._run_backward(grads, top, m._native["pseg"] + (1 if m._native["lseg"] is not None else 0))   # LLM autograd (+ splice backward)
    if m._native["lseg"] is None:
        pass
    ev[2].record()
    m._run_backward(grads, min(top, m._native["pseg"]), 0)
    ev[3].record()
    torch.cuda.synchronize()
    m.llm.forward = llm_fwd
    grads.zero_()
    enc_fwd = ev[0].elapsed_time(marks["a"])
    llm_f = marks["a"].elapsed_time(ev[1])
    llm_b = ev[1].elapsed_time(ev[2]) if m._native["lseg"] is not None else None
    rest_b = ev[2].elapsed_time(ev[3])
    if rank == 0:
        n_tr = {"ndt1": m.ndt1._total, "projector": sum(p.numel() for p in m.projector.parameters()),
                "llm_trainable": sum(p.numel() for _n, p, _o in m._native["eentries"])}
        res = {"metric": "train-step samples/sec, BCI (NDT1 encoder + coupler + Llama LoRA)", "value": round(B * world * args.steps / el, 2),
               "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 2),
               "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"BCI: default NDT1 256 ch x {T} bins -> 143 tokens, projector 1024->2048->{LLMS[args.llm]['hidden_size']}, "
                                      f"Llama shape '{args.llm}' fp16 random init, LoRA r=8 on 7 projections x {LLMS[args.llm]['num_hidden_layers']} layers"
                                      if not args.no_lora else f"BCI, LLM '{args.llm}' frozen",
                          "per_gpu_batch": B, "text_tokens": Lt, "trainable_elements": n_tr, "flat_buffer_elements": m._total},
               "phase_ms": {"encoder+coupler forward (ours)": round(enc_fwd, 3), "LLM forward + CE (stock)": round(llm_f, 3),
                            "LLM backward + splice backward": None if llm_b is None else round(llm_b, 3),
                            "coupler + encoder backward (ours; incl. LLM backward when nothing in the LLM trains)": round(rest_b, 3)},
               "model_build_s": round(build_s, 1)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
