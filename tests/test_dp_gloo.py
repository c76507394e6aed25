"""world_size-2 data parallelism on CPU (gloo): batch sharding (split_batches=True), bucketed
async all-reduce over the flat gradient layout, DDP-mean folded into AdamW, scalar-stat reduce.
The model math here is the numpy oracle (the HIP kernels need a GPU); what is under test is
llm_bci_amd.dp + the flat layout/segments of llm_bci_amd.ndt1 — the code bench.py runs at N>1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ndt1 as O
from oracle.step import CpuTrainer

OVER = {"encoder": {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
                    "transformer": {"n_layers": 2, "hidden_size": 32, "n_heads": 2, "inter_size": 48}}}
CFG = dict(n_channels=16, input_dim=16, stack_size=4, stack_stride=2, hidden=32, n_layers=2, n_heads=2, inter=48, vocab=11,
           max_F=64)


def _batch():
    g = np.random.default_rng(0)
    B, T = 4, 30
    return dict(spikes=g.standard_normal((B, T, 16)).astype(np.float32), spikes_mask=np.ones((B, T), np.int64),
                spikes_timestamp=np.tile(np.arange(T), (B, 1)), spikes_lengths=np.full(B, T, np.int64),
                targets=g.integers(1, 11, (B, 5)).astype(np.int64), targets_lengths=np.array([5, 4, 3, 2], np.int64))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ga, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from llm_bci_amd.dp import GradReducer, reduce_stats, shard_batch
    from llm_bci_amd.ndt1 import NDT1
    torch.manual_seed(1)
    m = NDT1(OVER, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32")
    p0 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    full = {k: torch.from_numpy(v) for k, v in _batch().items()}
    mine = {k: v.numpy() for k, v in shard_batch(full, rank, world).items()}
    assert mine["spikes"].shape[0] == 2 and np.array_equal(mine["spikes"], _batch()["spikes"][rank * 2:(rank + 1) * 2])
    red = GradReducer(m._segments, min_bucket_elems=4096)   # small buckets -> several all-reduces in flight
    flat = torch.zeros(m._total)

    def reduce_fn(gd):
        flat.zero_()
        for (name, off, numel, shape, _seg) in m._layout:
            flat[off:off + numel] = torch.from_numpy(gd[name].reshape(-1))
        for seg in range(len(m._segments) - 1, 0, -1):      # backward order: head first, embedder last
            red.segment_done(flat, seg)
        b0, e0 = m._segments[0]                             # the embedder goes in two ranges, as NativeTrainer.train_step does
        assert b0 < m._embed_split < e0
        red.range_done(flat, m._embed_split, e0)
        red.range_done(flat, b0, m._embed_split)
        red.finish(flat)
        return {name: flat[off:off + numel].view(shape).numpy().copy() for (name, off, numel, shape, _s) in m._layout}

    tr = CpuTrainer(O.make_config(**CFG), p0, total_steps=10, ga=ga, world=world)
    stats = torch.zeros(2, dtype=torch.float64)
    for s in range(3):
        out = tr.step(mine, train=False, reduce_fn=reduce_fn)
        stats[0] += float(out["loss"]); stats[1] += mine["spikes"].shape[0]
    reduce_stats(stats)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), stats=stats.numpy(), **{"w:" + k: v for k, v in tr.p.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ga", [1, 2])
def test_two_rank_dp_equals_ddp_mean_semantics(tmp_path, ga):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ga, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(tmp_path, "dp.npz"))
    # single-process emulation of DDP over the same global batch: grads of the sum-loss over the
    # full batch, divided by world (DDP averages, the loss is a sum)
    from llm_bci_amd.ndt1 import NDT1
    torch.manual_seed(1)
    m = NDT1(OVER, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32")
    p0 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    ref = CpuTrainer(O.make_config(**CFG), p0, total_steps=10, ga=ga, world=2)
    loss = 0.0
    for s in range(3):
        loss += float(ref.step(_batch(), train=False)["loss"])
    np.testing.assert_allclose(got["stats"], [loss, 12.0], rtol=1e-5)
    for k, v in ref.p.items():
        if k.endswith("attn.key.bias"):
            continue
        d = np.abs(got["w:" + k] - v)
        assert (d > 2e-5).mean() < 0.01 and d.max() < 4e-3, (k, d.max())


def test_segments_cover_flat_buffer_contiguously():
    from llm_bci_amd.ndt1 import NDT1
    m = NDT1(OVER, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    segs = m._segments
    assert segs[0][0] == 0 and segs[-1][1] == m._total
    for a, b in zip(segs[:-1], segs[1:]):
        assert a[1] == b[0]
    for (name, off, numel, shape, seg) in m._layout:
        assert segs[seg][0] <= off and off + numel <= segs[seg][1], name
        assert off % 8 == 0
    # q,k,v weights / biases are adjacent so they run as one [3H][H] GEMM
    lay = {n: (o, k) for (n, o, k, _s, _g) in m._layout}
    q, k_, v = (lay[f"encoder.layers.0.attn.{n}.weight"] for n in ("query", "key", "value"))
    assert q[0] + q[1] == k_[0] and k_[0] + k_[1] == v[0]
    qb, kb, vb = (lay[f"encoder.layers.0.attn.{n}.bias"] for n in ("query", "key", "value"))
    assert qb[0] + qb[1] == kb[0] and kb[0] + kb[1] == vb[0]


# ------------------------------------------------------------------------------------------------------------------------
# BCI (BASELINE configs[3]): encoder + coupler + LoRA gradients through the SAME reduce schedule NativeTrainer runs on the GPU
# ------------------------------------------------------------------------------------------------------------------------
def _bci_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformers import AutoModelForCausalLM, LlamaConfig
    from llm_bci_amd.bci import BCI
    from llm_bci_amd.trainer import NativeTrainer
    torch.manual_seed(1)
    llm = AutoModelForCausalLM.from_config(LlamaConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                                                       num_attention_heads=4, num_key_value_heads=4))
    llm = BCI._add_lora(llm, dict(r=4, alpha=8, dropout=0.0, target_modules=["q_proj", "v_proj"], modules_to_save=[]))
    m = BCI({"projector": {"stacking": 1, "inter_size": 48}, "ndt1": OVER}, llm=llm, compute_dtype="fp32")
    calls = []
    segs = m._segments

    def fake_backward(grads, seg_hi, seg_lo, embed_part=0):
        """stands in for the HIP/LLM backward (no GPU here): writes rank- and position-dependent values into exactly the range the
        real call would finish, and records the call order."""
        calls.append((seg_hi, seg_lo, embed_part))
        for seg in range(seg_hi, seg_lo - 1, -1):
            b, e = segs[seg]
            if seg == 0 and embed_part == 1:
                b = m._embed_split
            elif seg == 0 and embed_part == 2:
                e = m._embed_split
            grads[b:e] += (rank + 1) * torch.arange(b, e, dtype=torch.float32) * 1e-3

    m._run_backward = fake_backward
    tr = NativeTrainer(m, total_steps=4, compute_per=False)
    tr.reducer.min_bucket = 2048               # small buckets: several all-reduces in flight
    tr._backward_and_reduce(sync=True)
    covered = torch.zeros(m._total, dtype=torch.int32)
    for b, e in tr.reducer.drain(tr.grads):
        covered[b:e] += 1
    n = len(segs)
    assert calls == [(s, s, 0) for s in range(n - 1, 0, -1)] + [(0, 0, 1), (0, 0, 2)], calls   # LLM adapters, projector, head..layer 0, embedder in 2 parts
    assert bool((covered == 1).all())                                                           # every element reduced exactly once
    want = sum(r + 1 for r in range(world)) * torch.arange(m._total, dtype=torch.float32) * 1e-3
    assert torch.allclose(tr.grads, want, rtol=1e-6)
    # an accumulation micro-step (no_sync, trainer.py:345): one backward call, nothing on the wire
    calls.clear()
    tr.grads.zero_()
    tr._backward_and_reduce(sync=False)
    assert calls == [(n - 1, 0, 0)] and not tr.reducer._works
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write(str(m._total))
    dist.barrier()
    dist.destroy_process_group()


def test_bci_two_rank_reduce_schedule_covers_encoder_coupler_and_adapters(tmp_path):
    mp.spawn(_bci_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(tmp_path, "ok"))


def _bf16_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from llm_bci_amd.dp import GradReducer
    segs = [(0, 5000), (5000, 9000), (9000, 16000)]
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(16000, generator=g)
    outs = {}
    for mode in ("fp32", "bf16"):
        flat = mine.clone()
        red = GradReducer(segs, min_bucket_elems=4000, comm_dtype=mode)
        for seg in (2, 1, 0):
            red.segment_done(flat, seg)
        covered = torch.zeros(16000, dtype=torch.int32)
        for b, e in red.drain(flat):
            covered[b:e] += 1
        assert bool((covered == 1).all())
        outs[mode] = flat
    # reference: the exact f32 sum over ranks
    ref = sum(torch.randn(16000, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    assert torch.allclose(outs["fp32"], ref, rtol=1e-6, atol=1e-6)
    err = (outs["bf16"] - ref).abs()
    bound = 2.0 ** -8 * (sum(torch.randn(16000, generator=torch.Generator().manual_seed(100 + r)).abs() for r in range(world)) + ref.abs())
    assert bool((err <= bound + 1e-6).all()), float((err - bound).max())     # bf16 rounding of each addend and of the sum, no more
    assert float(err.mean()) > 0.0                                             # (it really went through bf16)
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write("1")
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_allreduce_option_is_within_bf16_rounding_of_the_f32_sum(tmp_path):
    mp.spawn(_bf16_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(tmp_path, "ok"))
