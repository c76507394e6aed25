"""One-rank RCCL smoke on the GPU box (the 8-GPU run is the driver's): the process group comes up on backend "nccl" (= RCCL), an async
all-reduce of a gradient-bucket-sized tensor completes, and GradReducer / NativeTrainer take their world > 1 code paths with W = 1
forced to 2 buckets. Catches environment problems (librccl, HSA_ENABLE_IPC_MODE_LEGACY) before the driver's scaling run does."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x = torch.ones(41_056_560, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
w = dist.all_reduce(x, async_op=True); w.wait(); torch.cuda.synchronize()
print(f"rccl all_reduce 164 MB, 1 rank: {1e3 * (time.perf_counter() - t0):.2f} ms, sum ok {bool(x[0] == 1)}")
from bench import make_batch
from llm_bci_amd.ndt1 import NDT1
from llm_bci_amd.trainer import NativeTrainer
_, b = make_batch(8, 600, 256, 60, 41, dev, 0)


def run(comm, dp):
    """three train steps from the same initial weights; dp: the world > 1 code path (per-segment backward, bucketed async all-reduce over RCCL on
    the side stream, per-bucket AdamW) with the one-rank group standing in for W ranks (the all-reduce of one rank is the identity)"""
    torch.manual_seed(1)
    m = NDT1({}, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype="bf16", residual_dtype="bf16").to(dev)
    tr = NativeTrainer(m, total_steps=100, comm_dtype=comm, side_stream=True if dp else False)
    if dp:
        tr.reducer.world = 2; tr.world = 1
    losses = []
    for i in range(3):
        loss, _ = tr.train_step(b, seed=i)
        losses.append(float(loss.sum()))
    torch.cuda.synchronize()
    st = tr.read_stats()
    return losses, tr.reducer.last_buckets, tr.opt_step, st


ref, _, _, _ = run("fp32", False)
print(f"one stream, no exchange: losses {[round(x, 3) for x in ref]}")
for comm in ("fp32", "bf16"):
    losses, buckets, steps, st = run(comm, True)
    print(f"DP code path over RCCL (1 rank), comm {comm}: losses {[round(x, 3) for x in losses]}, buckets {buckets}, opt steps {steps}")
    assert buckets >= 2 and steps == 3
    if comm == "fp32":     # same kernels, same operands, identity exchange: the same losses
        assert all(abs(a - r) <= 1e-4 * abs(r) for a, r in zip(losses, ref)), (losses, ref)
    else:                  # bf16 buckets: the gradients are rounded once before AdamW
        assert all(abs(a - r) <= 2e-2 * abs(r) for a, r in zip(losses, ref)), (losses, ref)
dist.destroy_process_group()
print("RCCL_SMOKE OK")
