"""Two data-parallel ranks of the HIP trainer on the one GPU of the box (gloo carries the exchange; RCCL needs a GPU per
rank) against a single-process run: tools/dp_check.py, started as its own torch.distributed.run job."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_native_trainer_equals_single_process():
    env = dict(os.environ, NBCI_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tools", "dp_check.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DP_CHECK OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_rccl_comes_up_and_carries_the_data_parallel_code_path():
    """tools/rccl_smoke.py in its own process (a tool the driver never ran): the process group comes up on backend "nccl"
    (= RCCL; needs librccl and HSA_ENABLE_IPC_MODE_LEGACY=0), a 164 MB async all-reduce completes, and NativeTrainer's world > 1 path
    (per-segment backward on two streams, bucketed async all-reduce, per-bucket AdamW) runs over it with fp32 and bf16 buckets; its
    losses equal the one-stream run's. One rank: the 8-GPU run is the driver's - this catches bring-up bugs before it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_smoke.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SMOKE OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rccl all_reduce 164 MB" in r.stdout and "comm bf16" in r.stdout


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus2_refuses_on_a_one_gpu_box():
    """`python bench.py --gpus 2` with one GPU visible and RCCL asked for must fail loudly, never print a 1-GPU number."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two GPUs: the refusal does not apply")
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0, r.stdout[-2000:]
    assert "n_gpus" not in r.stdout and "only 1 GPU" in r.stderr, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_self_launches_its_ranks_and_reports_both_scaling_modes():
    """gloo rehearsal on the one GPU: bench.py starts its own two ranks, prints n_gpus 2, the weak-scaling value, the strong-scaling
    point (global batch split over the ranks: trainer.py:77-80) and the exchange bookkeeping."""
    import json
    r = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2", "--no-roofline"],      # the shipped defaults: 64 per rank (weak), global 64 (strong)
               {"NBCI_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 128 and d["config"]["per_gpu_batch"] == 64
    o = d["other_scaling"]
    assert o["scaling"] == "strong" and o["global_batch"] == 64 and o["per_gpu_batch"] == 32 and o["value"] > 0
    assert d["dp"]["ranks"] == 2 and d["dp"]["allreduce_bytes_per_step"] == 4 * d["config"]["params_padded"]
    assert d["dp"]["buckets_per_step"] >= 2 and "exposed_comm_ms" in d["dp"]["weak"] and "exposed_comm_ms" in d["dp"]["strong"]
    # both stream storages are top-level keys of the line
    assert d["config"]["residual_dtype"] == "bf16" and d["value"] == d["value_bf16_streams"] and d["value_reference_precision"] > 0
    assert d["ms_per_step_reference_precision"] > 0 and d["cpu_baseline"] is None
