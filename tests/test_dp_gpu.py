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
