"""GPU: the C-ABI library is re-entrant (no global mutable state a second caller can trip over). Two host threads,
each with its own model (= its own plan and workspace) on its own stream, run train steps at the same time; ctypes releases the GIL
around every library call, so the host sides of the two callers interleave inside the library (plan bookkeeping, the per-device
LDS-attribute table, the grouped GEMM's per-stream scratch, the thread-local error string). Each thread's gradients must equal the
ones the same model computes alone."""
import threading

import numpy as np
import pytest
import torch

from test_ndt1_gpu import _model, _rand_batch, _to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _grads_of(m, bd, seed, n=1):
    g = torch.zeros(m._total, device=DEV)
    for i in range(n):
        g.zero_()
        m.train()
        m._run_forward(bd, want_grad=True, seed=seed)
        m._run_backward(g)
    return g


def test_two_threads_two_plans_two_streams():
    from llm_bci_amd._lib import lib
    import ctypes as C
    overs = [{"encoder": {"embedder": {"n_channels": 32}}},                                      # default widths: grouped (stream-K) weight gradients
             {"encoder": {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
                          "transformer": {"n_layers": 3, "hidden_size": 128, "n_heads": 1, "inter_size": 256}}}]
    batches = [_to_dev(_rand_batch(3, 200, 32, 6, 11, [200, 150, 97], [6, 4, 3])), _to_dev(_rand_batch(5, 60, 16, 5, 11, [60, 44, 34, 21, 60], [5, 4, 2, 3, 1]))]
    models = [_model(o, 11, dtype="bf16", seed=3 + i).to(DEV) for i, o in enumerate(overs)]
    alone = [_grads_of(m, b, 77).clone() for m, b in zip(models, batches)]
    torch.cuda.synchronize()
    out, errs = [None, None], []
    start = threading.Barrier(2)

    def work(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                start.wait()
                out[i] = _grads_of(models[i], batches[i], 77, n=25)      # 25 steps each, back to back, so the two really overlap
            s.synchronize()
        except Exception as e:   # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(2):
        d = (out[i] - alone[i]).abs().max().item()
        ref = alone[i].abs().max().item()
        assert ref > 0 and d <= 2e-5 * max(1.0, ref), (i, d, ref)
    n = C.c_int64(-1)
    assert lib().nbci_streamk_timeouts(C.byref(n)) == 0 and n.value == 0      # no owner of the balanced grouped launch gave up waiting
