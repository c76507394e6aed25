"""Pins oracle/torch_step.py (the PyTorch-CPU restatement bench.py times as `cpu_baseline`) to the reference's own outputs:
g_c1 / g_c2 fixtures (log-probs, loss, gradients, the weights after two AdamW + OneCycle steps). CPU only."""
import numpy as np
import pytest
import torch

from oracle import torch_step as TS
from test_oracle_golden import _regen, batch_of, load


@pytest.mark.parametrize("name,over,layers", [
    ("g_c1", {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}}}, 2),
    ("g_c2", {}, 5),
])
def test_torch_cpu_step_matches_reference(name, over, layers):
    fx = load(name)
    _m, p = _regen(over)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch_of(fx).items()}
    hp = dict(TS.default_hparams(), n_layers=layers, noise=False, embed_dropout=0.0, dropout=0.0)
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in p.items()}
    loss, lp, lens = TS.forward(P, batch, hp, train=False)
    assert np.array_equal(lens.numpy(), fx["token_lens"])
    np.testing.assert_allclose(lp.detach().numpy(), fx["eval_preds"], atol=1e-3)
    np.testing.assert_allclose(loss.item(), float(fx["eval_loss"]), rtol=2e-5)
    assert np.array_equal(lp.detach().argmax(-1).numpy(), fx["argmax"])
    loss.backward()
    for k, t in P.items():
        ref = fx["gval:" + k]
        got = t.grad.reshape(-1).numpy()[fx["gidx:" + k]]
        np.testing.assert_allclose(got, ref, atol=2e-4 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    # two optimizer steps in the reference's deterministic mode (dropouts 0, noise off): updated weights
    tr = TS.TorchCpuTrainer(p, hp, total_steps=100)
    for s in range(2):
        l, _ = tr.step(batch, train=True)
        np.testing.assert_allclose(l.item(), float(fx[f"loss_step{s}"]), rtol=5e-5)
    for k, t in tr.p.items():
        if k.endswith("attn.key.bias"):
            continue   # true gradient 0: Adam amplifies rounding noise to +-lr
        ref = fx["w2val:" + k]
        d = np.abs(t.detach().reshape(-1).numpy()[fx["w0idx:" + k]] - ref)      # sampled at the same indices as w0val
        assert (d > 2e-5).mean() <= 0.02 and d.max() <= 2.1e-3, (k, d.max())


def test_host_cpu_description_has_the_fields_the_bench_prints():
    d = TS.host_cpu_description()
    assert set(d) == {"model", "sockets", "physical_cores", "logical_cpus"} and d["physical_cores"] >= 1 and d["logical_cpus"] >= 1
