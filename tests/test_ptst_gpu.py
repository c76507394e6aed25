"""GPU: the PatchTST HIP path (ctc + mlm heads) through the C-ABI against (a) fixtures generated from the reference
(tests/golden/g_ptst_*.npz; fp32 path, the reference's patch masks replayed), (b) the numpy oracle with identical dropout
and patch-mask draws, (c) itself in bf16."""
import json

import numpy as np
import pytest
import torch

from oracle import patchtst as OP
from test_oracle_ptst_golden import load, ptst_batch, ptst_cfg, split_state

from conftest import measured

pytestmark = pytest.mark.gpu
DEV = "cuda"
STAT = ("running_mean", "running_var", "num_batches_tracked")


def _dev(batch):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in batch.items()}


def _model(fx_or_over, dtype="fp32", kwargs=None, extra=None):
    from llm_bci_amd.patchtst import PatchTSTForSpikingActivity
    if isinstance(fx_or_over, dict):
        over, kw = fx_or_over, dict(kwargs)
    else:
        over, kw = json.loads(str(fx_or_over["config_json"])), json.loads(str(fx_or_over["kwargs_json"]))
    kw.update(extra or {})
    torch.manual_seed(1)
    return PatchTSTForSpikingActivity(over, compute_dtype=dtype, **kw)


def _grads_of(m, batch, seed=None):
    m.train()
    loss, preds = m._run_forward(batch, want_grad=True, seed=seed)
    g = torch.zeros_like(m._flat)
    m._run_backward(g)
    torch.cuda.synchronize()
    return loss, preds, {n: g[o:o + k].view(s).cpu().numpy() for (n, o, k, s, _sg) in m._layout}


@pytest.mark.parametrize("name", ["g_ptst_tiny", "g_ptst_tiny_ov", "g_ptst_tiny_mlm", "g_ptst_tiny_mlm_rate"])
def test_fp32_matches_reference_golden_tiny(name):
    from llm_bci_amd.trainer import NativeTrainer
    fx = load(name)
    mlm = json.loads(str(fx["kwargs_json"]))["method_name"] == "mlm"
    m = _model(fx).to(DEV)
    for k, v in m.state_dict().items():
        assert np.array_equal(v.float().cpu().numpy(), fx["w0:" + k]), k          # reference-order init is bit-equal
    batch = _dev(ptst_batch(fx))

    def mask_of(tag):
        return torch.from_numpy(fx[tag + "_raw_mask"]) if mlm else None

    m.eval()
    m.mask_override = mask_of("eval0")
    with torch.no_grad():
        out = m(**batch)
    np.testing.assert_allclose(out.preds.cpu().numpy(), fx["eval0_preds"], atol=1e-3)     # north_star tolerance (fp32 path)
    np.testing.assert_allclose(float(out.loss), float(fx["eval0_loss"]), rtol=1e-4)
    assert int(out.n_examples) == int(fx["eval0_n_examples"])
    if mlm:
        assert np.array_equal(out.mask.cpu().numpy(), fx["eval0_mask"])
        assert np.array_equal(out.patch_input.cpu().numpy(), fx["patch_input"])
    # eval mode must not touch the BatchNorm statistics
    for k, v in m.state_dict().items():
        if k.endswith(STAT):
            assert np.array_equal(v.float().cpu().numpy(), fx["w0:" + k]), k
    # two AdamW + OneCycle steps of the native trainer (stochastic ops off, the reference's masks replayed)
    tr = NativeTrainer(m, lr=1e-3, wd=5e-5, eps=1e-8, scheduler="cosine", total_steps=100, warmup_pct=0.0, div_factor=25, compute_per=not mlm)
    p0 = m._flat.clone()
    for s in range(2):
        m.mask_override = mask_of(f"step{s}")
        if s == 0:
            loss, _, g = _grads_of(m, batch)
            for k in g:
                ref = fx["grad:" + k]
                np.testing.assert_allclose(g[k], ref, atol=2e-6 + 1e-3 * np.abs(ref).max(), err_msg=k)
            # the probe forward above advanced the running statistics once: restore them for the trainer run
            sd = {k: torch.from_numpy(fx["w0:" + k]) for k in m.state_dict() if k.endswith(STAT)}
            sd = {k: (v.long() if k.endswith("num_batches_tracked") else v) for k, v in sd.items()}
            m.load_state_dict(sd, strict=False)
            assert torch.equal(m._flat, p0)
        loss, preds = tr.train_step(batch)
        np.testing.assert_allclose(float(loss.sum()), float(fx[f"step{s}_loss"]), rtol=2e-4)
        np.testing.assert_allclose(preds.cpu().numpy(), fx[f"step{s}_preds"], atol=1e-3)
    torch.cuda.synchronize()
    for k, v in m.state_dict().items():
        got, ref = v.float().cpu().numpy(), fx["w2:" + k]
        if k.endswith(STAT):
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6, err_msg=k)        # running stats + counters after 2 steps
        elif not k.endswith("k_proj.bias"):
            d = np.abs(got - ref)
            assert (d > 3e-5).mean() <= 0.05 and d.max() <= 2.1e-3, (k, d.max())
    m.eval()
    m.mask_override = mask_of("eval2")
    with torch.no_grad():
        out = m(**batch)
    np.testing.assert_allclose(out.preds.cpu().numpy(), fx["eval2_preds"], atol=2e-3)     # eval reads the running statistics


def test_fp32_matches_reference_golden_c5_shapes_and_bf16_close():
    fx = load("g_ptst_c5")
    batch = _dev(ptst_batch(fx))
    m = _model(fx).to(DEV)
    for k, v in m.state_dict().items():
        assert np.array_equal(v.float().cpu().numpy().reshape(-1)[fx["w0idx:" + k]], fx["w0val:" + k]), k
    m.eval()
    with torch.no_grad():
        out = m(**batch)
    p32 = out.preds.cpu().numpy()
    np.testing.assert_allclose(p32[..., ::3, :], fx["eval0_preds"], atol=1e-3)
    np.testing.assert_allclose(float(out.loss), float(fx["eval0_loss"]), rtol=2e-4)
    loss, preds, g = _grads_of(m, batch)
    np.testing.assert_allclose(float(loss.sum()), float(fx["step0_loss"]), rtol=2e-4)
    np.testing.assert_allclose(preds.cpu().numpy()[..., ::3, :], fx["step0_preds"], atol=1e-3)
    for k in g:
        ref = fx["gval:" + k]
        got = g[k].reshape(-1)[fx["gidx:" + k]]
        np.testing.assert_allclose(got, ref, atol=2e-6 + 3e-3 * max(np.abs(ref).max(), fx["gsum:" + k][1] / g[k].size), err_msg=k)
    for streams in ("fp32", "bf16"):   # the residual / gradient streams stored in f32 (the default), or in bf16 (opt-in)
        mb = _model(fx, dtype="bf16", extra={"residual_dtype": streams}).to(DEV)
        lb, pb, gb = _grads_of(mb, batch)
        measured(f"ptst.c5_golden.{streams}_streams.pred", np.abs(pb.cpu().numpy() - preds.cpu().numpy()).max())
        np.testing.assert_allclose(float(lb.sum()), float(loss.sum()), rtol=2e-2)
        for k in g:
            if k.endswith("k_proj.bias"):
                continue
            num, den = np.abs(gb[k] - g[k]).sum(), np.abs(g[k]).sum() + 1e-6
            measured(f"ptst.c5_golden.{streams}_streams.grad_l1_rel", num / den)


@pytest.mark.parametrize("method,dtype", [("ctc", "fp32"), ("mlm", "fp32"), ("mlm", "bf16"), ("ctc", "bf16"), ("mlm", "bf16/f32 streams")])
def test_train_mode_dropout_and_random_mask_match_oracle(method, dtype):
    enc = {"num_input_channels": 7, "context_length": 64, "patch_length": 8, "patch_stride": 4, "num_hidden_layers": 2, "d_model": 32,
           "num_attention_heads": 2, "ffn_dim": 64, "attention_dropout": 0.3, "ff_dropout": 0.4, "path_dropout": 0.1, "positional_dropout": 0.1,
           "do_mask_input": True, "random_mask_ratio": 0.4}
    kw = dict(method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True) if method == "ctc" else dict(method_name="mlm", loss="poisson_nll", log_input=True)
    extra = {"residual_dtype": "fp32" if (dtype.endswith("f32 streams") or dtype == "fp32") else "bf16"}   # ("bf16": the opt-in bf16 streams)
    dtype = dtype.split("/")[0]
    m = _model({"encoder": enc}, dtype=dtype, kwargs=kw, extra=extra).to(DEV)
    st = {k: v.detach().float().cpu().numpy().copy() for k, v in m.state_dict().items()}
    p = {k: v for k, v in st.items() if not k.endswith(STAT) and not k.endswith("position_enc")}
    p["encoder.encoder.positional_encoder.position_enc"] = st["encoder.encoder.positional_encoder.position_enc"]
    bufs = {k: v for k, v in st.items() if k.endswith(STAT)}
    g = np.random.default_rng(2)
    B, T, C = 4, 64, 7
    lens = [64, 50, 41, 64]
    spikes = (g.poisson(0.8, (B, T, C)) if method == "mlm" else g.standard_normal((B, T, C))).astype(np.float32)
    smask = np.zeros((B, T), np.int64)
    for b, L in enumerate(lens):
        spikes[b, L:] = 0; smask[b, :L] = 1
    batch = dict(spikes=spikes, spikes_mask=smask, spikes_lengths=np.array(lens))
    if method == "ctc":
        batch["targets"] = g.integers(1, 11, (B, 5)).astype(np.int64); batch["targets_lengths"] = np.array([5, 4, 3, 5])
    cfg = OP.make_config(**enc, method=method, vocab=11)
    seed = 777
    loss, preds, gh = _grads_of(m, _dev(batch), seed=seed)
    out, cache, nb = OP.forward(cfg, p, bufs, batch, train=True, seed=seed)
    go = OP.backward(cache)
    tol = 1e-3 if dtype == "fp32" else 0.1
    np.testing.assert_allclose(preds.cpu().numpy(), out["preds"], atol=tol)
    np.testing.assert_allclose(float(loss.sum()), float(out["loss"]), rtol=1e-4 if dtype == "fp32" else 3e-2)
    if method == "mlm":
        assert np.array_equal(m.last_mask.cpu().numpy(), out["mask"]) and int(m.last_n_examples) == int(out["n_examples"]) > 0
    for k in go:
        if dtype == "fp32":
            np.testing.assert_allclose(gh[k], go[k], atol=2e-6 + 2e-3 * np.abs(go[k]).max(), err_msg=k)
        elif not k.endswith("k_proj.bias"):
            assert np.abs(gh[k] - go[k]).sum() / (np.abs(go[k]).sum() + 1e-6) < 0.1, k
    sd = m.state_dict()
    for k, v in nb.items():       # running statistics after one train-mode forward
        np.testing.assert_allclose(sd[k].float().cpu().numpy(), np.asarray(v, np.float32), rtol=2e-3 if dtype == "bf16" else 1e-4, atol=1e-5, err_msg=k)


def test_autograd_bridge_checkpoint_roundtrip_and_errors(tmp_path):
    from llm_bci_amd.patchtst import PatchTSTForSpikingActivity
    fx = load("g_ptst_tiny")
    m = _model(fx).to(DEV)
    batch = _dev(ptst_batch(fx))
    m.train()
    out = m(**batch)
    out.loss.backward()
    for k, p in m.named_parameters():
        if not p.requires_grad:
            assert k.endswith("position_enc") and p.grad is None
            continue
        ref = fx["grad:" + k]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2e-6 + 1e-3 * np.abs(ref).max(), err_msg=k)
    m.save_checkpoint(str(tmp_path))
    over = json.loads(str(fx["config_json"]))
    over["encoder"]["from_pt"] = str(tmp_path); over["decoder"] = {"from_pt": str(tmp_path)}
    torch.manual_seed(9)
    m2 = PatchTSTForSpikingActivity(over, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32").to(DEV)
    a, b = m.state_dict(), m2.state_dict()
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    with pytest.raises(Exception, match="does not support"):
        PatchTSTForSpikingActivity({"encoder": {"channel_attention": True}}, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    with pytest.raises(Exception, match="not implemented"):
        PatchTSTForSpikingActivity({}, method_name="autoregressive")
    with pytest.raises(Exception):
        m(torch.zeros(1, 45, 6), torch.ones(1, 45, dtype=torch.int64), torch.tensor([45]))      # CPU tensors: no fallback


def test_hidden_out_is_f32_last_hidden_state_whatever_the_stream_storage():
    """nbci_ptst_io.hidden_out is (B, C, P, D) f32 (the reference's last_hidden_state): with bf16 streams it is a widening copy of the bf16
    stream; it must match the f32-stream run of the same bf16 model within bf16 rounding of values of order one, and the fp32 model closely."""
    enc = {"num_input_channels": 5, "context_length": 64, "patch_length": 8, "patch_stride": 4, "num_hidden_layers": 2, "d_model": 32,
           "num_attention_heads": 2, "ffn_dim": 64, "do_mask_input": False}
    kw = dict(method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    g = np.random.default_rng(3)
    B, T, C = 3, 64, 5
    batch = _dev(dict(spikes=g.standard_normal((B, T, C)).astype(np.float32), spikes_mask=np.ones((B, T), np.int64),
                      spikes_lengths=np.array([64, 64, 64]), targets=g.integers(1, 11, (B, 4)).astype(np.int64), targets_lengths=np.array([4, 3, 4])))
    outs = {}
    for name, dtype, extra in (("fp32", "fp32", None), ("bf16/f32", "bf16", {"residual_dtype": "fp32"}), ("bf16/bf16", "bf16", {"residual_dtype": "bf16"})):
        m = _model({"encoder": enc}, dtype=dtype, kwargs=kw, extra=extra).to(DEV)
        m.eval()
        P = m._ccfg_P if hasattr(m, "_ccfg_P") else 1 + (T - 8) // 4
        h = torch.full((B, C, P, 32), float("nan"), device=DEV)
        with torch.no_grad():
            m._run_forward(batch, want_grad=False, hidden_out=h)
        torch.cuda.synchronize()
        assert h.dtype == torch.float32 and torch.isfinite(h).all(), name
        outs[name] = h.cpu().numpy()
    assert np.abs(outs["bf16/bf16"] - outs["bf16/f32"]).max() <= 2.0 ** -7 * max(1.0, np.abs(outs["bf16/f32"]).max())
    assert np.abs(outs["bf16/f32"] - outs["fp32"]).max() < 0.08
