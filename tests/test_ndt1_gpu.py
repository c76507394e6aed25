"""GPU parity of the HIP NDT1-CTC path (through the C-ABI) against
  (a) the golden fixtures produced by the reference itself (tests/golden), and
  (b) the numpy oracle on the same seeded inputs, including train mode with dropout + noise
      (the oracle mirrors the kernels' stateless RNG bit for bit).
Tolerances: fp32 path logits/log-probs <= 1e-3 abs (north_star), argmax / decode / PER bit-exact;
bf16 path: stated per test.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import metrics as OM
from oracle import ndt1 as O
from test_oracle_golden import batch_of, cfg_from_json, load

from conftest import measured

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(over, vocab, dtype="fp32", seed=1, streams="fp32"):
    """streams: NDT1(residual_dtype=...), the storage of the residual stream and its gradient stream between kernels"""
    from llm_bci_amd.ndt1 import NDT1
    torch.manual_seed(seed)
    return NDT1(over, method_name="ctc", vocab_size=vocab, blank_id=0, zero_infinity=True, compute_dtype=dtype, residual_dtype=streams)


STREAMS = ["fp32", "bf16"]   # bf16 path, residual / gradient streams stored in f32 (what bf16 autocast keeps in f32) or in bf16


def _to_dev(batch):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in batch.items()}


def _grads(model, batch, train, seed=7):
    """native (non-autograd) forward + full backward; returns loss vector, preds, {name: grad}"""
    model.train(train)
    loss, preds = model._run_forward(batch, want_grad=True, seed=seed)
    g = torch.zeros_like(model._flat)
    model._run_backward(g)
    torch.cuda.synchronize()
    out = {}
    for (name, off, numel, shape, _s) in model._layout:
        out[name] = g[off:off + numel].view(shape).cpu().numpy()
    return loss.cpu().numpy(), preds.cpu().numpy(), out


def _det_over(js):
    over = json.loads(js)
    e = over.setdefault("encoder", {})
    e.setdefault("smooth_and_noise", {})["noise"] = False
    e.setdefault("embedder", {})["dropout"] = 0.0
    e.setdefault("transformer", {})["dropout"] = 0.0
    return over


@pytest.mark.parametrize("name", ["g_tiny", "g_tiny_ctx", "g_tiny_rope", "g_tiny_factors", "g_tiny_factors_fix", "g_tiny_adapt", "g_tiny_tokens", "g_tiny_daytoken"])
def test_tiny_golden_fp32(name):
    fx = load(name)
    m = _model(_det_over(str(fx["config_json"])), 11)
    sd = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w0:")}
    own = m.state_dict()
    assert set(own) == set(sd)                      # the reference's state-dict keys, nothing more or less
    for k, v in sd.items():                         # reference-order initialisation: bit-equal under torch.manual_seed(1)
        assert torch.equal(own[k], v), k
    m.load_state_dict(sd)
    m.to(DEV)
    batch = _to_dev(batch_of(fx))
    m.eval()
    with torch.no_grad():
        out = m(**batch)
    torch.cuda.synchronize()
    preds = out.preds.cpu().numpy()
    np.testing.assert_allclose(preds, fx["eval_preds"], atol=1e-3)
    np.testing.assert_allclose(out.loss.item(), float(fx["eval_loss"]), rtol=1e-4)
    assert int(out.n_examples) == int(fx["n_examples"])
    assert np.array_equal(m.last_argmax.cpu().numpy(), fx["argmax"])
    loss, _, g = _grads(m, batch, train=True)
    np.testing.assert_allclose(loss.sum(), float(fx["train_loss"]), rtol=1e-4)
    for k, gv in g.items():
        ref = fx["grad:" + k]
        np.testing.assert_allclose(gv, ref, atol=5e-4 * max(1.0, float(np.abs(ref).max())), err_msg=k)


def test_endtoend_method_is_the_ctc_branch():
    """ndt1.py:488,498,516,580: the reference handles "endtoend" exactly like "ctc" (trainer_bci.yaml passes it): same head, same loss"""
    from llm_bci_amd.ndt1 import NDT1
    fx = load("g_tiny")
    over = _det_over(str(fx["config_json"]))
    torch.manual_seed(1)
    m = NDT1(over, method_name="endtoend", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32")
    m.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w0:")})
    m.to(DEV).eval()
    with torch.no_grad():
        out = m(**_to_dev(batch_of(fx)))
    np.testing.assert_allclose(out.preds.cpu().numpy(), fx["eval_preds"], atol=1e-3)
    np.testing.assert_allclose(out.loss.item(), float(fx["eval_loss"]), rtol=1e-4)
    with pytest.raises(Exception, match="not implemented"):
        NDT1(over, method_name="mlm", vocab_size=11, blank_id=0, zero_infinity=True)


LONG = {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}}}
LONG_CTX = {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}, "context": {"forward": 5, "backward": 40}}}
ROPE = {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2, "use_rope": True}}}   # make_golden.py --rope: head 128


@pytest.mark.parametrize("name,over", [
    ("g_c1", {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}}}),
    ("g_c2", {}),
    ("g_long", LONG), ("g_long_ctx", LONG_CTX),       # 1200 bins -> 293 tokens, ragged (make_golden.py --long)
    ("g_c1_rope", ROPE), ("g_long_rope", ROPE),       # RoPE at 2 layers x 1024, head 128: T' = 18 / 10 and T' = 293 / 218 (make_golden.py --rope)
])
def test_c1_c2_golden_fp32(name, over):
    fx = load(name)
    m = _model(_det_over(json.dumps(over)), 41).to(DEV)
    batch = _to_dev(batch_of(fx))
    m.eval()
    with torch.no_grad():
        out = m(**batch)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.preds.cpu().numpy(), fx["eval_preds"], atol=1e-3)
    np.testing.assert_allclose(out.loss.item(), float(fx["eval_loss"]), rtol=1e-4)
    assert np.array_equal(m.last_argmax.cpu().numpy(), fx["argmax"])           # bit-exact alignment indices
    loss, _, g = _grads(m, batch, train=True)
    np.testing.assert_allclose(loss.sum(), float(fx["train_loss"]), rtol=1e-4)
    for k, gv in g.items():
        ref = fx["gval:" + k]
        got = gv.reshape(-1)[fx["gidx:" + k]]
        np.testing.assert_allclose(got, ref, atol=1e-3 * max(1.0, float(np.abs(ref).max())), err_msg=k)
        s, a = fx["gsum:" + k]
        assert abs(np.abs(gv.astype(np.float64)).sum() - a) <= 5e-3 * a + 1e-5, k


@pytest.mark.parametrize("streams", STREAMS)
def test_c2_golden_bf16(streams):
    """bf16 operands, f32 accumulate: log-probs within 0.08 abs of the reference's fp32 run and
    >= 97 % of the greedy path identical (frames whose fp32 top-2 margin is below bf16 noise may flip). Both stream dtypes are held
    to the same bounds (bf16 streams: measured 0.014 against 0.006 at B = 16, tools/ab_residual.py)."""
    fx = load("g_c2")
    m = _model(_det_over("{}"), 41, dtype="bf16", streams=streams).to(DEV)
    batch = _to_dev(batch_of(fx))
    m.eval()
    with torch.no_grad():
        out = m(**batch)
    torch.cuda.synchronize()
    preds = out.preds.cpu().numpy()
    measured(f"ndt1.c2_golden.{streams}_streams.logprob", np.abs(preds - fx["eval_preds"]).max())
    measured(f"ndt1.c2_golden.{streams}_streams.loss_rel", abs(out.loss.item() - float(fx["eval_loss"])) / float(fx["eval_loss"]))
    am = m.last_argmax.cpu().numpy()
    safe = fx["margin"] > 0.1
    assert np.array_equal(am[safe], fx["argmax"][safe])
    assert (am == fx["argmax"]).mean() > 0.97
    _, _, g = _grads(m, batch, train=True)
    for k, gv in g.items():
        s, a = fx["gsum:" + k]
        if a > 1e-3:
            measured(f"ndt1.c2_golden.{streams}_streams.grad_abs_sum_rel", abs(np.abs(gv.astype(np.float64)).sum() - a) / a)


@pytest.mark.parametrize("streams", STREAMS)
@pytest.mark.parametrize("name,over", [("g_long", LONG), ("g_long_ctx", LONG_CTX), ("g_c1_rope", ROPE), ("g_long_rope", ROPE)])
def test_long_sequence_golden_bf16_streaming_attention(name, over, streams):
    """293 tokens (> the 160 of the one-workgroup attention kernel): the bf16 path runs the MASKED streaming kernels of
    attn_flash.hip (key validity + context span + self). g_c1_rope / g_long_rope: rotary positions at head 128 through the fused (T' = 18)
    and the streaming (T' = 293) attention in bf16. Against the reference's fp32 run: log-probs within 0.08, argmax equal
    wherever the fp32 top-2 margin exceeds 0.1, gradient L1 within 5 %."""
    fx = load(name)
    m = _model(_det_over(json.dumps(over)), 41, dtype="bf16", streams=streams).to(DEV)
    batch = _to_dev(batch_of(fx))
    m.eval()
    with torch.no_grad():
        out = m(**batch)
    torch.cuda.synchronize()
    preds = out.preds.cpu().numpy()
    lens = fx["token_lens"]
    for b, L in enumerate(lens):                       # frames beyond a sample's tokens never reach the loss (their keys are padding)
        measured(f"ndt1.{name}.{streams}_streams.logprob", np.abs(preds[b, :L] - fx["eval_preds"][b, :L]).max())
    measured(f"ndt1.{name}.{streams}_streams.loss_rel", abs(out.loss.item() - float(fx["eval_loss"])) / float(fx["eval_loss"]))
    am = m.last_argmax.cpu().numpy()
    safe = fx["margin"] > 0.1
    for b, L in enumerate(lens):
        assert np.array_equal(am[b, :L][safe[b, :L]], fx["argmax"][b, :L][safe[b, :L]])
    _, _, g = _grads(m, batch, train=True)
    for k, gv in g.items():
        s_, a = fx["gsum:" + k]
        if a > 1e-3:
            measured(f"ndt1.{name}.{streams}_streams.grad_abs_sum_rel", abs(np.abs(gv.astype(np.float64)).sum() - a) / a)


def _oracle_cfg(m, **kw):
    c = m._ccfg
    return O.make_config(n_channels=c.n_channels, input_dim=c.input_dim, stack_size=c.stack_size, stack_stride=c.stack_stride,
                         hidden=c.hidden, n_layers=c.n_layers, n_heads=c.n_heads, inter=c.inter, vocab=c.vocab, max_F=c.max_F,
                         smooth_sd=int(c.smooth_sd), noise=bool(c.noise), white_noise_sd=c.white_noise_sd,
                         constant_offset_sd=c.constant_offset_sd, embed_dropout=c.embed_dropout, dropout=c.dropout,
                         use_rope=bool(c.use_rope), context_forward=c.context_forward, context_backward=c.context_backward,
                         factors_size=c.factors_size, factors_act={0: None, 1: "softsign", 2: "gelu", 3: "relu", 4: "tanh"}[c.factors_act],
                         factors_bias=bool(c.factors_bias), adapt_days=c.adapt_days, day_token_days=c.day_token_days,
                         block_token_blocks=c.block_token_blocks, **kw)


def _rand_batch(B, T, N, S, vocab, lens, tlens, seed=0):
    g = np.random.default_rng(seed)
    spikes = g.standard_normal((B, T, N)).astype(np.float32)
    mask = np.zeros((B, T), np.int64)
    ts = np.zeros((B, T), np.int64)
    for b, L in enumerate(lens):
        spikes[b, L:] = 0
        mask[b, :L] = 1
        ts[b, :L] = np.arange(L)
    return dict(spikes=spikes, spikes_mask=mask, spikes_timestamp=ts, spikes_lengths=np.array(lens, np.int64),
                targets=g.integers(1, vocab, (B, S)).astype(np.int64), targets_lengths=np.array(tlens, np.int64))


def _bf16_vs_oracle(over, vocab, batch, streams="fp32", tag="x"):
    """bf16 path vs the f32 oracle with identical dropout / noise draws: log-probs and per-tensor gradient L1 ratio within the measured bounds."""
    m = _model(over, vocab, dtype="bf16", streams=streams).to(DEV)
    p = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    loss, preds, g = _grads(m, _to_dev(batch), train=True, seed=1234)
    o, cache = O.forward(_oracle_cfg(m), p, batch, train=True, seed=1234)
    go = O.backward(cache)
    measured(f"ndt1.vs_oracle.{tag}.{streams}_streams.logprob", np.abs(preds - o["preds"]).max())
    for k in g:
        if k.endswith("attn.key.bias"):   # mathematically zero (softmax is shift-invariant along the keys): rounding noise only
            continue
        den = np.abs(go[k]).sum()
        if den > 1e-3:
            measured(f"ndt1.vs_oracle.{tag}.{streams}_streams.grad_l1_rel", np.abs(g[k] - go[k]).sum() / den)
        else:   # (a day no sample came from: both exactly zero)
            assert np.abs(g[k]).sum() < 1e-3, k


@pytest.mark.parametrize("which", ["tiny", "tiny_factors", "tiny_adapt", "tiny_tokens", "tiny_all", "c1", "c1_adapt_bf16", "c1_tokens_factors_bf16",
                                   "c1_bf16s", "c1_adapt_bf16s", "c1_tokens_factors_bf16s",     # ..._bf16s: bf16 residual / gradient streams
                                   "c1_rope", "c1_rope_bf16", "c1_rope_bf16s"])                  # RoPE at head 128 (ndt1.py:46-71,285-286), fused attention
def test_train_mode_matches_oracle_with_dropout_and_noise(which):
    """recipe dropout (0.2 / 0.4) and noise ON: HIP and oracle draw identical masks (same counter RNG)."""
    if which.startswith("tiny"):
        over = {"encoder": {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
                            "transformer": {"n_layers": 2, "hidden_size": 32, "n_heads": 2, "inter_size": 48}}}
        if which == "tiny_factors":   # NeuralFactorsProjection between out_norm and the decoder (ndt1.py:348-373)
            over["encoder"]["factors"] = {"active": True, "size": 24, "act": "relu", "bias": True}
        vocab, batch = 11, _rand_batch(3, 30, 16, 5, 11, [30, 22, 17], [5, 4, 2])
        if which == "tiny_adapt":     # day-specific embed layers, two samples sharing a day (ndt1.py:124-129,170-171)
            over["encoder"]["embedder"].update(adapt=True, n_days=3)
            batch["day_idx"] = np.array([2, 0, 2], np.int64)
        if which == "tiny_tokens":    # learned day + block tokens in front of the spike tokens (ndt1.py:192-201)
            over["encoder"]["embedder"].update(day_token=True, block_token=True, n_days=3, n_blocks=4)
            batch["day_idx"], batch["block_idx"] = np.array([2, 0, 2], np.int64), np.array([1, 3, 3], np.int64)
        if which == "tiny_all":       # every optional piece at once: day-specific layers + both tokens + factors + a context span
            over["encoder"]["embedder"].update(adapt=True, day_token=True, block_token=True, n_days=3, n_blocks=2)
            over["encoder"]["factors"] = {"active": True, "size": 16, "act": "tanh", "bias": False}
            over["encoder"]["context"] = {"forward": 4, "backward": 3}
            batch["day_idx"], batch["block_idx"] = np.array([1, 1, 0], np.int64), np.array([0, 1, 1], np.int64)
    else:
        over = {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}}}
        vocab, batch = 41, _rand_batch(4, 100, 64, 10, 41, [100, 100, 80, 64], [10, 8, 6, 3])
        if which.startswith("c1_rope"):
            over["encoder"]["transformer"]["use_rope"] = True
        if which.startswith("c1_adapt_bf16"):  # real widths: the batched direct-to-LDS GEMMs (K = 100 bins is not a multiple of 64)
            over["encoder"]["embedder"].update(adapt=True, n_days=5)
            batch["day_idx"] = np.array([4, 1, 4, 0], np.int64)
        if which.startswith("c1_tokens_factors_bf16"):   # real widths (fused attention with 18 + 2 tokens), block token + factors projection together
            over["encoder"]["embedder"].update(day_token=True, block_token=True, n_days=5, n_blocks=6)
            over["encoder"]["factors"] = {"active": True, "size": 512, "act": "relu", "bias": True}
            batch["day_idx"], batch["block_idx"] = np.array([4, 1, 4, 0], np.int64), np.array([5, 5, 2, 0], np.int64)
    if which.endswith("_bf16") or which.endswith("_bf16s"):
        _bf16_vs_oracle(over, vocab, batch, streams="bf16" if which.endswith("s") else "fp32", tag=which.rsplit("_bf16", 1)[0])
        return
    m = _model(over, vocab).to(DEV)
    p = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    loss, preds, g = _grads(m, _to_dev(batch), train=True, seed=1234)
    cfg = _oracle_cfg(m)
    o, cache = O.forward(cfg, p, batch, train=True, seed=1234)
    go = O.backward(cache)
    np.testing.assert_allclose(preds, o["preds"], atol=1e-3)
    np.testing.assert_allclose(loss, o["loss_per_sample"], rtol=2e-4, atol=1e-3)
    for k in g:
        tol = 1e-3 * max(1.0, float(np.abs(go[k]).max()))
        np.testing.assert_allclose(g[k], go[k], atol=tol, err_msg=k)
    # dropout actually happened: eval-mode output differs
    m.eval()
    l2, _ = m._run_forward(_to_dev(batch), want_grad=False)
    assert abs(float(l2.sum()) - float(loss.sum())) > 1e-3


def test_autograd_path_equals_native_and_ga_scaling():
    over = {"encoder": {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
                        "transformer": {"n_layers": 1, "hidden_size": 32, "n_heads": 2, "inter_size": 48}}}
    m = _model(over, 11).to(DEV)
    batch = _to_dev(_rand_batch(3, 30, 16, 5, 11, [30, 22, 17], [5, 4, 2]))
    _, _, g = _grads(m, batch, train=False)
    m.eval()
    out = m(**batch)
    (out.loss / 4).backward()                      # trainer.py:339 divides by gradient_accumulation_steps
    torch.cuda.synchronize()
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        np.testing.assert_allclose(p.grad.cpu().numpy() * 4, g[name], atol=1e-5 * max(1.0, np.abs(g[name]).max()), err_msg=name)


def test_infeasible_ctc_sample_contributes_nothing():
    """zero_infinity=True: a target longer than the token count gives loss 0 and zero gradients."""
    over = {"encoder": {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
                        "transformer": {"n_layers": 1, "hidden_size": 32, "n_heads": 2, "inter_size": 48}}}
    m = _model(over, 11).to(DEV)
    b = _rand_batch(2, 12, 16, 9, 11, [12, 12], [9, 2])   # T'=5 tokens < 9 labels for sample 0
    loss, preds, g = _grads(m, _to_dev(b), train=False)
    assert loss[0] == 0.0 and loss[1] > 0
    p = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    o, cache = O.forward(_oracle_cfg(m), p, b, train=False)
    go = O.backward(cache)
    for k in g:
        np.testing.assert_allclose(g[k], go[k], atol=1e-4 * max(1.0, float(np.abs(go[k]).max())), err_msg=k)


def test_per_metric_device_matches_reference_fixture_and_oracle():
    import ctypes as C
    from llm_bci_amd._lib import check, lib
    fx = load("metric_cases")
    po = np.concatenate([[0], np.cumsum(fx["paths_len"])]); to = np.concatenate([[0], np.cumsum(fx["tgts_len"])])
    n = len(fx["paths_len"]); Tp = int(fx["paths_len"].max()); S = int(fx["tgts_len"].max())
    paths = np.zeros((n, Tp), np.int32); tg = np.zeros((n, S), np.int64)
    for i in range(n):
        pth = fx["paths_flat"][po[i]:po[i + 1]]
        paths[i, :len(pth)] = pth                     # trailing frames = blank, as padded frames decode
        tg[i, :fx["tgts_len"][i]] = fx["tgts_flat"][to[i]:to[i + 1]]
    d = lambda a: torch.from_numpy(a).to(DEV)
    pa, tgd, tl = d(paths), d(tg), d(fx["tgts_len"].astype(np.int64))
    dec = torch.zeros(n, Tp, dtype=torch.int32, device=DEV); dl = torch.zeros(n, dtype=torch.int32, device=DEV)
    err = torch.zeros(n, 2, dtype=torch.int32, device=DEV); scr = torch.zeros(n * 2 * (S + 2), dtype=torch.int32, device=DEV)
    vp = lambda t: C.c_void_p(t.data_ptr())
    check(lib().nbci_per(vp(pa), vp(tgd), vp(tl), n, Tp, S, 0, vp(dec), vp(dl), vp(err), vp(scr),
                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nbci_per")
    torch.cuda.synchronize()
    assert np.array_equal(err.cpu().numpy(), fx["per"])
    do = np.concatenate([[0], np.cumsum(fx["dec_len"])])
    for i in range(n):
        assert list(dec[i, :dl[i]].cpu().numpy()) == list(fx["dec_flat"][do[i]:do[i + 1]])
    # random paths vs oracle
    g = np.random.default_rng(5)
    paths = g.integers(0, 6, (16, 40)).astype(np.int32); tg = g.integers(1, 6, (16, 12)).astype(np.int64)
    tl = g.integers(0, 13, (16,)).astype(np.int64)
    pa, tgd, tld = d(paths), d(tg), d(tl)
    dec = torch.zeros(16, 40, dtype=torch.int32, device=DEV); dl = torch.zeros(16, dtype=torch.int32, device=DEV)
    err = torch.zeros(16, 2, dtype=torch.int32, device=DEV); scr = torch.zeros(16 * 2 * 14, dtype=torch.int32, device=DEV)
    check(lib().nbci_per(vp(pa), vp(tgd), vp(tld), 16, 40, 12, 0, vp(dec), vp(dl), vp(err), vp(scr),
                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nbci_per")
    torch.cuda.synchronize()
    for i in range(16):
        dd = OM.format_ctc(paths[i], 0)
        tt = list(tg[i, :tl[i]])
        assert (OM.edit_distance(dd if dd else [""], tt if tt else [""]), max(1, len(tt))) == tuple(err[i].cpu().numpy())


def test_checkpoint_roundtrip_and_reference_key_layout(tmp_path):
    over = {"encoder": {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
                        "transformer": {"n_layers": 1, "hidden_size": 32, "n_heads": 2, "inter_size": 48}}}
    m = _model(over, 11).to(DEV)
    batch = _to_dev(_rand_batch(2, 20, 16, 3, 11, [20, 20], [3, 2]))
    m.eval()
    with torch.no_grad():
        a = m(**batch).preds.cpu()
    m.save_checkpoint(str(tmp_path))
    enc = torch.load(os.path.join(tmp_path, "encoder.bin"))
    assert "embedder.stack_projection.weight" in enc and "layers.0.attn.query.weight" in enc and "out_norm.bias" in enc
    assert set(torch.load(os.path.join(tmp_path, "decoder.bin")).keys()) == {"0.weight", "0.bias"}
    m2 = _model(over, 11, seed=99).to(DEV)
    m2.load_checkpoint(str(tmp_path))
    m2.eval()
    with torch.no_grad():
        b = m2(**batch).preds.cpu()
    assert torch.equal(a, b)


@pytest.mark.parametrize("T,lens,ctx,which", [
    (600, [600, 450], (-2, -2), "fused"), (100, [100, 70], (3, 2), "fused"), (664, [664, 664], (-2, -2), "fused"),
    # the masked streaming kernels (attn_flash.hip): what NDT1 runs beyond 160 tokens, and at shorter lengths when asked to
    (1200, [1200, 900], (-2, -2), "flash"), (1200, [1200, 731], (5, 40), "flash"), (600, [600, 450], (-2, -2), "flash"),
    (2048, [2048, 1500, 33], (-1, 64), "flash"), (100, [100, 70], (3, 2), "flash")])
def test_fused_attention_matches_batched_gemm_path(T, lens, ctx, which, monkeypatch):
    """bf16: attention.hip (fused, T' <= 160) resp. the masked streaming kernels of attn_flash.hip (any length) vs the
    batched-GEMM + softmax kernels on the same weights, inputs and dropout streams (train mode, recipe dropout, ragged lengths,
    context spans). All are bf16 pipelines that round at different points, so the bound is bf16-level: 0.03 on log-probs, 3 % of
    each gradient's max-abs."""
    over = {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2},
                        "context": {"forward": ctx[0], "backward": ctx[1]}}}
    B = len(lens)
    batch = _to_dev(_rand_batch(B, T, 64, 12, 41, lens, [12, 7, 3][:B]))
    outs = []
    for new in (True, False):
        monkeypatch.setenv("NBCI_FUSED_ATTN", "1" if (new and which == "fused") else "0")
        monkeypatch.setenv("NBCI_FLASH_ATTN", "1" if (new and which == "flash") else "0")
        m = _model(over, 41, dtype="bf16").to(DEV)
        outs.append(_grads(m, batch, train=True, seed=77))
    (l1, p1, g1), (l0, p0, g0) = outs
    assert np.abs(p1 - p0).max() < 0.03, np.abs(p1 - p0).max()
    np.testing.assert_allclose(l1, l0, rtol=5e-3)
    for k in g1:
        if k.endswith("attn.key.bias"):
            continue  # exactly zero in theory (softmax shift invariance): both paths return rounding noise
        scale = max(1e-6, float(np.abs(g0[k]).max()))
        assert np.abs(g1[k] - g0[k]).max() <= 0.03 * scale + 1e-6, (k, np.abs(g1[k] - g0[k]).max(), scale)


@pytest.mark.parametrize("streams", STREAMS)
def test_segmentwise_backward_equals_single_call(streams):
    """The DP path runs the backward one segment per call (head, layers L..1, embedder) so each bucket can be all-reduced
    while the next segment computes; a single process runs it as ONE call. Both must give the same gradients."""
    over = {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}}}
    batch = _to_dev(_rand_batch(4, 100, 64, 10, 41, [100, 100, 80, 64], [10, 8, 6, 3]))
    m = _model(over, 41, dtype="bf16", streams=streams).to(DEV)
    m.train()
    m._run_forward(batch, want_grad=True, seed=5)
    g1 = torch.zeros_like(m._flat)
    m._run_backward(g1)
    m._run_forward(batch, want_grad=True, seed=5)
    g2 = torch.zeros_like(m._flat)
    for seg in range(len(m._segments) - 1, -1, -1):
        m._run_backward(g2, seg, seg)
    torch.cuda.synchronize()
    assert float(g1.abs().sum()) > 0
    # bit-equal except where f32 atomics order the sums (bias / LayerNorm replicas, position table): compare with a tight tolerance
    assert torch.allclose(g1, g2, rtol=1e-4, atol=1e-5 * float(g1.abs().max()))


@pytest.mark.parametrize("variant", ["plain", "adapt_tokens", "f32"])
def test_embedder_backward_in_two_parts(variant):
    """The DP trainer runs segment 0 as embed_part=1 (stack projection / position / token tables: all-reduced at once)
    then embed_part=2 (embed_spikes). Part 1 must leave [begin, split) untouched and already hold the final values of
    [split, end); part 2 must not write [split, end); together they equal the one-call backward."""
    over = {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 1}}}
    extra = {}
    if variant == "adapt_tokens":
        over["encoder"]["embedder"].update({"adapt": True, "n_days": 3, "day_token": True, "block_token": True, "n_blocks": 2})
        extra = {"day_idx": torch.tensor([0, 2, 2, 1]), "block_idx": torch.tensor([1, 0, 1, 1])}
    batch = _rand_batch(4, 100, 64, 10, 41, [100, 100, 80, 64], [10, 8, 6, 3])
    batch.update(extra)
    batch = _to_dev(batch)
    m = _model(over, 41, dtype="fp32" if variant == "f32" else "bf16").to(DEV)
    m.train()
    nseg = len(m._segments)
    b0, e0 = m._segments[0]
    split = m._embed_split
    assert b0 < split < e0
    m._run_forward(batch, want_grad=True, seed=9)
    g1 = torch.zeros_like(m._flat)
    m._run_backward(g1)
    m._run_forward(batch, want_grad=True, seed=9)
    g2 = torch.zeros_like(m._flat)
    for seg in range(nseg - 1, 0, -1):
        m._run_backward(g2, seg, seg)
    m._run_backward(g2, 0, 0, embed_part=1)
    torch.cuda.synchronize()
    after1 = g2.clone()
    assert float(after1[b0:split].abs().sum()) == 0.0
    m._run_backward(g2, 0, 0, embed_part=2)
    torch.cuda.synchronize()
    assert torch.equal(after1[split:e0], g2[split:e0])          # part 2 writes nothing on part 1's side
    assert float(g2[b0:split].abs().sum()) > 0
    assert torch.allclose(g1, g2, rtol=1e-4, atol=1e-5 * float(g1.abs().max()))


@pytest.mark.parametrize("streams", STREAMS)
def test_c2_bf16_per_parity(capsys, streams):
    """PER parity of the bf16 path at C2 (the dtype bench.py times) against the reference's fp32 run recorded in g_c2.npz.
    (a) the device decode + edit distance on the bf16 greedy path is bit-exact against the oracle's format_ctc / word_error_count
        on the same path; (b) token counts equal the fixture's exactly and the error count differs from the fixture's by no more
        than the number of frames whose argmax flipped (each flipped frame changes the collapsed sequence by at most one edit);
    (c) every flipped frame has an fp32 top-2 margin below 0.1 nats. The measured flip rate is printed."""
    from llm_bci_amd.trainer import NativeTrainer
    fx = load("g_c2")
    m = _model(_det_over("{}"), 41, dtype="bf16", streams=streams).to(DEV)
    batch = _to_dev(batch_of(fx))
    m.eval()
    tr = NativeTrainer(m, total_steps=4)
    with torch.no_grad():
        m._run_forward(batch, want_grad=False)
        err = tr._per(batch).cpu().numpy()
    torch.cuda.synchronize()
    am = m.last_argmax.cpu().numpy()
    flips = am != fx["argmax"]
    tg, tl = fx["in_targets"], fx["in_targets_lengths"].reshape(-1)
    errs = toks = 0
    for b in range(am.shape[0]):
        dec = OM.format_ctc(am[b], 0)
        tt = list(tg[b, :tl[b]])
        e = OM.edit_distance(dec if dec else [""], tt if tt else [""])
        assert (e, max(1, len(tt))) == tuple(err[b]), b          # (a) device metric == oracle metric on the bf16 path
        errs += e; toks += max(1, len(tt))
    assert toks == int(fx["per_tokens"])
    assert abs(errs - int(fx["per_errors"])) <= int(flips.sum())
    assert fx["margin"][flips].max(initial=0.0) < 0.1
    with capsys.disabled():
        print(f"\n[bf16 PER parity @C2, {streams} streams] argmax flips {int(flips.sum())}/{flips.size} = {100.0 * flips.mean():.2f} % "
              f"(largest fp32 margin among them {fx['margin'][flips].max(initial=0.0):.4f}); "
              f"PER bf16 {errs}/{toks} vs reference fp32 {int(fx['per_errors'])}/{int(fx['per_tokens'])}")


def test_c2_at_the_benched_batch_64_against_the_cpu_restatement():
    """g_c2 is two samples; bench.py times C2 at B = 64. The same B = 64 ragged batch through (a) the fp32 HIP path, (b) the bf16
    path bench.py runs, against oracle/torch_step.py (the PyTorch-CPU restatement pinned to g_c1 / g_c2) in eval mode: log-probs
    <= 1e-3 (fp32) / 0.08 (bf16), CTC losses, greedy paths equal wherever the reference's top-2 margin exceeds 0.1 nats, and the
    device PER of the bf16 path within the flipped-frame count of the restatement's."""
    from oracle import torch_step as TS
    from llm_bci_amd.trainer import NativeTrainer
    B, T, N, S = 64, 600, 256, 60
    g = np.random.default_rng(12)
    lens = g.integers(300, T + 1, B); lens[0] = T
    tlens = np.maximum(1, (S * lens // T)).astype(np.int64)
    batch = _rand_batch(B, T, N, S, 41, [int(x) for x in lens], [int(x) for x in tlens], seed=12)
    m32 = _model(_det_over("{}"), 41, dtype="fp32").to(DEV)
    p = {k: v.detach().cpu().clone() for k, v in m32.state_dict().items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cb = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}
    with torch.no_grad():
        ref_loss, ref_lp, _tl = TS.forward(p, cb, TS.default_hparams(), train=False)
    ref_lp = ref_lp.numpy()
    top2 = np.sort(ref_lp, -1)[..., -2:]
    margin = top2[..., 1] - top2[..., 0]
    ref_am = ref_lp.argmax(-1)
    dev = _to_dev(batch)
    tok_len = 1 + (lens - 32) // 4
    valid = np.arange(ref_lp.shape[1])[None, :] < tok_len[:, None]          # frames of real tokens (padded frames are not compared)
    res = {}
    for dt, m in (("fp32", m32), ("bf16", _model(_det_over("{}"), 41, dtype="bf16").to(DEV)),
                  ("bf16s", _model(_det_over("{}"), 41, dtype="bf16", streams="bf16").to(DEV))):   # bf16s: bf16 residual / gradient streams
        m.eval()
        tr = NativeTrainer(m, total_steps=4)
        with torch.no_grad():
            loss, preds = m._run_forward(dev, want_grad=False)
            err = tr._per(dev).cpu().numpy()
        torch.cuda.synchronize()
        lp = preds.cpu().numpy()
        d = np.abs(lp - ref_lp)[valid].max()
        am = m.last_argmax.cpu().numpy()
        flips = (am != ref_am) & valid
        if dt == "fp32":
            assert d <= 1e-3, (dt, d)
        else:
            measured(f"ndt1.c2_b64.{dt}.logprob", d)
        np.testing.assert_allclose(float(loss.sum()), float(ref_loss), rtol=2e-4 if dt == "fp32" else 1e-2)
        assert margin[flips].max(initial=0.0) < (2e-3 if dt == "fp32" else 0.1), (dt, margin[flips].max(initial=0.0))
        res[dt] = (err, int(flips.sum()))
    e32 = res["fp32"][0]
    for dt in ("bf16", "bf16s"):
        e16 = res[dt][0]
        assert e32[:, 1].sum() == e16[:, 1].sum() == int(np.maximum(1, tlens).sum())
        assert abs(int(e16[:, 0].sum()) - int(e32[:, 0].sum())) <= res[dt][1] + res["fp32"][1]
        print(f"[C2 @ B=64] {dt} argmax flips {res[dt][1]} / {int(valid.sum())} frames; PER errors {int(e16[:, 0].sum())} vs fp32 {int(e32[:, 0].sum())} / {int(e32[:, 1].sum())}")


def test_maximum_length_max_F_tokens_against_the_cpu_restatement():
    """The longest sequence the reference's position table allows (embedder.max_F = 1024 tokens, configs/ndt1.yaml:39 -> 32 + 4 * 1023 =
    4124 bins), default widths, ragged: eval log-probs of the fp32 HIP path <= 1e-3 and of the bf16 path (masked streaming attention,
    chunked CTC) <= 0.08 against oracle/torch_step.py; greedy paths equal where the top-2 margin > 0.1; the CTC sum-loss; and one bin
    more is refused (ndt1.py:181-189 would index the position table out of range)."""
    from oracle import torch_step as TS
    B, T, N, S = 2, 4124, 256, 120
    batch = _rand_batch(B, T, N, S, 41, [T, 3301], [S, 77], seed=21)
    m32 = _model(_det_over("{}"), 41, dtype="fp32").to(DEV)
    p = {k: v.detach().cpu().clone() for k, v in m32.state_dict().items()}
    cb = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        ref_loss, ref_lp, tok = TS.forward(p, cb, TS.default_hparams(), train=False)
    ref_lp = ref_lp.numpy()
    assert ref_lp.shape[1] == 1024 and list(tok.numpy()) == [1024, 1 + (3301 - 32) // 4]
    valid = np.arange(1024)[None, :] < tok.numpy()[:, None]
    top2 = np.sort(ref_lp, -1)[..., -2:]
    margin = top2[..., 1] - top2[..., 0]
    dev = _to_dev(batch)
    for dt, m in (("fp32", m32), ("bf16", _model(_det_over("{}"), 41, dtype="bf16").to(DEV))):
        m.eval()
        with torch.no_grad():
            loss, preds = m._run_forward(dev, want_grad=False)
        torch.cuda.synchronize()
        lp = preds.cpu().numpy()
        if dt == "fp32":
            assert np.abs(lp - ref_lp)[valid].max() <= 1e-3, dt
        else:
            measured("ndt1.max_F.bf16.logprob", np.abs(lp - ref_lp)[valid].max())
        np.testing.assert_allclose(float(loss.sum()), float(ref_loss), rtol=2e-4 if dt == "fp32" else 1e-2)
        flips = (m.last_argmax.cpu().numpy() != ref_lp.argmax(-1)) & valid
        assert margin[flips].max(initial=0.0) < (2e-3 if dt == "fp32" else 0.1), dt
    too_long = _to_dev(_rand_batch(1, T + 4, N, 4, 41, [T + 4], [4], seed=1))
    with pytest.raises(Exception):
        m32._run_forward(too_long, want_grad=False)


def _tiny_over():
    return {"encoder": {"smooth_and_noise": {"noise": False},
                        "embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}, "dropout": 0.0},
                        "transformer": {"n_layers": 2, "hidden_size": 64, "n_heads": 2, "inter_size": 64, "dropout": 0.0}}}


def test_bf16_registry_swap_route_trains_every_weight():
    """The documented drop-in route: model(**batch).loss.backward() + torch.optim.AdamW(model.parameters()).
    The optimizer steps the f32 views in place; the bf16 shadow the GEMMs read must follow (it did not in round 1: weight
    matrices stayed frozen). Three steps in bf16 track the same three steps in fp32, and the eval forward after them (no_grad
    route) sees the updated weights too."""
    batch = _to_dev(_rand_batch(4, 40, 16, 5, 11, [40, 40, 36, 30], [5, 4, 3, 5]))
    runs = {}
    for dt in ("fp32", "bf16"):
        m = _model(_tiny_over(), 11, dtype=dt).to(DEV)
        w0 = dict(m.named_parameters())["encoder.layers.0.attn.query.weight"].detach().clone()
        opt = torch.optim.AdamW(m.parameters(), lr=3e-3, weight_decay=0.0)
        losses = []
        m.train()
        for _ in range(3):
            out = m(**batch)
            out.loss.backward()
            opt.step(); opt.zero_grad()
            losses.append(out.loss.item())
        m.eval()
        with torch.no_grad():
            ev = m(**batch)
        torch.cuda.synchronize()
        w1 = dict(m.named_parameters())["encoder.layers.0.attn.query.weight"].detach()
        assert (w1 - w0).abs().max() > 1e-4                      # the f32 weight moved ...
        if dt == "bf16":                                         # ... and the shadow the kernels read equals its rounding
            assert torch.equal(m._flat_lp, m._flat.to(torch.bfloat16))
        runs[dt] = (losses, ev.loss.item(), ev.preds.cpu().numpy())
    lf, lb = runs["fp32"][0], runs["bf16"][0]
    assert lf[2] < lf[0] and lb[2] < lb[0]                       # training moves the loss
    for a, b in zip(lf, lb):
        assert abs(a - b) / abs(a) < 0.02, (lf, lb)              # bf16 tracks fp32 step by step (a frozen shadow does not)
    assert abs(runs["fp32"][1] - runs["bf16"][1]) / abs(runs["fp32"][1]) < 0.02
    assert np.abs(runs["fp32"][2] - runs["bf16"][2]).max() < 0.08


def test_bridge_backward_refuses_another_forwards_activations():
    """Two grad-enabled forwards before a backward: the first loss's backward would read the second forward's workspace."""
    m = _model(_tiny_over(), 11).to(DEV)
    b1 = _to_dev(_rand_batch(2, 40, 16, 4, 11, [40, 40], [4, 4], seed=1))
    b2 = _to_dev(_rand_batch(2, 40, 16, 4, 11, [40, 40], [4, 4], seed=2))
    m.train()
    l1 = m(**b1).loss
    l2 = m(**b2).loss
    with pytest.raises(RuntimeError, match="saved activations are gone"):
        l1.backward()
    l2.backward()                                                # the most recent forward's backward is fine
