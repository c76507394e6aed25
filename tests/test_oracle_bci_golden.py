"""Pins oracle/bci.py (projector, stacking, splice, shifted CE, their backward) and oracle.ndt1.backward(d_enc_out=...) to the
fixtures the reference's own BCI.prepare_embeds / BCI.forward produced (tests/golden/make_golden.py --bci). CPU only.
The HF LLM between the splice and the loss is third-party: where a gradient has to pass through it the test rebuilds it with
transformers from the weights stored in the fixture (it is importable on both boxes) — it is never restated."""
import json

import numpy as np
import pytest

from oracle import bci as OB
from oracle import ndt1 as O
from test_oracle_golden import cfg_from_json, load


def _encode(fx, prefix="w:ndt1."):
    cfg = cfg_from_json(json.dumps(json.loads(str(fx["config_json"]))["ndt1"]), 11)
    p = {k[len(prefix):]: fx[k] for k in fx.files if k.startswith(prefix)}
    batch = dict(spikes=fx["spikes"], spikes_mask=fx["spikes_mask"], spikes_timestamp=fx["spikes_timestamp"],
                 spikes_lengths=fx["spikes_lengths"], targets=None, targets_lengths=None)
    out, cache = O.forward(cfg, p, batch, train=False)
    return out, cache


def _proj_params(fx):
    return {k[len("w:projector."):]: fx[k] for k in fx.files if k.startswith("w:projector.")}


def test_prepare_embeds_and_its_gradients_match_reference():
    fx = load("g_bci")
    pj = json.loads(str(fx["config_json"]))["projector"]
    out, cache = _encode(fx)
    x, valid = OB.stack_tokens(out["enc_out"], out["token_mask"], pj["stacking"])
    pp = _proj_params(fx)
    y, pc = OB.projector_fwd(x, pp, pj["act"])
    text = fx["embed_table"][fx["input_ids"]]
    emb, mask, tg = OB.splice_fwd(text, y, fx["attention_mask"], valid, fx["targets"], fx["input_split"])
    assert np.array_equal(mask, fx["out_mask"]) and np.array_equal(tg, fx["out_targets"])
    np.testing.assert_allclose(emb, fx["out_embeds"], atol=5e-5)
    # backward of sum(embeds * R)
    _dt, d_sp = OB.splice_bwd(fx["R"], fx["input_split"], text.shape[1], y.shape[1])
    g, dx = OB.projector_bwd(d_sp, pp, pc)
    for k, v in g.items():
        ref = fx["g:projector." + k]
        np.testing.assert_allclose(v, ref, atol=2e-4 * max(1.0, np.abs(ref).max()), err_msg=k)
    B, Tp, H = out["enc_out"].shape
    d_enc = dx.reshape(B, -1, H)[:, :Tp]            # the zero-padded stacking rows carry no encoder gradient
    ge = O.backward(cache, d_enc_out=d_enc)
    n = 0
    for k in fx.files:
        if k.startswith("g:ndt1."):
            ref = fx[k]
            np.testing.assert_allclose(ge[k[len("g:ndt1."):]], ref, atol=2e-4 * max(1.0, np.abs(ref).max()), err_msg=k)
            n += 1
    assert n >= 4


def test_forward_loss_and_all_gradients_match_reference_bci_forward():
    import torch
    from transformers import AutoModelForCausalLM, LlamaConfig
    fx = load("g_bci_fwd")
    pj = json.loads(str(fx["config_json"]))["projector"]
    out, cache = _encode(fx)
    x, valid = OB.stack_tokens(out["enc_out"], out["token_mask"], pj["stacking"])
    pp = _proj_params(fx)
    y, pc = OB.projector_fwd(x, pp, pj["act"])
    text = fx["w:llm.model.embed_tokens.weight"][fx["input_ids"]]
    emb, mask, tg = OB.splice_fwd(text, y, fx["attention_mask"], valid, fx["targets"], fx["input_split"])
    assert np.array_equal(tg, fx["out_targets"])
    # shifted CE on the reference's own logits
    loss, n, _dl = OB.shifted_ce_sum(fx["f32_logits"], tg)
    assert n == int(fx["n_examples"])
    np.testing.assert_allclose(loss, float(fx["f32_loss"]), rtol=1e-6)
    # through the third-party LLM (rebuilt from the stored weights) with the ORACLE's embeddings
    llm = AutoModelForCausalLM.from_config(LlamaConfig(**json.loads(str(fx["llm_config_json"]))))
    llm.load_state_dict({k[len("w:llm."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w:llm.")})
    llm.eval()
    e = torch.from_numpy(emb).requires_grad_(True)
    logits = llm(inputs_embeds=e, attention_mask=torch.from_numpy(mask), return_dict=True).logits
    np.testing.assert_allclose(logits.detach().numpy(), fx["f32_logits"], atol=2e-4)
    loss2, n2, dl = OB.shifted_ce_sum(logits.detach().numpy(), tg)
    np.testing.assert_allclose(loss2, float(fx["f32_loss"]), rtol=2e-5)
    logits.backward(torch.from_numpy(dl.astype(np.float32)))
    _dt, d_sp = OB.splice_bwd(e.grad.numpy(), fx["input_split"], text.shape[1], y.shape[1])
    g, dx = OB.projector_bwd(d_sp, pp, pc)
    for k, v in g.items():
        ref = fx["g32:projector." + k]
        np.testing.assert_allclose(v, ref, atol=2e-4 * max(1.0, np.abs(ref).max()), err_msg=k)
    B, Tp, H = out["enc_out"].shape
    ge = O.backward(cache, d_enc_out=dx.reshape(B, -1, H)[:, :Tp])
    for k in fx.files:
        if k.startswith("g32:ndt1."):
            ref = fx[k]
            np.testing.assert_allclose(ge[k[len("g32:ndt1."):]], ref, atol=2e-4 * max(1.0, np.abs(ref).max()), err_msg=k)
    # the reference's own precision (LLM in fp16): same loss within fp16 resolution
    assert abs(float(fx["f16_loss"]) - float(fx["f32_loss"])) / float(fx["f32_loss"]) < 2e-3


@pytest.mark.parametrize("act,bias,two", [("tanh", False, True), ("relu", True, True), ("relu", True, False)])
def test_projector_backward_is_the_gradient_of_its_forward(act, bias, two):
    """finite differences in float64 (covers the 41-input tanh / no-bias coupler of configs/phoneme_coupler.yaml)."""
    g = np.random.default_rng(0)
    I, M, N = 41, 12, 9
    p = {}
    if two:
        p["0.weight"], p["2.weight"] = g.standard_normal((M, I)) * 0.3, g.standard_normal((N, M)) * 0.3
        if bias:
            p["0.bias"], p["2.bias"] = g.standard_normal(M) * 0.1, g.standard_normal(N) * 0.1
    else:
        p["weight"] = g.standard_normal((N, I)) * 0.3
        if bias:
            p["bias"] = g.standard_normal(N) * 0.1
    x = g.standard_normal((2, 5, I))
    R = g.standard_normal((2, 5, N))
    y, c = OB.projector_fwd(x, p, act)
    gr, dx = OB.projector_bwd(R, p, c)
    f = lambda: float((OB.projector_fwd(x, p, act)[0] * R).sum())
    for k in p:
        idx = tuple(g.integers(0, s) for s in p[k].shape)
        old = p[k][idx]; h = 1e-6
        p[k][idx] = old + h; fp = f(); p[k][idx] = old - h; fm = f(); p[k][idx] = old
        assert abs((fp - fm) / (2 * h) - gr[k][idx]) < 1e-5, k
    old = x[1, 2, 3]; h = 1e-6
    x[1, 2, 3] = old + h; fp = f(); x[1, 2, 3] = old - h; fm = f(); x[1, 2, 3] = old
    assert abs((fp - fm) / (2 * h) - dx.reshape(x.shape)[1, 2, 3]) < 1e-5
