"""Pins the numpy oracle (oracle/) to fixtures produced by the REFERENCE itself
(tests/golden/make_golden.py, run in the build container). CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import collate as OC
from oracle import ctc as OCTC
from oracle import metrics as OM
from oracle import ndt1 as O
from oracle import optim as OO

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def cfg_from_json(js, vocab):
    over = json.loads(js)
    enc = over.get("encoder", {})
    emb, tr, ctx = enc.get("embedder", {}), enc.get("transformer", {}), enc.get("context", {})
    kw = dict(vocab=vocab, noise=False, embed_dropout=0.0, dropout=0.0)
    for src, dst in (("n_channels", "n_channels"), ("input_dim", "input_dim"), ("max_F", "max_F")):
        if src in emb:
            kw[dst] = emb[src]
    if "stack" in emb:
        kw["stack_size"], kw["stack_stride"] = emb["stack"]["size"], emb["stack"]["stride"]
    for src, dst in (("n_layers", "n_layers"), ("hidden_size", "hidden"), ("n_heads", "n_heads"),
                     ("inter_size", "inter"), ("use_rope", "use_rope")):
        if src in tr:
            kw[dst] = tr[src]
    if "forward" in ctx:
        kw["context_forward"], kw["context_backward"] = ctx["forward"], ctx["backward"]
    if emb.get("adapt", False):
        kw["adapt_days"] = emb["n_days"]
    if emb.get("day_token", False):
        kw["day_token_days"] = emb["n_days"]
    if emb.get("block_token", False):
        kw["block_token_blocks"] = emb["n_blocks"]
    fac = enc.get("factors", {})
    if fac.get("active", False):
        kw["factors_size"], kw["factors_act"], kw["factors_bias"] = fac["size"], fac.get("act", "relu"), fac.get("bias", True)
    return O.make_config(**kw)


def batch_of(fx):
    return {k[3:]: fx[k] for k in fx.files if k.startswith("in_")}


@pytest.mark.parametrize("name", ["g_tiny", "g_tiny_ctx", "g_tiny_rope", "g_tiny_factors", "g_tiny_factors_fix", "g_tiny_adapt", "g_tiny_tokens", "g_tiny_daytoken"])
def test_tiny_forward_backward_adamw(name):
    fx = load(name)
    cfg = cfg_from_json(str(fx["config_json"]), 11)
    p = {k[3:]: fx[k] for k in fx.files if k.startswith("w0:")}
    batch = batch_of(fx)
    out, cache = O.forward(cfg, p, batch, train=False)
    assert np.array_equal(out["token_lens"], fx["token_lens"])
    assert np.array_equal(out["token_mask"], fx["token_mask"])
    np.testing.assert_allclose(out["xs"], fx["smooth"], atol=2e-6)
    np.testing.assert_allclose(out["x_embed"], fx["embed_x"], atol=2e-5)
    for l in range(cfg["n_layers"]):
        lo = cache["layers"][l + 1]["x_in"] if l + 1 < cfg["n_layers"] else cache["x_last"]
        np.testing.assert_allclose(lo, fx[f"layer{l}_out"], atol=5e-5)
    np.testing.assert_allclose(out["x_final"], fx["out_norm"], atol=5e-5)
    if "factors" in fx.files:
        np.testing.assert_allclose(out["enc_out"], fx["factors"], atol=5e-5)
    np.testing.assert_allclose(out["preds"], fx["eval_preds"], atol=1e-4)   # north_star: logits <= 1e-3
    np.testing.assert_allclose(out["loss"], fx["eval_loss"], rtol=1e-5)
    assert int(out["n_examples"]) == int(fx["n_examples"])
    # bit-exact integer outputs: argmax path, decode, PER counts
    assert np.array_equal(np.argmax(out["preds"], -1), fx["argmax"])
    tl = batch["targets_lengths"].reshape(-1)
    tg = [batch["targets"][b][:tl[b]] for b in range(len(tl))]
    e, n, dec = OM.per_counts(out["preds"], tg)
    assert (e, n) == (int(fx["per_errors"]), int(fx["per_tokens"]))
    assert np.array_equal(np.array([x for d in dec for x in d], np.int64), fx["decoded_flat"])
    # gradients of the sum-loss
    g = O.backward(cache)
    for k in p:
        ref = fx["grad:" + k]
        tol = 2e-4 * max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(g[k], ref, atol=tol, err_msg=k)
    # two AdamW + OneCycle steps (trainer.py:229,240-246,340-343)
    w = {k: v.copy() for k, v in p.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v2 = {k: np.zeros_like(v) for k, v in p.items()}
    for s in range(2):
        lr, b1 = OO.onecycle(s, 100, 1e-3, 0.0, 25)
        assert abs(lr - float(fx[f"lr_step{s}"])) < 1e-12 and abs(b1 - float(fx[f"beta1_step{s}"])) < 1e-12
        o, c = O.forward(cfg, w, batch, train=True)
        np.testing.assert_allclose(o["loss"], fx[f"loss_step{s}"], rtol=2e-5)
        gs = O.backward(c)
        for k in w:
            OO.adamw_step(w[k], gs[k], m[k], v2[k], s + 1, lr, b1, 0.999, 1e-8, 5e-5)
    # Adam normalises by |g|: where a gradient is at rounding-noise level the update is
    # +-lr with an arbitrary sign, so allow a handful of such elements (bounded by 2 steps * lr).
    for k in w:
        if k.endswith("attn.key.bias"):
            continue  # softmax is invariant to a key bias: true grad == 0, Adam amplifies pure rounding noise
        diff = np.abs(w[k] - fx["w2:" + k])
        assert (diff > 2e-5).mean() <= 0.005 and diff.max() <= 2.1e-3, (k, diff.max())


def test_ctc_cases():
    fx = load("ctc_cases")
    for nm in ("basic", "repeat_infeasible", "too_short", "empty_target", "long"):
        loss, grad = OCTC.ctc_loss_and_grad(fx[nm + "_lp"], fx[nm + "_targets"], fx[nm + "_il"], fx[nm + "_tl"])
        np.testing.assert_allclose(loss, fx[nm + "_loss"], rtol=1e-5, atol=1e-5, err_msg=nm)
        np.testing.assert_allclose(grad, fx[nm + "_grad"], atol=2e-5, err_msg=nm)


def test_metric_cases():
    fx = load("metric_cases")
    po = np.concatenate([[0], np.cumsum(fx["paths_len"])])
    to = np.concatenate([[0], np.cumsum(fx["tgts_len"])])
    do = np.concatenate([[0], np.cumsum(fx["dec_len"])])
    for i in range(len(fx["paths_len"])):
        path = fx["paths_flat"][po[i]:po[i + 1]]
        dec = OM.format_ctc(path, 0)
        assert dec == list(fx["dec_flat"][do[i]:do[i + 1]])
        tgt = list(fx["tgts_flat"][to[i]:to[i + 1]])
        d_tok = dec if dec else [""]
        assert (OM.edit_distance(d_tok, tgt), len(tgt)) == tuple(fx["per"][i])
    # the reference quirk: A blank A -> A
    assert OM.format_ctc([3, 0, 3], 0) == [3]


def test_context_masks():
    fx = load("misc_cases")
    for key in fx.files:
        _, f, b = key.split("_")
        assert np.array_equal(O.context_mask(int(f), int(b), 24), fx[key]), key


def test_collate_matches_fixture_inputs():
    """oracle.collate reproduces the reference's pad_collate_fn output stored in the fixture."""
    fx = load("g_tiny")
    g = np.random.default_rng(0)
    sp, tg = [], []
    for L, S in zip([30, 22, 17], [5, 4, 2]):
        sp.append(g.standard_normal((L, 16)).astype(np.float32))
        tg.append(g.integers(1, 11, (S,)).astype(np.int64))
    names = ["spikes", "spikes_mask", "spikes_timestamp", "spikes_lengths", "targets", "targets_lengths"]
    batch, unused = OC.pad_collate(OC.make_rows(sp, tg), names)
    for k in names:
        assert np.array_equal(np.asarray(batch[k]), fx["in_" + k]), k
    assert "targets_mask" in unused and "spikes_spacestamp" in unused


# ---------------------------------------------------------------------------------------------
# C1 / C2 shapes: weights are regenerated from seed 1 by the host module's reference-order init
# (llm_bci_amd.ndt1.NDT1, CPU construction only) and checked against the fixture's checksums.
def _regen(over, vocab=41):
    import torch
    from llm_bci_amd.ndt1 import NDT1
    torch.manual_seed(1)
    m = NDT1(over, method_name="ctc", vocab_size=vocab, blank_id=0, zero_infinity=True, compute_dtype="fp32")
    return m, {k: v.detach().numpy() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("name,over", [
    ("g_c1", {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}}}),
    ("g_c2", {}),
    ("g_long", {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}}}),
    ("g_long_ctx", {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}, "context": {"forward": 5, "backward": 40}}}),
    ("g_c1_rope", {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2, "use_rope": True}}}),     # RoPE at head 128
    ("g_long_rope", {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2, "use_rope": True}}}),
])
def test_c1_c2_init_and_oracle(name, over):
    fx = load(name)
    m, p = _regen(over)
    ref_keys = sorted(k[6:] for k in fx.files if k.startswith("w0sum:"))
    assert sorted(p.keys()) == ref_keys, "state-dict keys differ from the reference's"
    for k in p:
        flat = p[k].reshape(-1).astype(np.float64)
        s, a = fx["w0sum:" + k]
        assert abs(flat.sum() - s) <= 1e-6 * max(1.0, a) and abs(np.abs(flat).sum() - a) <= 1e-6 * max(1.0, a), k
        np.testing.assert_array_equal(p[k].reshape(-1)[fx["w0idx:" + k]], fx["w0val:" + k])
    cfg = cfg_from_json(str(fx["config_json"]), 41)
    batch = batch_of(fx)
    out, cache = O.forward(cfg, p, batch, train=False)
    assert np.array_equal(out["token_lens"], fx["token_lens"])
    np.testing.assert_allclose(out["preds"], fx["eval_preds"], atol=1e-3)          # north_star tolerance
    np.testing.assert_allclose(out["loss"], fx["eval_loss"], rtol=2e-5)
    assert np.array_equal(np.argmax(out["preds"], -1), fx["argmax"])               # bit-exact alignment indices
    np.testing.assert_allclose(out["x_embed"][:, :, ::37], fx["embed_x"], atol=1e-3)
    np.testing.assert_allclose(out["x_final"][:, :, ::37], fx["out_norm"], atol=1e-3)
    tl = batch["targets_lengths"].reshape(-1)
    e, n, _ = OM.per_counts(out["preds"], [batch["targets"][b][:tl[b]] for b in range(len(tl))])
    assert (e, n) == (int(fx["per_errors"]), int(fx["per_tokens"]))
    g = O.backward(cache)
    for k in p:
        s, a = fx["gsum:" + k]
        got = g[k].reshape(-1).astype(np.float64)
        assert abs(np.abs(got).sum() - a) <= 2e-3 * a + 1e-6, (k, np.abs(got).sum(), a)
        ref = fx["gval:" + k]
        np.testing.assert_allclose(g[k].reshape(-1)[fx["gidx:" + k]], ref, atol=2e-4 * max(1.0, float(np.abs(ref).max())), err_msg=k)
