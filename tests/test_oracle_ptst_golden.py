"""Pins oracle/patchtst.py (PatchTST encoder restated from transformers + the reference's heads, BatchNorm batch/running
statistics, backward, AdamW) to fixtures generated from the REFERENCE (tests/golden/make_golden.py --ptst). CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import optim as OO
from oracle import patchtst as OP

G = os.path.join(os.path.dirname(__file__), "golden")
ENC_KEYS = ("num_input_channels", "context_length", "patch_length", "patch_stride", "num_hidden_layers", "d_model", "num_attention_heads",
            "ffn_dim", "attention_dropout", "ff_dropout", "do_mask_input", "random_mask_ratio", "positional_dropout", "path_dropout")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def ptst_cfg(fx):
    over, kw = json.loads(str(fx["config_json"])), json.loads(str(fx["kwargs_json"]))
    c = {k: v for k, v in over.get("encoder", {}).items() if k in ENC_KEYS}
    c.update({k: v for k, v in over.get("decoder", {}).items() if k in ("mlp_decoder", "head_dropout", "mlp_activation")})
    c["method"] = kw["method_name"]
    if c["method"] == "ctc":
        c["vocab"] = kw["vocab_size"]
    else:
        c["log_input"], c["loss"] = kw["log_input"], kw["loss"]
    return OP.make_config(**c)


def split_state(fx, prefix="w0:"):
    p, bufs = {}, {}
    for k in fx.files:
        if not k.startswith(prefix):
            continue
        n = k[len(prefix):]
        (bufs if n.endswith(("running_mean", "running_var", "num_batches_tracked")) else p)[n] = fx[k]
    return p, bufs


def ptst_batch(fx):
    return {k[3:]: fx[k] for k in fx.files if k.startswith("in_")}


def state_for(fx, full):
    """full fixtures carry the reference's weights; the big one is rebuilt by the host-side reference-order init (pure CPU
    torch / transformers constructors) and checked against the fixture's sampled values, which must match bit for bit."""
    if full:
        return split_state(fx)
    from llm_bci_amd.config import patchtst_config
    from llm_bci_amd.patchtst_init import reference_order_init
    kw = json.loads(str(fx["kwargs_json"]))
    cfg = patchtst_config(json.loads(str(fx["config_json"])))
    st = {k: v.float().numpy() for k, v in reference_order_init(cfg["encoder"], cfg["decoder"], kw["method_name"], kw.get("vocab_size"), seed=1).items()}
    for k, v in st.items():
        assert np.array_equal(v.reshape(-1)[fx["w0idx:" + k]], fx["w0val:" + k]), k
    p = {k: v for k, v in st.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    return p, {k: v for k, v in st.items() if k not in p}


def run_steps(fx, cfg, full=True, nsteps=2):
    p, bufs = state_for(fx, full)
    p = {k: v.copy() for k, v in p.items()}
    batch = ptst_batch(fx)
    mlm = cfg["method"] == "mlm"
    cut = (lambda a: a) if full else (lambda a: a[:, ::max(1, a.shape[1] // 16), ::7, ::13])   # as make_golden.py samples (B, C, P, D)
    cutp = (lambda a: a) if full else (lambda a: a[..., ::3, :])

    def check(tag, out, tol):
        np.testing.assert_allclose(cut(out["embed"]), fx[tag + "_embed"], atol=tol)
        for l in range(cfg["num_hidden_layers"]):
            np.testing.assert_allclose(cut(out["layer_out"][l]), fx[f"{tag}_layer{l}"], atol=5 * tol)
        np.testing.assert_allclose(cutp(out["preds"]), fx[tag + "_preds"], atol=1e-4 if full else 1e-3)
        np.testing.assert_allclose(out["loss"], fx[tag + "_loss"], rtol=5e-5 if full else 2e-4)
        assert int(out["n_examples"]) == int(fx[tag + "_n_examples"])
        if mlm:
            assert np.array_equal(out["mask"], fx[tag + "_mask"])

    out, _, _ = OP.forward(cfg, p, bufs, batch, mask=fx["eval0_raw_mask"] if mlm else None, train=False)
    check("eval0", out, 2e-6)
    if mlm and full:
        np.testing.assert_array_equal(out["patch_input"], fx["patch_input"])
    m_ = {k: np.zeros_like(x) for k, x in p.items()}
    v_ = {k: np.zeros_like(x) for k, x in p.items()}
    for s in range(nsteps):
        # sampled (C5-shape) fixture: 52 k token rows; numpy's f32 column sums over them carry ~1e-2 relative noise in the small gradients
        # (the f64 oracle agrees with the reference's f32 run to 1e-4), so the restatement is evaluated in f64 there
        out, cache, bufs = OP.forward(cfg, p, bufs, batch, mask=fx[f"step{s}_raw_mask"] if mlm else None, train=True,
                                      dtype=np.float32 if full else np.float64)
        check(f"step{s}", out, 2e-5 if (s == 0 or full) else 2e-4)   # after an Adam step rounding noise in tiny gradients is amplified
        g = OP.backward(cache)
        if s == 0:
            for k in g:
                if full:
                    ref = fx["grad:" + k]
                    np.testing.assert_allclose(g[k], ref, atol=2e-6 + 3e-4 * np.abs(ref).max(), err_msg=k)
                else:
                    ref = fx["gval:" + k]
                    got = g[k].reshape(-1)[fx["gidx:" + k]]
                    np.testing.assert_allclose(got, ref, atol=1e-6 + 2e-3 * max(np.abs(ref).max(), fx["gsum:" + k][1] / g[k].size), err_msg=k)
        lr, b1 = OO.onecycle(s, 100, 1e-3, 0.0, 25.0)
        for k in g:        # position_enc has requires_grad=False: no gradient, AdamW skips it
            OO.adamw_step(p[k], g[k].astype(np.float32), m_[k], v_[k], s + 1, lr, b1, 0.999, 1e-8, 5e-5)
    return p, bufs, batch


@pytest.mark.parametrize("name", ["g_ptst_tiny", "g_ptst_tiny_ov", "g_ptst_tiny_mlm", "g_ptst_tiny_mlm_rate"])
def test_ptst_tiny_forward_backward_adamw_bn_stats(name):
    fx = load(name)
    cfg = ptst_cfg(fx)
    p, bufs, batch = run_steps(fx, cfg)
    for k, v in bufs.items():     # BatchNorm running statistics after two train steps
        np.testing.assert_allclose(np.asarray(v, np.float32), fx["w2:" + k], rtol=1e-4, atol=1e-6, err_msg=k)
    for k in p:
        if k.endswith("k_proj.bias"):     # true gradient is 0 (softmax shift invariance): Adam amplifies rounding noise
            continue
        d = np.abs(p[k] - fx["w2:" + k])
        assert np.mean(d > 3e-5) < 0.03 and d.max() < 2.1e-3, (k, d.max())
    mlm = cfg["method"] == "mlm"
    out, _, _ = OP.forward(cfg, {k: fx["w2:" + k] for k in p}, bufs, batch, mask=fx["eval2_raw_mask"] if mlm else None, train=False)
    np.testing.assert_allclose(out["preds"], fx["eval2_preds"], atol=2e-4)      # eval mode reads the running statistics


def test_ptst_position_enc_and_mask_rule():
    fx = load("g_ptst_c5")
    pe = OP.position_enc(205, 256)
    np.testing.assert_allclose(pe.reshape(-1)[fx["w0idx:encoder.encoder.positional_encoder.position_enc"]],
                               fx["w0val:encoder.encoder.positional_encoder.position_enc"], atol=2e-6)
    cfg = OP.make_config(random_mask_ratio=0.4, channel_consistent_masking=False)
    m = OP.random_mask(cfg, 3, 5, 10, seed=9)
    assert m.shape == (3, 5, 10) and (m.sum(-1) == 10 - int(10 * 0.6)).all()     # exactly P - len_keep per (b,c) row
    ref = load("g_ptst_tiny_mlm")["eval0_raw_mask"]
    assert (ref.sum(-1) == 4 - int(4 * (1 - 0.4))).all()
    m2 = OP.random_mask(dict(cfg, channel_consistent_masking=True), 3, 5, 10, seed=9)
    assert (m2 == m2[:, :1]).all()


def test_ptst_c5_shapes():
    fx = load("g_ptst_c5")
    run_steps(fx, ptst_cfg(fx), full=False, nsteps=1)   # (one train step: the f64 restatement of 52 k token rows takes a minute per step)
