"""Pins oracle/itransformer.py (iTransformer SSL: masker rules, encoder, mlm head, losses, backward, AdamW) to fixtures
generated from the REFERENCE (tests/golden/make_golden.py --itr). CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import itransformer as OI
from oracle import optim as OO

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def itr_cfg(fx):
    over = json.loads(str(fx["config_json"]))
    kw = json.loads(str(fx["kwargs_json"]))
    enc = over.get("encoder", {})
    c = dict(embed_dropout=0.0, dropout=0.0, log_input=kw["log_input"], loss=kw["loss"])
    for src, dst in (("hidden_size", "hidden"), ("n_heads", "n_heads"), ("n_layers", "n_layers"), ("max_n_channels", "max_n_channels")):
        if src in enc:
            c[dst] = enc[src]
    emb = enc.get("embedder", {})
    if "max_n_bins" in emb:
        c["max_n_bins"] = emb["max_n_bins"]
    if enc.get("embed_region", False):       # (every fixture config states embed_region; configs/itransformer.yaml's default is true)
        c["n_regions"] = len(enc["regions"])
    c["embed_depth"] = bool(enc.get("embed_depth", False))
    if emb.get("mode", "mlp") == "transformer":   # configs/itransformer.yaml:21-27 defaults: 128 x 4 heads x 4 layers
        c.update(embedder_mode="transformer", emb_hidden=emb.get("hidden_size", 128), emb_heads=emb.get("n_heads", 4),
                 emb_layers=emb.get("n_layers", 4), emb_act=emb.get("activation", "relu"))
    return OI.make_config(**c)


def itr_regions(fx):
    over = json.loads(str(fx["config_json"]))
    return over.get("encoder", {}).get("regions") if over.get("encoder", {}).get("embed_region", False) else None


def itr_batch(fx):
    b = {k[3:]: fx[k] for k in fx.files if k.startswith("in_")}
    regs = itr_regions(fx)
    if regs is not None:       # the reference's region_to_indx (itransformer.py:136,196): position in config.regions
        lut = {r: i for i, r in enumerate(regs)}
        b["region_idx"] = np.vectorize(lut.__getitem__)(b["neuron_regions"]).astype(np.int64)
    return b


def masked_of(spikes, mask):
    out = spikes.copy()
    out[mask.astype(bool)] = 0      # zero_ratio = 1.0 in the fixtures' masker config
    return out


TINY = ["g_itr_tiny", "g_itr_tiny_ss", "g_itr_tiny_rate", "g_itr_tiny_mse",
        "g_itr_tiny_region", "g_itr_tiny_region_depth", "g_itr_tiny_depth",      # embed_region (the shipped default) / embed_depth
        "g_itr_tiny_uni", "g_itr_tiny_uni_all"]                                  # embedder.mode: transformer (UnivariateTransformer)


@pytest.mark.parametrize("name", TINY)
def test_itr_tiny_forward_backward_adamw(name):
    fx = load(name)
    cfg = itr_cfg(fx)
    p = {k[3:]: fx[k] for k in fx.files if k.startswith("w0:")}
    assert set(p) == set(OI.init_params(cfg)), set(p) ^ set(OI.init_params(cfg))     # the oracle knows every tensor of the reference's state dict
    batch = itr_batch(fx)
    m = fx["eval_raw_mask"]
    out, _ = OI.forward(cfg, p, batch, masked_of(batch["spikes"], m), m, train=False)
    if cfg["embedder_mode"] == "transformer":
        for l in range(cfg["emb_layers"]):
            np.testing.assert_allclose(out["emb_layer_out"][l], fx[f"emb_layer{l}_out"], atol=5e-5)
        np.testing.assert_allclose(out["emb_out"], fx["emb_out"], atol=5e-5)
        np.testing.assert_allclose(out["emb_out"][:, 0, :].reshape(fx["emb_cls"].shape), fx["emb_cls"], atol=5e-5)
    np.testing.assert_allclose(out["embed"], fx["embed"], atol=2e-5)
    np.testing.assert_allclose(out["tokens"], fx["tokens"], atol=2e-5)
    for l in range(cfg["n_layers"]):
        np.testing.assert_allclose(out["layer_out"][l], fx[f"layer{l}_out"], atol=5e-5)
    np.testing.assert_allclose(out["encoder_out"], fx["encoder_out"], atol=5e-5)
    np.testing.assert_allclose(out["preds"], fx["eval_preds"], atol=1e-4)
    assert np.array_equal(out["mask"], fx["eval_mask"])
    assert int(out["n_examples"]) == int(fx["eval_n_examples"])
    np.testing.assert_allclose(out["loss"], fx["eval_loss"], rtol=2e-5)
    # two train steps (each with the mask the reference drew that step) vs torch AdamW + OneCycleLR
    m_, v_ = {k: np.zeros_like(x) for k, x in p.items()}, {k: np.zeros_like(x) for k, x in p.items()}
    p = {k: x.copy() for k, x in p.items()}
    for s in range(2):
        mk = fx[f"raw_mask_step{s}"]
        out, cache = OI.forward(cfg, p, batch, masked_of(batch["spikes"], mk), mk, train=True)
        np.testing.assert_allclose(out["loss"], fx[f"loss_step{s}"], rtol=3e-5)
        assert int(out["n_examples"]) == int(fx[f"n_examples_step{s}"])
        g = OI.backward(cache)
        if s == 0:
            for k in p:
                ref = fx["grad:" + k]
                np.testing.assert_allclose(g[k], ref, atol=2e-5 + 2e-4 * np.abs(ref).max(), err_msg=k)
        lr, b1 = OO.onecycle(s, 100, 1e-4, 0.15, 25.0)
        assert abs(lr - float(fx[f"lr_step{s}"])) < 1e-12 and abs(b1 - float(fx[f"beta1_step{s}"])) < 1e-9
        for k in p:
            OO.adamw_step(p[k], g[k], m_[k], v_[k], s + 1, lr, b1, 0.999, 1e-8, 0.01)
    for k in p:
        # Adam normalises: a gradient at the rounding floor may flip the sign of a 1e-4-sized update
        d = np.abs(p[k] - fx["w2:" + k])
        assert np.mean(d > 2e-5) < 0.02 and d.max() < 5e-4, (k, d.max())


# 64 channels x 4 samples; the recipe's 668 channels x 2 samples, without and WITH region embeddings (the shipped default); the shipped
# UnivariateTransformer embedder widths (128 x 4 heads x 4 layers over 101 tokens) under the shipped encoder
@pytest.mark.parametrize("name", ["g_itr_c3", "g_itr_c3w", "g_itr_c3w_region", "g_itr_uni_c3"])
def test_itr_c3_real_shapes(name):
    fx = load(name)
    cfg = itr_cfg(fx)
    import torch
    from llm_bci_amd.itransformer import reference_order_init   # host-side init (pure torch CPU): bit-equal to the reference
    emb = dict(h=cfg["emb_hidden"], nh=cfg["emb_heads"], L=cfg["emb_layers"]) if cfg["embedder_mode"] == "transformer" else None
    p = reference_order_init(cfg_shapes=dict(T=cfg["max_n_bins"], H=cfg["hidden"], L=cfg["n_layers"], nh=cfg["n_heads"],
                                             C=cfg["max_n_channels"], use_cls=True, mlp_decoder=True), seed=1, n_regions=cfg["n_regions"],
                             embed_depth=cfg["embed_depth"], embedder=emb)
    p = {k: v.numpy() for k, v in p.items()}
    assert set(p) == {k[6:] for k in fx.files if k.startswith("w0idx:")}
    for k in p:
        idx = fx["w0idx:" + k]
        assert np.array_equal(p[k].reshape(-1)[idx], fx["w0val:" + k]), k
    batch = itr_batch(fx)
    m = fx["eval_raw_mask"]
    out, _ = OI.forward(cfg, p, batch, masked_of(batch["spikes"], m), m, train=False)
    np.testing.assert_allclose(out["preds"][:, ::3, ::5], fx["eval_preds"], atol=1e-3)
    np.testing.assert_allclose(out["encoder_out"][..., ::37], fx["encoder_out"], atol=1e-3)
    np.testing.assert_allclose(out["loss"], fx["eval_loss"], rtol=1e-4)
    assert int(out["n_examples"]) == int(fx["eval_n_examples"])
    mk = fx["raw_mask_step0"]
    out, cache = OI.forward(cfg, p, batch, masked_of(batch["spikes"], mk), mk, train=True)
    np.testing.assert_allclose(out["loss"], fx["loss_step0"], rtol=1e-4)
    g = OI.backward(cache)
    for k in p:
        gs = fx["gsum:" + k]
        ref = fx["gval:" + k]
        got = g[k].reshape(-1)[fx["gidx:" + k]]
        # (668 channels: the LayerNorm / bias gradients are f32 sums over 1 336 token rows with cancellation; numpy's summation order differs from torch's)
        tol = 1e-3 if name == "g_itr_c3" else 3e-3
        np.testing.assert_allclose(got, ref, atol=1e-5 + tol * max(np.abs(ref).max(), gs[1] / g[k].size), err_msg=k)


def test_masker_rules_match_reference_structure():
    """The reference's draws are torch's; what is pinned: axis structure of each mode, replacement rules, expand_timesteps."""
    fx = load("masker_cases")
    sp = fx["spikes"]
    for w in (1, 2, 3, 4):
        assert np.array_equal(OI.expand_timesteps(fx["expand_in"].astype(bool), w), fx[f"expand_{w}"]), w
    base = dict(active=True, force_active=True, ratio=0.3, zero_ratio=1.0, random_ratio=1.0, expand_prob=0.0, max_timespan=1)
    for mode in ("temporal", "neuron", "random", "co-smooth"):
        mc = dict(base, mode=mode, channels=[1, 5, 7])
        out, mask = OI.masker(mc, sp, True, seed=11, site=OI.SITE_MASKER)
        ref_mask, ref_out = fx[mode + "_mask"], fx[mode + "_out"]
        for mm, oo in ((mask, out), (ref_mask, ref_out)):      # same structural invariants on both
            if mode == "temporal":
                assert (mm == mm[:, :, :1]).all()
            if mode == "neuron":
                assert (mm == mm[:, :1, :]).all()
            if mode == "co-smooth":
                assert (mm == mm[:1, :1, :]).all() and set(np.nonzero(mm[0, 0])[0]) == {1, 5, 7}
            assert (oo[mm.astype(bool)] == 0).all() and np.array_equal(oo[~mm.astype(bool)], sp[~mm.astype(bool)])
        if mode != "co-smooth":
            assert 0.1 < mask.mean() < 0.55
    mc = dict(base, mode="random", zero_ratio=0.5, random_ratio=0.5)
    out, mask = OI.masker(mc, sp, True, seed=3, site=OI.SITE_MASKER)
    for mm, oo in ((mask, out), (fx["random_mixed_mask"], fx["random_mixed_out"])):
        mb = mm.astype(bool)
        assert np.array_equal(oo[~mb], sp[~mb])
        changed = oo[mb] != sp[mb]
        assert 0.3 < (oo[mb] == 0).mean() < 0.85 and changed.any()
        assert oo.max() <= sp.max() + 1e-6
    # inactive / eval pass-through (masker.py:50-51)
    out, mask = OI.masker(dict(base, mode="neuron", force_active=False), sp, False, 1, OI.SITE_MASKER)
    assert np.array_equal(out, sp) and mask.sum() == 0
    out, mask = OI.masker(dict(base, mode="temporal", expand_prob=1.0, max_timespan=3), sp, True, 5, OI.SITE_MASKER)
    assert (mask == mask[:, :, :1]).all()


def test_masker_copy_modes_match_reference_structure():
    """forward-pred / inter-region / intra-region of the reference's "models/masker copy.py" (:81-104,117,133), recorded by
    make_golden.py --masker-copy. Deterministic cases (ratio 1, every listed region sampled) must match bit for bit; in the
    others the draws differ (torch / Python RNG vs the counter RNG), so the same structural invariants are checked on both."""
    fx = load("masker_copy_cases")
    sp, regions = fx["spikes"], fx["regions"]
    B, T, N = sp.shape

    def run(name):
        mc = json.loads(str(fx[name + "_cfg"]))
        mc["active"] = True
        return mc, OI.masker(mc, sp, True, seed=21, site=OI.SITE_MASKER, neuron_regions=regions)

    for name in ("forward_pred", "inter_all", "intra_all"):          # deterministic: exact
        mc, (out, mask) = run(name)
        assert np.array_equal(mask, fx[name + "_mask"]), name
        assert np.array_equal(out, fx[name + "_out"]), name
    assert set(np.nonzero(fx["forward_pred_mask"][0, :, 0])[0]) == {2, 5, 6}
    assert np.array_equal(fx["inter_all_mask"][:, 0, :].astype(bool), np.isin(regions, ["CA1", "PO"]))
    assert (fx["intra_all_out"] == 0).all() and np.array_equal(fx["intra_all_mask"][:, 0, :].astype(bool), regions == "DG")

    mc, (out, mask) = run("inter_half")                              # one of three regions, half of its neurons
    for mm, oo in ((mask, out), (fx["inter_half_mask"], fx["inter_half_out"])):
        mb = mm.astype(bool)
        assert (mm == mm[:, :1, :]).all()                            # constant along time
        hit = set(regions[mb[:, 0, :]])
        assert len(hit) <= 1 and hit <= {"CA1", "PO", "DG"}          # a single sampled region
        assert (oo[mb] == 0).all() and np.array_equal(oo[~mb], sp[~mb])
    assert 0 < mask.sum() < np.isin(regions, ["CA1", "PO", "DG"]).sum() * T

    mc, (out, mask) = run("intra_some")                              # targets inside DG + VIS; everything outside is masked too
    tgt = np.isin(regions, ["DG", "VIS"])
    for mm, oo in ((mask, out), (fx["intra_some_mask"], fx["intra_some_out"])):
        mb = mm.astype(bool)
        assert (mm == mm[:, :1, :]).all() and not mb[:, 0, :][~tgt].any()         # returned targets lie inside the target regions
        assert (oo[:, :, :][np.broadcast_to(~tgt[:, None, :], oo.shape)] == 0).all()   # every neuron outside them is corrupted
        inside = np.broadcast_to(tgt[:, None, :], oo.shape)
        assert np.array_equal((oo == 0) & inside, mb)                              # inside: corrupted exactly where returned
    assert 0 < mask.sum() < tgt.sum() * T
    # the region sample: distinct members of the list, a function of (seed, site)
    s1 = OI.region_sample(5, OI.SITE_MASKER, ["a", "b", "c", "d"], 3)
    assert len(set(s1)) == 3 and set(s1) <= {"a", "b", "c", "d"} and s1 == OI.region_sample(5, OI.SITE_MASKER, ["a", "b", "c", "d"], 3)
    assert {tuple(OI.region_sample(s, OI.SITE_MASKER, ["a", "b", "c", "d"], 1)) for s in range(40)} == {("a",), ("b",), ("c",), ("d",)}
