"""GPU: the iTransformer SSL (mlm) HIP path through the C-ABI against (a) fixtures generated from the reference
(tests/golden/g_itr_*.npz; fp32 path, deterministic mask replayed), (b) the numpy oracle with identical dropout / masker
draws (train mode), (c) itself in bf16; and the device Masker against the oracle bit for bit."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

from oracle import itransformer as OI
from oracle import optim as OO
from test_oracle_itr_golden import itr_batch, itr_cfg, load, masked_of

from conftest import measured

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dev(batch):
    """device batch with the forward's keyword names; neuron_regions stays what datasets.py hands over (an array of names), the oracle's
    region_idx is dropped"""
    out = {}
    for k, v in batch.items():
        if k == "region_idx":
            continue
        out[k] = v if k == "neuron_regions" else torch.from_numpy(np.ascontiguousarray(v)).to(DEV)
    return out


def _model(fx_or_over, dtype="fp32", **kw):
    from llm_bci_amd.itransformer import iTransformer
    if isinstance(fx_or_over, dict):
        over, kwargs = fx_or_over, dict(log_input=True, loss="poisson_nll")
    else:
        over, kwargs = json.loads(str(fx_or_over["config_json"])), json.loads(str(fx_or_over["kwargs_json"]))
    kwargs.update(kw)
    torch.manual_seed(1)
    return iTransformer(over, method_name="mlm", compute_dtype=dtype, **kwargs)


def _grads_of(m, batch, seed=None):
    m.train()
    loss, preds = m._run_forward(batch, want_grad=True, seed=seed)
    g = torch.zeros_like(m._flat)
    m._run_backward(g)
    torch.cuda.synchronize()
    return loss, preds, {n: g[o:o + k].view(s).cpu().numpy() for (n, o, k, s, _sg) in m._layout}


# ----------------------------------------------------------------------------------------------- masker
@pytest.mark.parametrize("mode,extra", [("temporal", {}), ("neuron", {}), ("random", {}), ("co-smooth", {"channels": [1, 5, 7]}),
                                        ("temporal", {"expand_prob": 1.0, "max_timespan": 3}),
                                        ("random", {"zero_ratio": 0.5, "random_ratio": 0.5}),
                                        ("region", {"regions": ["CA1", "PO"]})])
def test_masker_matches_oracle_bit_exact(mode, extra):
    from llm_bci_amd.itransformer import SITE_MASKER, iTransformer
    g = np.random.default_rng(3)
    B, T, N = 5, 12, 37
    spikes = g.poisson(2.0, (B, T, N)).astype(np.float32)
    mc = dict(active=True, force_active=True, mode=mode, ratio=0.3, zero_ratio=1.0, random_ratio=1.0, expand_prob=0.0, max_timespan=1)
    mc.update(extra)
    over = {"encoder": {"embedder": {"max_n_bins": T, "dropout": 0.0}, "hidden_size": 32, "n_heads": 2, "n_layers": 1, "dropout": 0.0,
                        "max_n_channels": 64, "embed_region": False}, "masker": {"main": mc}}
    m = _model(over).to(DEV)
    m.train()
    regions = np.array([["CA1", "DG", "PO", "VIS"][i % 4] for i in range(B * N)]).reshape(B, N)
    probs = np.isin(regions, ["CA1", "PO"]).astype(np.float32)
    for seed in (7, 123456789):
        sp = torch.from_numpy(spikes).to(DEV)
        masked, mask, _keep = m._apply_maskers(sp, regions, seed)
        torch.cuda.synchronize()
        ref_out, ref_mask = OI.masker(mc, spikes, True, seed, SITE_MASKER, probs=probs)
        assert np.array_equal(mask.cpu().numpy(), ref_mask)
        np.testing.assert_allclose(masked.cpu().numpy(), ref_out, rtol=1e-6, atol=0)
        assert np.array_equal(sp.cpu().numpy(), spikes)      # the caller's tensor is left alone
        assert 0 < ref_mask.mean() < 1
    m.eval()
    mc2 = dict(mc, force_active=False)
    m.masker_cfg = [("main", mc2)]
    masked, mask, _ = m._apply_maskers(torch.from_numpy(spikes).to(DEV), regions, 1)
    assert mask.sum().item() == 0 and np.array_equal(masked.cpu().numpy(), spikes)   # masker.py:50-51


@pytest.mark.parametrize("case", ["forward_pred", "inter_all", "inter_half", "intra_all", "intra_some"])
def test_masker_copy_modes_bit_exact_vs_oracle_and_reference_structure(case):
    """the three extra modes of the reference's "models/masker copy.py" on the device: bit-exact against the oracle (same counter
    RNG, same region sample), and in the deterministic cases equal to what the reference's own Masker returned."""
    from llm_bci_amd.itransformer import SITE_MASKER, region_sample
    fx = load("masker_copy_cases")
    sp, regions = fx["spikes"], fx["regions"]
    B, T, N = sp.shape
    mc = json.loads(str(fx[case + "_cfg"]))
    mc["active"] = True
    over = {"encoder": {"embedder": {"max_n_bins": T, "dropout": 0.0}, "hidden_size": 32, "n_heads": 2, "n_layers": 1, "dropout": 0.0,
                        "max_n_channels": 16, "embed_region": False}, "masker": {"main": mc}}
    m = _model(over).to(DEV)
    m.train()
    for seed in (21, 987654321):
        d = torch.from_numpy(sp).to(DEV)
        masked, mask, _keep = m._apply_maskers(d, regions, seed)
        torch.cuda.synchronize()
        ref_out, ref_mask = OI.masker(mc, sp, True, seed, SITE_MASKER, neuron_regions=regions)
        assert np.array_equal(mask.cpu().numpy(), ref_mask)
        np.testing.assert_allclose(masked.cpu().numpy(), ref_out, rtol=1e-6, atol=0)
        assert np.array_equal(d.cpu().numpy(), sp)                    # the caller's tensor is left alone
        if case in ("forward_pred", "inter_all", "intra_all"):        # deterministic: the reference's own output
            assert np.array_equal(mask.cpu().numpy(), fx[case + "_mask"]) and np.array_equal(masked.cpu().numpy(), fx[case + "_out"])
    assert region_sample(5, SITE_MASKER, ["a", "b", "c", "d"], 3) == OI.region_sample(5, SITE_MASKER, ["a", "b", "c", "d"], 3)


def test_masker_two_maskers_accumulate():
    from llm_bci_amd.itransformer import SITE_MASKER
    g = np.random.default_rng(4)
    spikes = g.poisson(2.0, (3, 12, 10)).astype(np.float32)
    a = dict(active=True, force_active=True, mode="neuron", ratio=0.3, zero_ratio=1.0, random_ratio=1.0, expand_prob=0.0, max_timespan=1)
    b = dict(a, mode="temporal", ratio=0.25)
    over = {"encoder": {"embedder": {"max_n_bins": 12, "dropout": 0.0}, "hidden_size": 32, "n_heads": 2, "n_layers": 1, "dropout": 0.0,
                        "max_n_channels": 16, "embed_region": False}, "masker": {"main": a, "second": b}}
    m = _model(over).to(DEV)
    m.train()
    masked, mask, _ = m._apply_maskers(torch.from_numpy(spikes).to(DEV), None, 99)
    o1, m1 = OI.masker(a, spikes, True, 99, SITE_MASKER)
    o2, m2 = OI.masker(b, o1, True, 99, SITE_MASKER + 8)
    assert np.array_equal(mask.cpu().numpy(), m1 | m2)
    np.testing.assert_allclose(masked.cpu().numpy(), o2)


# ----------------------------------------------------------------------------------------------- golden parity (fp32 path)
# (the last five: embed_region - the shipped default of configs/itransformer.yaml -, embed_depth, and embedder.mode: transformer = the
#  UnivariateTransformer embedder alone and with every embedding on; reference itransformer.py:40-93,119-150,195-202)
@pytest.mark.parametrize("name", ["g_itr_tiny", "g_itr_tiny_ss", "g_itr_tiny_rate", "g_itr_tiny_mse", "g_itr_tiny_region", "g_itr_tiny_region_depth",
                                  "g_itr_tiny_depth", "g_itr_tiny_uni", "g_itr_tiny_uni_all"])
def test_fp32_matches_reference_golden_tiny(name):
    from llm_bci_amd.trainer import NativeTrainer
    fx = load(name)
    m = _model(fx).to(DEV)
    for k, v in m.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), fx["w0:" + k]), k      # reference-order init is bit-equal
    batch = _dev(itr_batch(fx))
    m.eval()
    m.mask_override = torch.from_numpy(fx["eval_raw_mask"])
    with torch.no_grad():
        out = m(**batch)
    np.testing.assert_allclose(out.preds.cpu().numpy(), fx["eval_preds"], atol=1e-3)   # north_star tolerance (fp32 path)
    assert np.array_equal(out.mask.cpu().numpy(), fx["eval_mask"])
    assert int(out.n_examples) == int(fx["eval_n_examples"])
    np.testing.assert_allclose(float(out.loss), float(fx["eval_loss"]), rtol=1e-4)
    assert np.array_equal(out.targets.cpu().numpy(), fx["in_spikes"])
    # gradients of the first train step (stochastic ops off, the reference's mask replayed)
    m.mask_override = torch.from_numpy(fx["raw_mask_step0"])
    loss, _, g = _grads_of(m, batch)
    np.testing.assert_allclose(float(loss.sum()), float(fx["loss_step0"]), rtol=1e-4)
    assert set(g) == {k[5:] for k in fx.files if k.startswith("grad:")}      # every tensor of the reference's state dict has its gradient
    for k in g:
        ref = fx["grad:" + k]
        np.testing.assert_allclose(g[k], ref, atol=2e-5 + 1e-3 * np.abs(ref).max(), err_msg=k)
    # two AdamW + OneCycle steps of the native trainer vs torch.optim on the reference model
    m2 = _model(fx).to(DEV)
    tr = NativeTrainer(m2, lr=1e-4, wd=0.01, eps=1e-8, scheduler="cosine", total_steps=100, warmup_pct=0.15, div_factor=25, compute_per=False)
    for s in range(2):
        m2.mask_override = torch.from_numpy(fx[f"raw_mask_step{s}"])
        loss, _ = tr.train_step(batch)
        np.testing.assert_allclose(float(loss.sum()), float(fx[f"loss_step{s}"]), rtol=2e-4)
        assert int(m2.last_n_examples) == int(fx[f"n_examples_step{s}"])
    torch.cuda.synchronize()
    for k, v in m2.state_dict().items():
        d = np.abs(v.cpu().numpy() - fx["w2:" + k])
        assert (d > 3e-5).mean() <= 0.05 and d.max() <= 5e-4, (k, d.max())
    st = tr.read_stats()
    assert st["n_examples"] == int(fx["n_examples_step0"]) + int(fx["n_examples_step1"])


# 64 channels x 4 samples; the recipe's 668 channels x 2 samples (streaming attention, 669 tokens) without and WITH region embeddings (the shipped
# default); the shipped UnivariateTransformer widths (128 x 4 heads x 4 layers over 101-token sequences, streaming attention at head 32)
@pytest.mark.parametrize("name", ["g_itr_c3", "g_itr_c3w", "g_itr_c3w_region", "g_itr_uni_c3"])
def test_fp32_matches_reference_golden_c3_and_bf16_close(name):
    fx = load(name)
    batch = _dev(itr_batch(fx))
    m = _model(fx).to(DEV)
    for k, v in m.state_dict().items():
        assert np.array_equal(v.cpu().numpy().reshape(-1)[fx["w0idx:" + k]], fx["w0val:" + k]), k
    m.eval()
    m.mask_override = torch.from_numpy(fx["eval_raw_mask"])
    with torch.no_grad():
        out = m(**batch)
    p32 = out.preds.cpu().numpy()
    np.testing.assert_allclose(p32[:, ::3, ::5], fx["eval_preds"], atol=1e-3)
    np.testing.assert_allclose(float(out.loss), float(fx["eval_loss"]), rtol=2e-4)
    assert int(out.n_examples) == int(fx["eval_n_examples"])
    m.mask_override = torch.from_numpy(fx["raw_mask_step0"])
    loss, _, g = _grads_of(m, batch)
    np.testing.assert_allclose(float(loss.sum()), float(fx["loss_step0"]), rtol=2e-4)
    for k in g:
        ref = fx["gval:" + k]
        got = g[k].reshape(-1)[fx["gidx:" + k]]
        np.testing.assert_allclose(got, ref, atol=2e-5 + 2e-3 * max(np.abs(ref).max(), fx["gsum:" + k][1] / g[k].size), err_msg=k)
    # bf16 operands (f32 accumulate) stay close to the fp32 path - with the LayerNorm inputs / gradient streams stored in f32 and in bf16
    for streams in ("fp32", "bf16"):
        mb = _model(fx, dtype="bf16", residual_dtype=streams).to(DEV)
        mb.eval()
        mb.mask_override = torch.from_numpy(fx["eval_raw_mask"])
        with torch.no_grad():
            ob = mb(**batch)
        measured(f"itr.{name}.{streams}_streams.pred", np.abs(ob.preds.cpu().numpy() - p32).max())
        measured(f"itr.{name}.{streams}_streams.loss_rel", abs(float(ob.loss) - float(out.loss)) / abs(float(out.loss)))
        mb.mask_override = torch.from_numpy(fx["raw_mask_step0"])
        _, _, gb = _grads_of(mb, batch)
        for k in g:
            num, den = np.abs(gb[k] - g[k]).sum(), np.abs(g[k]).sum() + 1e-6
            # (worst: the channel / region table's LayerNorm weight - a small gradient summed over few rows)
            measured(f"itr.{name}.{streams}_streams.grad_l1_rel", num / den)


# ----------------------------------------------------------------------------------------------- train mode vs oracle
@pytest.mark.parametrize("dtype,N,lens", [("fp32", 10, [12, 9, 7]), ("fp32", 70, [12, 12, 5, 3]), ("bf16", 70, [12, 12, 5, 3]),
                                          ("bf16/f32 streams", 70, [12, 12, 5, 3])])
def test_train_mode_dropout_and_maskers_match_oracle(dtype, N, lens):
    """recipe-style step: masker on (device RNG), dropout 0.2 / 0.4 on; the oracle mirrors every draw."""
    from llm_bci_amd.itransformer import SITE_MASKER
    T, B = 12, len(lens)
    mc = dict(active=True, force_active=True, mode="neuron", ratio=0.3, zero_ratio=0.8, random_ratio=0.5, expand_prob=0.0, max_timespan=1)
    over = {"encoder": {"embedder": {"max_n_bins": T, "dropout": 0.2}, "hidden_size": 32, "n_heads": 2, "n_layers": 2, "dropout": 0.4,
                        "max_n_channels": 96, "embed_region": False}, "masker": {"main": mc}}
    kw = {"residual_dtype": "fp32" if (dtype.endswith("f32 streams") or dtype == "fp32") else "bf16"}   # ("bf16": the opt-in bf16 streams)
    dtype = dtype.split("/")[0]
    m = _model(over, dtype=dtype, **kw).to(DEV)
    p = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    g = np.random.default_rng(5)
    spikes = g.poisson(0.7, (B, T, N)).astype(np.float32)
    smask = np.zeros((B, T), np.int64)
    for b, L in enumerate(lens):
        spikes[b, :T - L] = 0; smask[b, T - L:] = 1
    ss = np.stack([g.permutation(96)[:N] for _ in range(B)]).astype(np.int64)
    batch = dict(spikes=spikes, spikes_mask=smask, spikes_spacestamp=ss)
    cfg = OI.make_config(max_n_bins=T, hidden=32, n_heads=2, n_layers=2, max_n_channels=96, embed_dropout=0.2, dropout=0.4)
    _train_mode_check(m, mc, p, batch, cfg, dtype)


def _train_mode_check(m, mc, p, batch, cfg, dtype, tag="mlp"):
    from llm_bci_amd.itransformer import SITE_MASKER
    from llm_bci_amd._lib import NBCI_BF16
    tag = f"itr.train_oracle.{tag}.{'bf16' if m.residual_dtype == NBCI_BF16 else 'fp32'}_streams"
    spikes = batch["spikes"]
    seed = 4242
    loss, preds, gh = _grads_of(m, _dev(batch), seed=seed)
    masked, mask = OI.masker(mc, spikes, True, seed, SITE_MASKER)
    out, cache = OI.forward(cfg, p, batch, masked, mask, train=True, seed=seed)
    go = OI.backward(cache)
    assert int(m.last_n_examples) == int(out["n_examples"]) and int(out["n_examples"]) > 0
    assert np.array_equal(m.last_mask.cpu().numpy(), out["mask"])
    if dtype == "fp32":
        np.testing.assert_allclose(preds.cpu().numpy(), out["preds"], atol=1e-3)
    else:
        measured(tag + ".pred", np.abs(preds.cpu().numpy() - out["preds"]).max())
    np.testing.assert_allclose(float(loss.sum()), float(out["loss"]), rtol=1e-4 if dtype == "fp32" else 3e-2)
    for k in go:
        if dtype == "fp32":
            np.testing.assert_allclose(gh[k], go[k], atol=2e-5 + 1e-3 * np.abs(go[k]).max(), err_msg=k)
        else:
            measured(tag + ".grad_l1_rel", np.abs(gh[k] - go[k]).sum() / (np.abs(go[k]).sum() + 1e-6))
    assert set(gh) == set(go)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16/f32 streams"])
@pytest.mark.parametrize("mode", ["mlp", "transformer"])
def test_train_mode_every_embedding_on_matches_oracle(mode, dtype):
    """recipe-style step (device masker, dropout 0.2 / 0.4) with region + depth embeddings on, under both embedders; in `transformer` mode the
    UnivariateTransformer's layers draw their own dropout (embedder.dropout, sites 128 + 4 l + k) and read spikes_timestamp."""
    T, N, lens = 12, 70, [12, 12, 5, 3]      # (the shapes of the test above: at a few dozen token rows the bf16 L1 ratios are dominated by single roundings)
    B = len(lens)
    regs = ["CA1", "DG", "LP", "PO"]
    mc = dict(active=True, force_active=True, mode="neuron", ratio=0.3, zero_ratio=0.8, random_ratio=0.5, expand_prob=0.0, max_timespan=1)
    emb = {"max_n_bins": T, "dropout": 0.2}
    if mode == "transformer":
        emb.update(mode="transformer", hidden_size=32, n_heads=2, n_layers=2, activation="relu")
    over = {"encoder": {"embedder": emb, "hidden_size": 32, "n_heads": 2, "n_layers": 2, "dropout": 0.4, "max_n_channels": 96,
                        "embed_region": True, "regions": regs, "embed_depth": True}, "masker": {"main": mc}}
    kw = {"residual_dtype": "fp32" if (dtype.endswith("f32 streams") or dtype == "fp32") else "bf16"}
    dtype = dtype.split("/")[0]
    m = _model(over, dtype=dtype, **kw).to(DEV)
    p = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    g = np.random.default_rng(9)
    spikes = g.poisson(0.7, (B, T, N)).astype(np.float32)
    smask = np.zeros((B, T), np.int64); ts = np.zeros((B, T), np.int64)
    for b, L in enumerate(lens):
        spikes[b, :T - L] = 0; smask[b, T - L:] = 1; ts[b, T - L:] = np.arange(L)
    nr = np.array(regs)[g.integers(0, len(regs), (B, N))]
    batch = dict(spikes=spikes, spikes_mask=smask, spikes_timestamp=ts, neuron_regions=nr,
                 region_idx=np.vectorize({r: i for i, r in enumerate(regs)}.__getitem__)(nr).astype(np.int64),
                 neuron_depths=g.uniform(0, 3.84, (B, N)).astype(np.float32))
    extra = dict(embedder_mode="transformer", emb_hidden=32, emb_heads=2, emb_layers=2) if mode == "transformer" else {}
    cfg = OI.make_config(max_n_bins=T, hidden=32, n_heads=2, n_layers=2, max_n_channels=96, embed_dropout=0.2, dropout=0.4, n_regions=len(regs),
                         embed_depth=True, **extra)
    assert set(p) == set(OI.init_params(cfg))
    _train_mode_check(m, mc, p, batch, cfg, dtype, tag="all_embeddings_" + mode)


def test_autograd_bridge_and_checkpoint_roundtrip(tmp_path):
    fx = load("g_itr_tiny")
    m = _model(fx).to(DEV)
    batch = _dev(itr_batch(fx))
    m.train()
    m.mask_override = torch.from_numpy(fx["raw_mask_step0"])
    out = m(**batch)
    out.loss.backward()
    for k, p in m.named_parameters():
        ref = fx["grad:" + k]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2e-5 + 1e-3 * np.abs(ref).max(), err_msg=k)
    m.save_checkpoint(str(tmp_path))
    over = json.loads(str(fx["config_json"]))
    over["encoder"]["from_pt"] = str(tmp_path); over["decoder"] = {"from_pt": str(tmp_path)}
    torch.manual_seed(5)
    from llm_bci_amd.itransformer import iTransformer
    m2 = iTransformer(over, method_name="mlm", loss="poisson_nll", log_input=True, compute_dtype="fp32").to(DEV)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    ref_keys = {k[3:] for k in fx.files if k.startswith("w0:")}
    assert set(m.state_dict().keys()) == ref_keys


def test_unsupported_configs_fail_loudly():
    from llm_bci_amd.itransformer import iTransformer
    with pytest.raises(Exception, match="not implemented"):
        iTransformer({"encoder": {"embed_region": False}}, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, loss="x")
    with pytest.raises(Exception, match="not implemented"):
        iTransformer({"encoder": {"embed_region": False, "embedder": {"mode": "conv"}}}, method_name="mlm", loss="poisson_nll", log_input=True)
    with pytest.raises(Exception, match="needs encoder.regions"):      # the shipped yaml: embed_region true, regions null (main.py:39-42 fills it)
        iTransformer({}, method_name="mlm", loss="poisson_nll", log_input=True)
    m = _model({"encoder": {"embedder": {"max_n_bins": 12}, "hidden_size": 32, "n_heads": 2, "n_layers": 1, "max_n_channels": 16,
                            "embed_region": False}})
    with pytest.raises(Exception):   # CPU tensors: no fallback
        m(torch.zeros(1, 12, 4), torch.ones(1, 12, dtype=torch.int64), torch.zeros(1, 12, dtype=torch.int64))
