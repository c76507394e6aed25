"""CPU-only checks of the host side: C-ABI library loads and exports every declared symbol, the
config loader, the plugin surface (ctor/forward signature, state-dict keys, checkpoint files),
and loud failure without a GPU."""
import inspect
import os

import numpy as np
import pytest
import torch


def test_library_exports_every_declared_symbol():
    from llm_bci_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    l = _lib.lib()
    names = _lib.exported_symbols()
    assert len(names) >= 25 and "nbci_gemm" in names and "nbci_ndt1_forward" in names
    for n in names:
        assert hasattr(l, n), n
    assert l.nbci_version() == 100
    for n in names:
        assert n in _lib._SIGNATURES or n in ("nbci_version", "nbci_last_error", "nbci_gemm"), f"no ctypes signature for {n}"


def test_config_merge_and_include(tmp_path):
    from llm_bci_amd.config import DictConfig, ndt1_config, update_config
    inc = tmp_path / "model.yaml"
    inc.write_text("a: 1\nb:\n  c: 2\n  d: [1, 2]\n")
    top = tmp_path / "top.yaml"
    top.write_text(f"model: include:{inc}\nx: 5\n")
    cfg = update_config(str(top), {"model": {"b": {"c": None, "e": 7}}, "y": {"z": 1}})
    assert cfg.model.a == 1 and cfg.model.b.c is None and cfg.model.b.d == [1, 2] and cfg.model.b.e == 7 and cfg.y.z == 1
    assert isinstance(cfg.model, DictConfig)
    d = ndt1_config({"encoder": {"transformer": {"n_layers": 2}}})
    assert d.encoder.transformer.n_layers == 2 and d.encoder.transformer.hidden_size == 1024
    assert d.encoder.embedder.stack.size == 32 and d.encoder.smooth_and_noise.smooth_sd == 2


def test_plugin_surface_matches_reference_contract(tmp_path):
    from llm_bci_amd.model_output import ModelOutput, NDT1Output
    from llm_bci_amd.ndt1 import NDT1
    from llm_bci_amd.trainer import NAME2MODEL
    assert NAME2MODEL["NDT1"] is NDT1
    # forward kwarg names drive the reference's collate (trainer.py:161-171)
    assert list(inspect.signature(NDT1.forward).parameters)[1:] == [
        "spikes", "spikes_mask", "spikes_timestamp", "spikes_lengths", "targets", "targets_lengths", "block_idx", "day_idx"]
    assert list(NDT1Output().to_dict().keys()) == ["loss", "n_examples", "mask", "preds", "targets"]
    assert issubclass(NDT1Output, ModelOutput)
    over = {"encoder": {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
                        "transformer": {"n_layers": 1, "hidden_size": 32, "n_heads": 2, "inter_size": 48}}}
    m = NDT1(over, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    keys = list(m.state_dict().keys())
    assert keys[0] == "encoder.embedder.embed_spikes.weight" and keys[-1] == "decoder.0.bias"
    assert "encoder.layers.0.attn.out_proj.weight" in keys and "encoder.out_norm.weight" in keys
    assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in m.state_dict().values())
    # parameters are views into one flat buffer; checkpoint files use the reference's names
    assert all(p.untyped_storage().data_ptr() == m._flat.untyped_storage().data_ptr() for p in m.parameters())
    m.save_checkpoint(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["decoder.bin", "encoder.bin", "encoder_config.pth"]
    enc_cfg = torch.load(os.path.join(tmp_path, "encoder_config.pth"), weights_only=False)
    assert enc_cfg["transformer"]["hidden_size"] == 32
    over2 = {"encoder": dict(over["encoder"], from_pt=str(tmp_path))}
    m2 = NDT1(over2, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)   # warm start (ndt1.py:468-476)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(Exception, match="not implemented"):
        NDT1(over, method_name="mlm", vocab_size=11, blank_id=0, zero_infinity=True)


def test_no_cpu_fallback():
    from llm_bci_amd._lib import NbciUnavailable
    from llm_bci_amd.ndt1 import NDT1
    over = {"encoder": {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
                        "transformer": {"n_layers": 1, "hidden_size": 32, "n_heads": 2, "inter_size": 48}}}
    m = NDT1(over, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    B, T = 2, 12
    with pytest.raises(NbciUnavailable):
        m(spikes=torch.zeros(B, T, 16), spikes_mask=torch.ones(B, T, dtype=torch.long),
          spikes_timestamp=torch.zeros(B, T, dtype=torch.long), spikes_lengths=torch.full((B,), T))


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _dn, fn in os.walk(os.path.join(root, "llm_bci_amd")):
        for f in fn:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_schedules_match_golden_lr_and_momentum():
    from llm_bci_amd.schedule import OneCycle
    from test_oracle_golden import load
    fx = load("g_tiny")
    oc = OneCycle(100, 1e-3, 0.0, 25)
    for s in range(2):
        lr, b1 = oc.at(s)
        assert abs(lr - float(fx[f"lr_step{s}"])) < 1e-12 and abs(b1 - float(fx[f"beta1_step{s}"])) < 1e-12


def test_product_collate_matches_reference_fixture_and_oracle():
    """llm_bci_amd.collate reproduces the reference's pad_collate_fn output stored in the golden fixture, and the
    oracle's restatement on left/right padding, truncate and min_length."""
    from llm_bci_amd import collate as PC
    from oracle import collate as OC
    from test_oracle_golden import load
    fx = load("g_tiny")
    g = np.random.default_rng(0)
    rows = []
    for L, S in zip([30, 22, 17], [5, 4, 2]):
        rows.append({"spikes": g.standard_normal((L, 16)).astype(np.float32), "phonemes_idx": g.integers(1, 11, (S,)).astype(np.int64),
                     "sentence": "x"})
    items = [PC.item_from_row(r, targets_name="phonemes_idx") for r in rows]
    names = ["spikes", "spikes_mask", "spikes_timestamp", "spikes_lengths", "targets", "targets_lengths"]
    batch, unused = PC.pad_collate_fn(items, names, OC.CTC_PAD)
    for k in names:
        assert np.array_equal(batch[k].numpy(), fx["in_" + k]), k
    assert unused["sentence"] == ["x"] * 3 and "targets_mask" in unused
    arrs = [np.arange(n * 2, dtype=np.float32).reshape(n, 2) + 1 for n in (5, 3, 7)]
    for kw in (dict(side="left"), dict(side="right", truncate=4), dict(side="left", truncate=6, min_length=2),
               dict(side="right", min_length=9, truncate=10), dict(side="left", value=-1, min_length=8, truncate=8)):
        a = PC.padded_array(arrs, dim=0, **kw).numpy()
        b = OC.padded_array(arrs, dim=0, **kw)
        assert np.array_equal(a, b), kw


def test_linear_step_and_onecycle_schedules_match_torch_and_hf_over_a_whole_run():
    """trainer.py:233-253: "linear" = transformers.get_linear_schedule_with_warmup, "step" = StepLR(step_size=1, gamma) stepped per
    epoch (trainer.py:418-419), "cosine" = OneCycleLR (also with a warm-up leg, pct_start 0.15 as trainer_ssl_itransformer.yaml).
    The values each optimizer.step() uses, for every step of a run, from the libraries themselves."""
    import torch
    from torch.optim.lr_scheduler import OneCycleLR, StepLR
    from transformers import get_linear_schedule_with_warmup
    from llm_bci_amd.schedule import LinearWarmup, OneCycle, StepDecay

    def opt():
        return torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=2e-3)

    total, warm_pct = 57, 0.15
    o = opt(); sch = get_linear_schedule_with_warmup(o, num_warmup_steps=round(warm_pct * total), num_training_steps=total)
    mine = LinearWarmup(total, 2e-3, round(warm_pct * total))
    for s in range(total):
        lr, b1 = mine.at(s)
        assert abs(lr - o.param_groups[0]["lr"]) < 1e-15 and b1 == 0.9, s
        o.step(); sch.step()
    o = opt(); sch = OneCycleLR(o, total_steps=total, max_lr=2e-3, pct_start=warm_pct, div_factor=25.0)
    mine = OneCycle(total, 2e-3, warm_pct, 25.0)
    for s in range(total):
        lr, b1 = mine.at(s)
        assert abs(lr - o.param_groups[0]["lr"]) < 1e-15 and abs(b1 - o.param_groups[0]["betas"][0]) < 1e-12, s
        o.step(); sch.step()
    o = opt(); sch = StepLR(o, step_size=1, gamma=0.95)
    mine = StepDecay(2e-3, 0.95)
    for epoch in range(6):
        for s in range(5):
            lr, b1 = mine.at(epoch * 5 + s)
            assert abs(lr - o.param_groups[0]["lr"]) < 1e-15 and b1 == 0.9
            o.step()
        sch.step(); mine.end_epoch()


def test_bridge_shadow_and_forward_stamp_logic():
    """flat.bridge_begin rebuilds the bf16 shadow exactly when a parameter's version counter moved (an external optimizer's
    in-place step), bridge_check refuses a backward whose forward is no longer the model's most recent one."""
    import torch
    from llm_bci_amd._lib import NBCI_BF16
    from llm_bci_amd.flat import bridge_begin, bridge_check, bridge_stamp

    class Fake:
        compute_dtype = NBCI_BF16
        def __init__(self):
            self._flat = torch.arange(8, dtype=torch.float32)
            self._param_list = [torch.nn.Parameter(self._flat[0:4]), torch.nn.Parameter(self._flat[4:8])]
            for p, (a, b) in zip(self._param_list, ((0, 4), (4, 8))):
                p.data = self._flat[a:b]
            self._flat_lp, self.n_refresh = None, 0
        def refresh_lp(self):
            self._flat_lp = self._flat.to(torch.bfloat16); self.n_refresh += 1

    m = Fake()
    bridge_begin(m); bridge_begin(m)
    assert m.n_refresh == 1
    opt = torch.optim.SGD(m._param_list, lr=1.0)
    for p in m._param_list:
        p.grad = torch.ones_like(p)
    opt.step()
    bridge_begin(m)
    assert m.n_refresh == 2 and torch.equal(m._flat_lp.float(), m._flat)
    a = bridge_stamp(m); bridge_check(m, a, "t")
    bridge_stamp(m)
    with pytest.raises(RuntimeError):
        bridge_check(m, a, "t")


def test_lora_fallback_is_lora_by_its_published_definition():
    """llm_bci_amd/lora.py stands in for peft (absent from the image; models/bci.py:10,56-62): y = base(x) + (alpha/r) B A x,
    B = 0 at init (the wrapped model starts out identical), only adapter tensors trainable, peft's parameter names."""
    import torch
    from transformers import AutoModelForCausalLM, LlamaConfig
    from llm_bci_amd.lora import LoRALinear, inject_lora
    torch.manual_seed(0)
    base = torch.nn.Linear(12, 7)
    l = LoRALinear(base, r=3, alpha=6, dropout=0.0)
    x = torch.randn(5, 12)
    assert torch.equal(l(x), base(x))                                  # B = 0
    l.lora_B["default"].weight.data.normal_()
    A, B = l.lora_A["default"].weight, l.lora_B["default"].weight
    assert torch.allclose(l(x), base(x) + 2.0 * (x @ A.T @ B.T), atol=1e-6)
    llm = AutoModelForCausalLM.from_config(LlamaConfig(vocab_size=64, hidden_size=16, intermediate_size=32, num_hidden_layers=2,
                                                       num_attention_heads=2, num_key_value_heads=2))
    n_base = sum(p.numel() for p in llm.parameters())
    inject_lora(llm, 4, 8, 0.1, ["q_proj", "v_proj"])
    tr = {n for n, p in llm.named_parameters() if p.requires_grad}
    assert len(tr) == 2 * 2 * 2 and all(n.endswith(("lora_A.default.weight", "lora_B.default.weight")) for n in tr)
    assert sum(p.numel() for n, p in llm.named_parameters() if not p.requires_grad) == n_base
    with pytest.raises(ValueError):
        inject_lora(llm, 4, 8, 0.0, ["no_such_module"])


def _tiny_bci(lora):
    from llm_bci_amd.bci import BCI
    over = {"ndt1": {"encoder": {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
                                 "transformer": {"n_layers": 1, "hidden_size": 32, "n_heads": 2, "inter_size": 48}}},
            "projector": {"inter_size": 24}}
    torch.manual_seed(0)
    return BCI(over, lora=lora, debug=True)


def test_lora_checkpoint_is_adapter_only_and_round_trips_without_peft(tmp_path):
    """the no-peft fallback used to write a full HF checkpoint with wrapper key names that from_pretrained re-initialises
    with a warning. BCI.save_checkpoint now writes peft's adapter-only format (reference bci.py:252 under peft) and load_checkpoint
    reloads the LLM side too (bci.py:262)."""
    import json
    from safetensors.torch import load_file, save_file
    from llm_bci_amd import lora as L
    cfg = dict(r=4, alpha=8, dropout=0.0, target_modules=["q_proj", "v_proj"], modules_to_save=None)
    m = _tiny_bci(cfg)
    if not L.has_injected_lora(m.llm):
        pytest.skip("peft is installed: its own save_pretrained / from_pretrained handle the adapter")
    for n, p in m.llm.named_parameters():
        if "lora_B" in n:
            p.data.normal_(0, 0.1)
    d = str(tmp_path / "ck")
    m.save_checkpoint(d)
    files = sorted(os.listdir(d))
    assert "adapter_config.json" in files and "adapter_model.safetensors" in files
    assert not any(f.startswith("model") or f == "config.json" for f in files)            # adapter ONLY, as peft writes it
    sd = load_file(os.path.join(d, "adapter_model.safetensors"))
    assert all(k.startswith("base_model.model.") and (k.endswith("lora_A.weight") or k.endswith("lora_B.weight")) for k in sd)
    assert len(sd) == 2 * 2 * 2                                                             # 2 layers x {q,v} x {A,B}
    ac = json.load(open(os.path.join(d, "adapter_config.json")))
    assert ac["peft_type"] == "LORA" and ac["r"] == 4 and ac["lora_alpha"] == 8 and ac["target_modules"] == ["q_proj", "v_proj"]
    m2 = _tiny_bci(cfg)
    m2.load_checkpoint(d)
    a, b = m.llm.state_dict(), m2.llm.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    # into a model WITHOUT wrappers: injected from adapter_config.json, then loaded
    m3 = _tiny_bci(None)
    assert not L.has_injected_lora(m3.llm)
    m3.load_checkpoint(d)
    c = m3.llm.state_dict()
    assert a.keys() == c.keys() and all(torch.equal(a[k], c[k]) for k in a if "lora_" in k)
    # nothing is silently dropped or re-initialised
    bad = dict(sd); bad["base_model.model.model.layers.0.mlp.up_proj.lora_A.weight"] = torch.zeros(4, 32)
    save_file(bad, os.path.join(d, "adapter_model.safetensors"))
    with pytest.raises(KeyError):
        _tiny_bci(cfg).load_checkpoint(d)
    few = {k: v for k, v in sd.items() if "layers.1" not in k}
    save_file(few, os.path.join(d, "adapter_model.safetensors"))
    with pytest.raises(KeyError):
        _tiny_bci(cfg).load_checkpoint(d)
    # adapter variants whose arithmetic this loader does not implement are refused by name, not loaded with the wrong scaling
    save_file(sd, os.path.join(d, "adapter_model.safetensors"))
    assert isinstance(ac["lora_alpha"], int)                                                # as peft writes an integral alpha
    for key, val in (("use_rslora", True), ("use_dora", True), ("rank_pattern", {"q_proj": 8}), ("alpha_pattern", {"q_proj": 4}),
                     ("bias", "all"), ("modules_to_save", ["lm_head"])):
        json.dump(dict(ac, **{key: val}), open(os.path.join(d, "adapter_config.json"), "w"))
        with pytest.raises(ValueError, match=key):
            L.load_adapter(_tiny_bci(cfg).llm, d)
    json.dump(ac, open(os.path.join(d, "adapter_config.json"), "w"))
    L.load_adapter(_tiny_bci(cfg).llm, d)


def test_bci_joint_flat_layout_on_cpu():
    """BCI's native layout [ndt1 | projector | trainable LLM tensors]: contiguous 8-aligned segments, ndt1 / projector parameters
    re-homed as views, adapter masters copied in and written back in the adapters' dtype."""
    import torch
    from transformers import AutoModelForCausalLM, LlamaConfig
    from llm_bci_amd.bci import BCI
    over = {"encoder": {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
                        "transformer": {"n_layers": 2, "hidden_size": 32, "n_heads": 2, "inter_size": 48}}}
    llm = AutoModelForCausalLM.from_config(LlamaConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                                                       num_attention_heads=4, num_key_value_heads=4))
    llm = BCI._add_lora(llm, dict(r=4, alpha=8, dropout=0.0, target_modules=["q_proj", "v_proj"], modules_to_save=[]))
    m = BCI({"projector": {"stacking": 2, "inter_size": 48}, "ndt1": over}, llm=llm, compute_dtype="bf16")
    keys0 = set(m.state_dict())
    w_before = m.projector.state_dict()["0.weight"].clone()
    segs = m._segments
    assert segs[0][0] == 0 and segs[-1][1] == m._total and all(a[1] == b[0] for a, b in zip(segs[:-1], segs[1:]))
    assert len(segs) == len(m.ndt1._segments) + 2 and segs[:len(m.ndt1._segments)] == m.ndt1._segments
    for (name, off, numel, shape, seg) in m._layout:
        assert off % 8 == 0 and segs[seg][0] <= off and off + numel <= segs[seg][1], name
    assert set(m.state_dict()) == keys0 and torch.equal(m.projector.state_dict()["0.weight"], w_before)
    lo, hi = m._flat.data_ptr(), m._flat.data_ptr() + 4 * m._total
    for p in list(m.ndt1.parameters()) + list(m.projector.parameters()):
        assert lo <= p.data_ptr() < hi
    ee = m._native["eentries"]
    assert len(ee) == 8 and all(p.dtype == torch.float16 for _n, p, _o in ee)   # llm.to(float16) (bci.py:71) took the adapters along
    n, p, off = ee[0]
    assert torch.equal(m._flat[off:off + p.numel()].view(p.shape), p.detach().float())
    m._flat[off:off + p.numel()] += 0.25
    m._after_optimizer_step()
    assert torch.equal(p.detach(), m._flat[off:off + p.numel()].view(p.shape).to(torch.float16))
    assert m._flat_lp.dtype == torch.bfloat16 and m.ndt1._flat_lp.data_ptr() == m._flat_lp.data_ptr()


def test_fp8_oracle_rounding_is_ocp_e4m3():
    """oracle/fp8.py against torch's float8_e4m3fn (the OCP format gfx950 implements): values and codes, incl. subnormals and saturation"""
    import torch
    from oracle import fp8 as OF
    g = np.random.default_rng(0)
    v = np.concatenate([g.standard_normal(50000).astype(np.float32) * 40, g.standard_normal(20000).astype(np.float32) * 0.01,
                        np.array([0.0, 448.0, -448.0, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 464.0, 1000.0], np.float32)])
    t = torch.from_numpy(np.clip(v, -448, 448)).to(torch.float8_e4m3fn)
    assert np.array_equal(OF.e4m3_round(v), t.float().numpy())
    assert np.array_equal(OF.e4m3_encode(OF.e4m3_round(v)) & 0x7F, t.view(torch.uint8).numpy() & 0x7F)
    x = g.standard_normal((7, 96)).astype(np.float32)
    d, codes, sb = OF.mx_quantize(x)
    amax = np.abs(x.reshape(7, 3, 32)).max(-1)
    E = sb.astype(np.int64) - 127
    base = np.floor(np.log2(amax)).astype(np.int64) - 8
    assert ((E == base) | (E == base + 1)).all() and (amax / np.exp2(E) <= 448.0).all() and (amax / np.exp2(E) > 224.0).all()
    assert (np.abs(d - x) <= 2.0 ** -4 * np.abs(x) + 2.0 ** -9 * np.exp2(E).repeat(32, -1).reshape(x.shape) + 1e-12).all()   # 3 mantissa bits, no saturation


def test_patchtst_fp8_rejects_widths_the_fp8_gemm_cannot_take():
    """d_model > 512 with compute_dtype fp8 used to fail in the middle of the first layer's forward; now at construction."""
    from llm_bci_amd.patchtst import PatchTSTForSpikingActivity
    with pytest.raises(Exception, match="fp8 needs d_model"):
        PatchTSTForSpikingActivity({"encoder": {"d_model": 640, "num_attention_heads": 8}}, method_name="ctc", vocab_size=11, blank_id=0,
                                   zero_infinity=True, compute_dtype="fp8")


def test_ndt1_residual_dtype_needs_the_bf16_path():
    """NDT1(residual_dtype="bf16") stores the residual / gradient streams in bf16 between kernels: only with compute_dtype bf16
    (the fp32 parity path keeps everything in f32); the choice reaches the plan config."""
    from llm_bci_amd.ndt1 import NDT1
    kw = dict(method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    with pytest.raises(Exception, match="residual_dtype 'bf16' needs compute_dtype 'bf16'"):
        NDT1({}, compute_dtype="fp32", residual_dtype="bf16", **kw)
    assert NDT1({}, compute_dtype="bf16", **kw)._ccfg.residual_dtype == 0            # default = the parity setting (reference ndt1.py:325,328)
    assert NDT1({}, compute_dtype="bf16", residual_dtype="bf16", **kw)._ccfg.residual_dtype == 1   # opt-in
    assert NDT1({}, compute_dtype="bf16", residual_dtype="fp32", **kw)._ccfg.residual_dtype == 0
    assert NDT1({}, compute_dtype="fp32", **kw)._ccfg.residual_dtype == 0
