"""GPU parity of the BCI coupler (llm_bci_amd/bci.py: HIP encoder -> HIP projector GEMMs -> HIP splice -> stock HF LLM -> shifted
CE) against the fixtures produced by the reference's BCI.prepare_embeds / BCI.forward (tests/golden/make_golden.py --bci), the
numpy oracle (oracle/bci.py) at the real coupler widths, and the native flat-buffer train step against the autograd route."""
import json
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from test_oracle_golden import load

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _StubLLM(nn.Module):
    """just enough of a HF causal LM for prepare_embeds: an embedding table, .config, .dtype"""

    def __init__(self, table):
        super().__init__()
        self.embed = nn.Embedding.from_pretrained(torch.from_numpy(table).clone(), freeze=False)
        self.config = types.SimpleNamespace(hidden_size=table.shape[1], vocab_size=table.shape[0])

    def get_input_embeddings(self):
        return self.embed

    @property
    def dtype(self):
        return self.embed.weight.dtype


def _build(fx, dtype):
    from llm_bci_amd.bci import BCI
    cfg = json.loads(str(fx["config_json"]))
    m = BCI(cfg, llm=_StubLLM(fx["embed_table"]), method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype=dtype)
    m.llm.float()
    m.ndt1.load_state_dict({k[len("w:ndt1."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w:ndt1.")})
    m.projector.load_state_dict({k[len("w:projector."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w:projector.")})
    return m.to(DEV)


def _inputs(fx):
    d = lambda k: torch.from_numpy(fx[k]).to(DEV)
    return (d("input_ids"), d("attention_mask"), d("input_split"), d("spikes"), d("spikes_mask"), d("spikes_timestamp"),
            d("spikes_lengths"), None, None, d("targets"))


def test_prepare_embeds_matches_reference_fp32():
    fx = load("g_bci")
    m = _build(fx, "fp32")
    m.eval()
    emb, mask, tg = m.prepare_embeds(*_inputs(fx))
    torch.cuda.synchronize()
    assert np.array_equal(mask.cpu().numpy(), fx["out_mask"])          # integer outputs: bit-exact
    assert np.array_equal(tg.cpu().numpy(), fx["out_targets"])
    np.testing.assert_allclose(emb.detach().float().cpu().numpy(), fx["out_embeds"], atol=1e-3)
    (emb.float() * torch.from_numpy(fx["R"]).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    for k in fx.files:
        if k.startswith("g:projector."):
            p = dict(m.projector.named_parameters())[k[len("g:projector."):]]
            ref = fx[k]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2e-3 * max(1.0, np.abs(ref).max()), err_msg=k)
        if k.startswith("g:ndt1."):
            p = dict(m.ndt1.named_parameters())[k[len("g:ndt1."):]]
            ref = fx[k]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2e-3 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_prepare_embeds_bf16_close():
    fx = load("g_bci")
    m = _build(fx, "bf16")
    m.eval()
    emb, mask, tg = m.prepare_embeds(*_inputs(fx))
    torch.cuda.synchronize()
    assert np.array_equal(mask.cpu().numpy(), fx["out_mask"]) and np.array_equal(tg.cpu().numpy(), fx["out_targets"])
    assert np.abs(emb.detach().float().cpu().numpy() - fx["out_embeds"]).max() < 0.06


def test_checkpoint_files(tmp_path):
    import os
    fx = load("g_bci")
    m = _build(fx, "fp32")
    m.llm.save_pretrained = lambda d: None     # the stub has no HF serialisation
    m.save_checkpoint(str(tmp_path))
    assert {"projector.bin", "projector_config.pth", "encoder.bin", "decoder.bin", "encoder_config.pth"} <= set(os.listdir(tmp_path))
    sd = torch.load(os.path.join(tmp_path, "projector.bin"))
    assert set(sd.keys()) == {"0.weight", "0.bias", "2.weight", "2.bias"}


# ------------------------------------------------------------------------------------------------------------------------
# BCI.forward end to end (models/bci.py:173-219) against the reference's own run, g_bci_fwd.npz
# ------------------------------------------------------------------------------------------------------------------------
def _build_fwd(fx, dtype, llm_dtype=torch.float32, lora=None):
    from transformers import AutoModelForCausalLM, LlamaConfig
    from llm_bci_amd.bci import BCI
    cfg = json.loads(str(fx["config_json"]))
    llm = AutoModelForCausalLM.from_config(LlamaConfig(**json.loads(str(fx["llm_config_json"]))))
    llm.load_state_dict({k[len("w:llm."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w:llm.")})
    if lora is not None:
        llm = BCI._add_lora(llm, lora)
    m = BCI(cfg, llm=llm, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype=dtype)
    m.llm.to(llm_dtype)
    m.ndt1.load_state_dict({k[len("w:ndt1."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w:ndt1.")})
    m.projector.load_state_dict({k[len("w:projector."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w:projector.")})
    return m.to(DEV)


def _check_grads(named, fx, prefix, tol):
    n = 0
    for k in fx.files:
        if k.startswith(prefix):
            ref = fx[k]
            got = named[k[len(prefix):]].grad.float().cpu().numpy()
            np.testing.assert_allclose(got, ref, atol=tol * max(1.0, np.abs(ref).max()), err_msg=k)
            n += 1
    return n


def test_bci_forward_matches_reference_fp32():
    """loss (shifted CE sum), n_examples, logits and the gradients of projector / encoder / LLM parameters of the reference's
    BCI.forward, with the LLM in fp32 so that the coupler's parity is visible at 1e-3."""
    fx = load("g_bci_fwd")
    m = _build_fwd(fx, "fp32")
    m.eval()
    out = m(*_inputs(fx))
    torch.cuda.synchronize()
    assert int(out.n_examples) == int(fx["n_examples"])
    assert np.array_equal(out.targets.cpu().numpy(), fx["out_targets"])
    np.testing.assert_allclose(out.preds.detach().float().cpu().numpy(), fx["f32_logits"], atol=1e-3)
    np.testing.assert_allclose(out.loss.item(), float(fx["f32_loss"]), rtol=1e-4)
    out.loss.backward()
    torch.cuda.synchronize()
    assert _check_grads(dict(m.projector.named_parameters()), fx, "g32:projector.", 2e-3) == 4
    assert _check_grads(dict(m.ndt1.named_parameters()), fx, "g32:ndt1.", 2e-3) >= 5
    assert _check_grads(dict(m.llm.named_parameters()), fx, "g32:llm.", 2e-3) >= 4


def test_bci_forward_reference_precision_fp16_llm_bf16_coupler():
    """exactly the reference's arrangement (LLM in fp16, bci.py:71,190) with the coupler in bf16: loss within 1 % of the
    reference's fp16 run, logits within fp16/bf16 noise."""
    fx = load("g_bci_fwd")
    m = _build_fwd(fx, "bf16", llm_dtype=torch.float16)
    m.eval()
    with torch.no_grad():
        out = m(*_inputs(fx))
    torch.cuda.synchronize()
    assert out.preds.dtype == torch.float16
    assert abs(out.loss.item() - float(fx["f16_loss"])) / float(fx["f16_loss"]) < 1e-2
    assert np.abs(out.preds.float().cpu().numpy() - fx["f16_logits"]).max() < 0.05
    assert int(out.n_examples) == int(fx["n_examples"])


LORA = dict(r=4, alpha=8, dropout=0.0, target_modules=["q_proj", "v_proj", "down_proj"], modules_to_save=[])


def _batch_dict(fx):
    d = lambda k: torch.from_numpy(fx[k]).to(DEV)
    return {k: d(k) for k in ("input_ids", "attention_mask", "input_split", "spikes", "spikes_mask", "spikes_timestamp",
                              "spikes_lengths", "targets")}


@pytest.mark.parametrize("lora", [None, LORA])
def test_native_step_gradients_equal_autograd_route(lora):
    """The flat-buffer step NativeTrainer drives (BCI._run_forward / _run_backward: encoder + projector + the LLM's trainable
    tensors in one buffer, no autograd outside the stock LLM) produces the gradients of the autograd route, segment by segment
    as the data-parallel trainer calls it."""
    fx = load("g_bci_fwd")
    m = _build_fwd(fx, "fp32", lora=lora)
    if lora is not None:
        torch.manual_seed(0)
        for n, p in m.llm.named_parameters():     # B = 0 at init would make every adapter-A gradient vanish
            if "lora_B" in n:
                p.data.normal_(0, 0.05)
    m.train()
    out = m(*_inputs(fx))
    out.loss.backward()
    ref = {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    loss, logits = m._run_forward(_batch_dict(fx), want_grad=True, grad_scale=1.0)
    grads = torch.zeros(m._total, device=DEV)
    for seg in range(len(m._segments) - 1, 0, -1):
        m._run_backward(grads, seg, seg)
    m._run_backward(grads, 0, 0, embed_part=1)
    m._run_backward(grads, 0, 0, embed_part=2)
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss.item(), out.loss.item(), rtol=1e-5)
    assert int(m.last_n_examples) == int(fx["n_examples"])
    seen = 0
    for (name, off, numel, shape, _seg) in m._layout:
        got = grads[off:off + numel].view(shape)
        if name in ref:
            r = ref[name]
            assert (got - r).abs().max().item() <= 2e-4 * max(1.0, r.abs().max().item()), name
            seen += 1
        else:       # the CTC decoder head is not on BCI's path: no gradient on either route
            assert name.startswith("ndt1.decoder.") and got.abs().max().item() == 0.0, name
    assert seen == len(ref)
    if lora is not None:
        assert any(n.startswith("llm.") and "lora_A" in n for (n, *_r) in m._layout)
        assert all(("lora_" in n) for (n, *_r) in m._layout if n.startswith("llm."))   # only the adapters are trainable


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_native_trainer_trains_bci_with_lora(dtype):
    """NativeTrainer on BCI: fused AdamW over the joint buffer, adapter masters written back into the fp16 LLM every step."""
    from llm_bci_amd.trainer import NativeTrainer
    fx = load("g_bci_fwd")
    m = _build_fwd(fx, dtype, llm_dtype=torch.float16, lora=LORA)
    a0 = {n: p.detach().clone() for n, p in m.llm.named_parameters() if "lora_" in n}
    base0 = m.llm.model.layers[0].self_attn.q_proj.base_layer.weight.detach().clone()
    w0 = dict(m.projector.named_parameters())["2.weight"].detach().clone()
    tr = NativeTrainer(m, lr=5e-3, wd=0.0, total_steps=80, compute_per=False)
    batch = _batch_dict(fx)
    losses = []
    for s in range(30):
        loss, _ = tr.train_step(batch, seed=s)
        losses.append(float(loss.sum()))
    torch.cuda.synchronize()
    st = tr.read_stats()
    assert st["n_examples"] == 30 * int(fx["n_examples"])
    assert np.all(np.isfinite(losses)) and losses[-1] < 0.93 * losses[0] and losses[-1] < losses[10] < losses[0], losses
    moved = [n for n, p in m.llm.named_parameters() if "lora_" in n and not torch.equal(p, a0[n])]
    assert len(moved) == len(a0)                                            # every adapter tensor was stepped ...
    assert torch.equal(m.llm.model.layers[0].self_attn.q_proj.base_layer.weight, base0)   # ... the frozen base was not
    assert not torch.equal(dict(m.projector.named_parameters())["2.weight"], w0)
    for n, p, off in m._native["eentries"]:                                 # fp16 tensors == rounding of their f32 masters
        assert torch.equal(p.detach(), m._flat[off:off + p.numel()].view(p.shape).to(p.dtype)), n
    if dtype == "bf16":
        assert torch.equal(m._flat_lp, m._flat.bfloat16())


def test_native_trainer_resume_restores_llm_adapters_and_masters(tmp_path):
    """ADVICE r2 (bci.py:575): save after 3 steps, 3 more steps; a FRESH LoRA'd BCI restored by NativeTrainer.load_checkpoint and fed
    the same steps ends on the same adapters / projector / encoder. Before the fix load_checkpoint left the LLM untouched and the
    stale f32 masters overwrote the adapters on the next step."""
    import os
    from llm_bci_amd.trainer import NativeTrainer
    from llm_bci_amd.lora import has_injected_lora
    fx = load("g_bci_fwd")
    batch = _batch_dict(fx)

    def fresh(seed):
        torch.manual_seed(seed)
        m = _build_fwd(fx, "bf16", llm_dtype=torch.float16, lora=LORA)
        return m, NativeTrainer(m, lr=5e-3, wd=0.0, total_steps=40, compute_per=False)

    m, tr = fresh(3)
    if not has_injected_lora(m.llm):
        pytest.skip("peft installed: adapter files are peft's own")
    for s in range(3):
        tr.train_step(batch, seed=s)
    tr.save_checkpoint(str(tmp_path))
    assert {"adapter_config.json", "adapter_model.safetensors", "trainer_state.pth", "projector.bin", "encoder.bin"} <= set(os.listdir(tmp_path))
    for s in range(3, 6):
        tr.train_step(batch, seed=s)
    torch.cuda.synchronize()
    want_llm = {n: p.detach().clone() for n, p in m.llm.named_parameters() if "lora_" in n}
    want_flat = m._flat.clone()

    m2, tr2 = fresh(99)                      # other adapter init (lora_A is random): everything must come from the checkpoint
    tr2.load_checkpoint(str(tmp_path))
    a3 = {n: p for n, p in m2.llm.named_parameters() if "lora_" in n}
    for n, p, off in m2._native["eentries"]:  # masters restored exactly, tensors = their fp16 rounding
        assert torch.equal(p.detach(), m2._flat[off:off + p.numel()].view(p.shape).to(p.dtype)), n
    for s in range(3, 6):
        tr2.train_step(batch, seed=s)
    torch.cuda.synchronize()
    assert tr2.opt_step == tr.opt_step == 6
    for n, p in a3.items():
        d = (p.float() - want_llm[n].float()).abs().max().item()
        assert d <= 2e-3, (n, d)             # (atomics / fp16 LLM backward order noise through Adam's normalised steps)
    d = (m2._flat - want_flat).abs()
    assert (d > 1e-4).float().mean() < 0.02 and d.max() < 2e-2, (d.max().item(), (d > 1e-4).float().mean().item())


# ------------------------------------------------------------------------------------------------------------------------
# the coupler at its real widths (configs/bci.yaml: 1024*s -> 2048 -> 4096) and the phoneme_coupler.yaml variant vs the oracle
# ------------------------------------------------------------------------------------------------------------------------
def _proj_case(in_size, inter, out, bias, act, M, dtype, seed=0):
    from llm_bci_amd.bci import Projector
    from oracle import bci as OB
    torch.manual_seed(seed)
    pj = Projector(in_size, inter, out, bias, act).to(DEV)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    g = np.random.default_rng(seed)
    x = torch.from_numpy(g.standard_normal((M, in_size)).astype(np.float32)).to(DEV).to(tdt).requires_grad_(True)
    R = torch.from_numpy(g.standard_normal((M, out)).astype(np.float32)).to(DEV).to(tdt)
    y = pj(x)
    y.backward(R)
    torch.cuda.synchronize()
    # oracle in f32 on the SAME (rounded) operands
    rnd = (lambda t: t.detach().to(tdt).float().cpu().numpy())
    p = {k: rnd(v) if k.endswith("weight") else v.detach().float().cpu().numpy() for k, v in pj.state_dict().items()}
    yo, c = OB.projector_fwd(rnd(x), p, act)
    go, dxo = OB.projector_bwd(rnd(R), p, c)
    tol = 2e-2 if dtype == "bf16" else 1e-3
    sc = lambda a: max(1.0, float(np.abs(a).max()))
    assert np.abs(y.detach().float().cpu().numpy() - yo).max() <= tol * sc(yo)
    assert np.abs(x.grad.float().cpu().numpy() - dxo).max() <= tol * sc(dxo)
    for k, v in pj.named_parameters():
        ref = go[k]
        got = v.grad.float().cpu().numpy()
        assert np.abs(got - ref).max() <= tol * sc(ref), k
        if dtype == "bf16":   # aggregate error well under the element bound
            assert np.abs(got - ref).sum() / max(1e-9, np.abs(ref).sum()) < 1.5e-2, k


@pytest.mark.parametrize("stacking,dtype", [(1, "bf16"), (2, "bf16"), (1, "fp32")])
def test_projector_real_widths_vs_oracle(stacking, dtype):
    """configs/bci.yaml widths: Linear(1024*s -> 2048) + ReLU + Linear(2048 -> 4096), biases on; rows = 2 samples x 143 tokens."""
    _proj_case(1024 * stacking, 2048, 4096, True, "relu", 286 // stacking, dtype)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_phoneme_coupler_variant_vs_oracle(dtype):
    """configs/phoneme_coupler.yaml:1-7: 41 -> 2048 -> H_llm, tanh, no bias (K = 41 is not a multiple of the MFMA k-step)."""
    _proj_case(41, 2048, 4096, False, "tanh", 2 * 60, dtype, seed=1)
    from llm_bci_amd.bci import PhonemeCoupler
    pc = PhonemeCoupler({}, 256, compute_dtype=dtype).to(DEV)
    assert set(pc.projector.state_dict()) == {"0.weight", "2.weight"} and pc.projector.state_dict()["0.weight"].shape == (2048, 41)
    y = pc(torch.randn(2, 7, 41, device=DEV).log_softmax(-1))
    assert y.shape == (2, 7, 256) and torch.isfinite(y.float()).all()


def test_prepare_embeds_real_widths_bf16_vs_oracle():
    """BASELINE configs[3] shapes for the encoder + coupler: default NDT1 (5 x 1024), 256 ch x 600 bins -> 143 tokens, stacking 1,
    projector 1024 -> 2048 -> 4096 (Llama-2-7B hidden), bf16, ragged lengths; the LLM is only an embedding table here (its
    arithmetic is not ours). Oracle = numpy NDT1 encoder + oracle/bci.py in f32 on the same weights."""
    from llm_bci_amd.bci import BCI
    from oracle import bci as OB
    from oracle import ndt1 as O
    torch.manual_seed(2)
    table = (torch.randn(64, 4096) * 0.02).numpy()
    cfg = {"projector": {"stacking": 1, "inter_size": 2048, "bias": True, "act": "relu"},
           "ndt1": {"encoder": {"smooth_and_noise": {"noise": False}, "embedder": {"dropout": 0.0}, "transformer": {"dropout": 0.0}}}}
    m = BCI(cfg, llm=_StubLLM(table), method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype="bf16")
    m.llm.float()
    p_nd = {k: v.detach().numpy().copy() for k, v in m.ndt1.state_dict().items()}
    p_pj = {k: v.detach().numpy().copy() for k, v in m.projector.state_dict().items()}
    m.to(DEV).eval()
    g = np.random.default_rng(3)
    B, T, Lt = 2, 600, 12
    lens = [600, 452]
    spikes = g.standard_normal((B, T, 256)).astype(np.float32)
    smask = np.zeros((B, T), np.int64); ts = np.zeros((B, T), np.int64)
    for b, L in enumerate(lens):
        spikes[b, L:] = 0; smask[b, :L] = 1; ts[b, :L] = np.arange(L)
    ids = g.integers(0, 64, (B, Lt)).astype(np.int64); am = np.ones((B, Lt), np.int64); split = np.array([4, 9], np.int64)
    tg = g.integers(0, 64, (B, Lt)).astype(np.int64)
    d = lambda a: torch.from_numpy(a).to(DEV)
    with torch.no_grad():
        emb, mask, tgo = m.prepare_embeds(d(ids), d(am), d(split), d(spikes), d(smask), d(ts), d(np.array(lens)), None, None, d(tg))
    torch.cuda.synchronize()
    out, _ = O.forward(O.make_config(noise=False, embed_dropout=0.0, dropout=0.0), p_nd,
                       dict(spikes=spikes, spikes_mask=smask, spikes_timestamp=ts, spikes_lengths=np.array(lens), targets=None,
                            targets_lengths=None), train=False, keep_cache=False)
    x, valid = OB.stack_tokens(out["enc_out"], out["token_mask"], 1)
    y, _c = OB.projector_fwd(x, p_pj, "relu")
    eo, mo, to = OB.splice_fwd(table[ids], y, am, valid, tg, split)
    assert np.array_equal(mask.cpu().numpy(), mo) and np.array_equal(tgo.cpu().numpy(), to)     # integer outputs: bit-exact
    diff = np.abs(emb.float().cpu().numpy() - eo)
    assert diff.max() < 0.08 and diff.mean() < 6e-3, (diff.max(), diff.mean())
