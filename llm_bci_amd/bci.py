"""BCI coupler behind the reference's surface (models/bci.py): NDT1 encoder (HIP) -> `projector` MLP
(HIP GEMMs with fused bias / activation) -> splice into the LLM's token embeddings (HIP gather kernel)
-> stock Hugging Face causal LM (fp16, frozen or LoRA'd; NOT re-implemented) -> shifted CE sum.

Same constructor keywords, `prepare_embeds` / `forward` signatures, `BCIOutput` fields and checkpoint
files (`projector.bin`, `projector_config.pth`, NDT1 files, `llm.save_pretrained`) as the reference
(bci.py:31-264). The projector follows configs/bci.yaml (Linear -> act -> Linear, `inter_size: null` =
single Linear); the 41-input tanh/no-bias variant of configs/phoneme_coupler.yaml is the same kernels
with another config (`PhonemeCoupler`).

Two ways to train it:
  * the reference's Trainer (registry swap): `model(**batch).loss.backward()` through the autograd bridges below;
  * `NativeTrainer` (llm_bci_amd/trainer.py): encoder + projector + the LLM's trainable tensors (LoRA adapters) live in ONE flat
    f32 buffer `[ndt1 | projector | llm-trainable]`, so the bucketed RCCL all-reduce, the fused AdamW and the bf16 shadow cover
    all of them (`_run_forward` / `_run_backward`, no autograd outside the stock LLM).
"""
import math
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT, NBCI_BF16, NBCI_F32, check, lib
from .config import DictConfig, update_config
from .flat import bridge_begin, bridge_check
from .model_output import ModelOutput
from .ndt1 import NDT1, _ptr, _stream


@dataclass
class BCIOutput(ModelOutput):
    preds: Optional[torch.Tensor] = None
    targets: Optional[torch.Tensor] = None


def bci_defaults():
    from .config import ndt1_defaults
    return dict(model_class="BCI", from_pt=None, projector=dict(stacking=1, inter_size=2048, bias=True, act="relu"),
                ndt1=ndt1_defaults())


class _EncodeFn(torch.autograd.Function):
    """NDT1 encoder as a feature extractor (bci.py:125): hidden states out, gradient of them in."""

    @staticmethod
    def forward(ctx, model, batch, *params):
        B, T, _ = batch["spikes"].shape
        Tp = model.tokens(T)
        dt = torch.bfloat16 if model.compute_dtype == NBCI_BF16 else torch.float32
        hidden = torch.empty(B, Tp, model._ccfg.factors_size or model._ccfg.hidden, dtype=dt, device=batch["spikes"].device)
        tmask = torch.empty(B, Tp, dtype=torch.int32, device=hidden.device)
        bridge_begin(model)   # torch.optim steps the f32 views in place: rebuild the bf16 shadow when they moved
        model._run_forward(batch, want_grad=True, hidden_out=hidden, token_mask_out=tmask)
        ctx.model, ctx.fwd_id = model, model._fwd_id
        ctx.mark_non_differentiable(tmask)
        return hidden, tmask

    @staticmethod
    def backward(ctx, g_hidden, _g_mask):
        m = ctx.model
        bridge_check(m, ctx.fwd_id, "_EncodeFn")
        grads = torch.zeros_like(m._flat)
        m._run_backward(grads, d_hidden=g_hidden.float().contiguous())
        out = [None, None]
        for (_, off, numel, shape, _seg) in m._layout:
            out.append(grads[off:off + numel].view(shape))
        return tuple(out)


# ------------------------------------------------------------------------------------------------------------------------
# projector math on nbci_gemm: ONE implementation for the autograd bridge and for the native train step
# ------------------------------------------------------------------------------------------------------------------------
def _proj_forward(x, w1, b1, w2, b2, act, want_grad):
    """y = act(x W1^T + b1) W2^T + b2 (bci.py:88-94), or y = x W1^T + b1 when w2 is None (:96). x, w* in the compute dtype, b* f32.
    The first GEMM stores act'(pre-activation) beside its output, so the backward needs no activation kernel."""
    M, K = x.shape
    d = ops._dt(x)
    if w2 is None:
        y = torch.empty(M, w1.shape[0], dtype=x.dtype, device=x.device)
        ops.gemm(M, w1.shape[0], K, ops.operand(x, K, True), ops.operand(w1, K, True), y, w1.shape[0], in_dtype=d, c_dtype=d, bias=b1)
        return y, None, None
    I, N = w1.shape[0], w2.shape[0]
    h = torch.empty(M, I, dtype=x.dtype, device=x.device)
    dact = torch.empty_like(h) if (want_grad and act) else None
    ops.gemm(M, I, K, ops.operand(x, K, True), ops.operand(w1, K, True), h, I, in_dtype=d, c_dtype=d, bias=b1, act=act,
             C2=dact, c2_grad=1 if dact is not None else 0)
    y = torch.empty(M, N, dtype=x.dtype, device=x.device)
    ops.gemm(M, N, I, ops.operand(h, I, True), ops.operand(w2, I, True), y, N, in_dtype=d, c_dtype=d, bias=b2)
    return y, h, dact


def _proj_backward(g, x, h, dact, w1, w2, gw1, gb1, gw2, gb2, need_dx=True):
    """Gradients of _proj_forward. g = dL/dy (M, N) compute dtype. gw* / gb* are f32 views that are ACCUMULATED into (beta = 1:
    gradient accumulation and the flat gradient buffer both want +=). The activation gate and the first Linear's bias sums ride in
    the epilogue of the GEMM that produces dL/d(pre-activation); only the last Linear's bias needs its own column-sum pass (its
    output gradient comes from the LLM's autograd, not from one of our GEMMs)."""
    M, K = x.shape
    d = ops._dt(x)
    g = g.contiguous()
    N = g.shape[1]
    if w2 is None:
        if gb1 is not None:
            check(lib().nbci_colsum(_ptr(g), d, N, M, N, _ptr(gb1), _stream()), "nbci_colsum")
        ops.gemm(N, K, M, ops.operand(g, N, False), ops.operand(x, K, False), gw1, K, in_dtype=d, c_dtype=NBCI_F32, beta=1.0)
        if not need_dx:
            return None
        dx = torch.empty(M, K, dtype=x.dtype, device=x.device)
        ops.gemm(M, K, N, ops.operand(g, N, True), ops.operand(w1, K, False), dx, K, in_dtype=d, c_dtype=d)
        return dx
    I = w1.shape[0]
    if gb2 is not None:
        check(lib().nbci_colsum(_ptr(g), d, N, M, N, _ptr(gb2), _stream()), "nbci_colsum")
    ops.gemm(N, I, M, ops.operand(g, N, False), ops.operand(h, I, False), gw2, I, in_dtype=d, c_dtype=NBCI_F32, beta=1.0)
    g1 = torch.empty(M, I, dtype=x.dtype, device=x.device)
    if dact is not None:
        ops.gemm(M, I, N, ops.operand(g, N, True), ops.operand(w2, I, False), g1, I, in_dtype=d, c_dtype=d,
                 gate=dact, ldg=I, gate_act=-1, colsum=gb1)
    else:
        ops.gemm(M, I, N, ops.operand(g, N, True), ops.operand(w2, I, False), g1, I, in_dtype=d, c_dtype=d, colsum=gb1)
    ops.gemm(I, K, M, ops.operand(g1, I, False), ops.operand(x, K, False), gw1, K, in_dtype=d, c_dtype=NBCI_F32, beta=1.0)
    if not need_dx:
        return None
    dx = torch.empty(M, K, dtype=x.dtype, device=x.device)
    ops.gemm(M, K, I, ops.operand(g1, I, True), ops.operand(w1, K, False), dx, K, in_dtype=d, c_dtype=d)
    return dx


class _ProjectorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, w1, b1, w2, b2):
        w1l = w1.to(x.dtype)
        w2l = w2.to(x.dtype) if w2 is not None else None
        b1f = b1.float().contiguous() if b1 is not None else None
        b2f = b2.float().contiguous() if b2 is not None else None
        x = x.contiguous()
        y, h, dact = _proj_forward(x, w1l, b1f, w2l, b2f, act, True)
        ctx.save_for_backward(x, h, dact, w1l, w2l)
        ctx.bias = (b1 is not None, b2 is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, h, dact, w1l, w2l = ctx.saved_tensors
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=x.device)
        gw1 = z(*w1l.shape)
        gb1 = z(w1l.shape[0]) if ctx.bias[0] else None
        gw2 = z(*w2l.shape) if w2l is not None else None
        gb2 = z(w2l.shape[0]) if (w2l is not None and ctx.bias[1]) else None
        dx = _proj_backward(gy.to(x.dtype), x, h, dact, w1l, w2l, gw1, gb1, gw2, gb2, need_dx=ctx.needs_input_grad[0])
        return dx, None, gw1, gb1, gw2, gb2


def _splice_forward(text, spikes, text_mask, spikes_valid, targets, split):
    B, Ts, H = spikes.shape
    Lt = text.shape[1]
    text = text.to(spikes.dtype).contiguous()
    spikes = spikes.contiguous()
    out = torch.empty(B, Lt + Ts, H, dtype=spikes.dtype, device=spikes.device)
    mask_out = torch.empty(B, Lt + Ts, dtype=torch.int64, device=spikes.device)
    tg_out = torch.empty(B, Lt + Ts, dtype=torch.int64, device=spikes.device) if targets is not None else None
    tm, sv, sp = text_mask.contiguous().long(), spikes_valid.contiguous().long(), split.contiguous().long()
    tg = targets.contiguous().long() if targets is not None else None
    check(lib().nbci_coupler_splice_fwd(_ptr(text), _ptr(spikes), _ptr(out), ops._dt(spikes), _ptr(tm), _ptr(sv), _ptr(mask_out),
                                        _ptr(tg), _ptr(tg_out), _ptr(sp), B, Lt, Ts, H, _stream()), "nbci_coupler_splice_fwd")
    return out, mask_out, tg_out, sp


def _splice_backward(g_out, sp, B, Lt, Ts, H, want_text):
    g_out = g_out.contiguous()
    d_text = torch.empty(B, Lt, H, dtype=g_out.dtype, device=g_out.device) if want_text else None   # every row is written
    d_sp = torch.empty(B, Ts, H, dtype=g_out.dtype, device=g_out.device)
    check(lib().nbci_coupler_splice_bwd(_ptr(g_out), _ptr(d_text), _ptr(d_sp), ops._dt(g_out), _ptr(sp), B, Lt, Ts, H, _stream()),
          "nbci_coupler_splice_bwd")
    return d_text, d_sp


class _SpliceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text, spikes, text_mask, spikes_valid, targets, split):
        out, mask_out, tg_out, sp = _splice_forward(text, spikes, text_mask, spikes_valid, targets, split)
        B, Ts, H = spikes.shape
        ctx.save_for_backward(sp)
        ctx.dims = (B, text.shape[1], Ts, H, text.dtype)
        ctx.mark_non_differentiable(mask_out)
        if tg_out is not None:
            ctx.mark_non_differentiable(tg_out)
        return out, mask_out, tg_out

    @staticmethod
    def backward(ctx, g_out, _gm, _gt):
        (sp,) = ctx.saved_tensors
        B, Lt, Ts, H, text_dtype = ctx.dims
        d_text, d_sp = _splice_backward(g_out, sp, B, Lt, Ts, H, ctx.needs_input_grad[0])
        return (d_text.to(text_dtype) if d_text is not None else None), d_sp, None, None, None, None


class _Lin(nn.Module):
    def __init__(self, fan_in, fan_out, bias):
        super().__init__()
        ref = nn.Linear(fan_in, fan_out, bias=bias)   # reference init (bci.py:88-96 builds nn.Linear)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone()) if bias else None


class Projector(nn.Module):
    """state-dict keys = the reference's nn.Sequential(Linear, act, Linear) ("0.weight", "2.weight", ...) or single Linear."""

    def __init__(self, in_size, inter_size, out_size, bias, act):
        super().__init__()
        self.act = ACT[act]
        if inter_size is not None:
            self.add_module("0", _Lin(in_size, inter_size, bias))
            self.add_module("2", _Lin(inter_size, out_size, bias))
            self.two = True
        else:
            lin = _Lin(in_size, out_size, bias)
            self.weight, self.bias = lin.weight, lin.bias
            self.two = False

    def tensors(self):
        """(w1, b1, w2, b2) parameters; w2 / b2 None for the single Linear."""
        if self.two:
            a, b = self._modules["0"], self._modules["2"]
            return a.weight, a.bias, b.weight, b.bias
        return self.weight, self.bias, None, None

    def forward(self, x2d):
        w1, b1, w2, b2 = self.tensors()
        return _ProjectorFn.apply(x2d, self.act, w1, b1, w2, b2)


class PhonemeCoupler(nn.Module):
    """The adapter MLP of configs/phoneme_coupler.yaml:1-7 (input_size 41 = phoneme vocabulary incl. BLANK and SIL -> inter_size
    2048 -> the LLM's hidden size, `act: tanh`, `bias: False`). The reference snapshot ships the config only (its
    models.phoneme_llm is absent), so this is the projector kernels under that config: forward((B, L, 41) phoneme
    features, e.g. the CTC head's log-probs) -> (B, L, H_llm)."""

    def __init__(self, config, llm_hidden_size, compute_dtype="bf16"):
        super().__init__()
        c = DictConfig(update_config(dict(input_size=41, inter_size=2048, act="tanh", bias=False, loss_reduction="sum"), config or {}))
        self.config = c
        self.dtype = torch.bfloat16 if compute_dtype in ("bf16", "bfloat16") else torch.float32
        self.projector = Projector(c.input_size, c.inter_size, llm_hidden_size, c.bias, c.act)

    def forward(self, feats):
        B, L, I = feats.shape
        return self.projector(feats.reshape(B * L, I).to(self.dtype)).view(B, L, -1)


def _shifted_ce(logits, targets, vocab):
    """bci.py:201-212: tokens < n predict n, CrossEntropyLoss(reduction="sum"), n_examples = #labels."""
    sl = logits[..., :-1, :].contiguous().view(-1, vocab)
    st = targets[..., 1:].contiguous().view(-1).to(sl.device)
    return nn.functional.cross_entropy(sl, st, reduction="sum"), (st != -100).sum()


class BCI(nn.Module):
    def __init__(self, config, llm_path=None, lora=None, freeze_llm=False, **kwargs):
        super().__init__()
        config = update_config(bci_defaults(), config if config is not None else {})
        pt_path = dict(config).pop("from_pt", None)
        if "llm" in kwargs:
            llm = kwargs.pop("llm")
        else:   # stock Hugging Face model, exactly as the reference builds it (bci.py:49-68)
            from transformers import AutoModelForCausalLM, LlamaConfig
            if kwargs.get("debug"):
                llm = AutoModelForCausalLM.from_config(LlamaConfig(num_hidden_layers=2, hidden_size=32, intermediate_size=32,
                                                                   num_attention_heads=4))
            else:
                llm = self._load_llm(pt_path, base_path=llm_path) if pt_path else AutoModelForCausalLM.from_pretrained(llm_path)
            if lora is not None and pt_path is None:
                llm = self._add_lora(llm, lora)
            if freeze_llm:
                for p in llm.parameters():
                    p.requires_grad = False
        kwargs.pop("debug", None)
        llm.to(torch.float16)
        self.llm = llm
        self.llm_config = llm.config
        ndt1_pt = pt_path or kwargs.pop("load_ndt1_from_pt", None)
        if ndt1_pt is not None:
            config["ndt1"]["encoder"]["from_pt"] = ndt1_pt
        nk = dict(kwargs)
        nk["method_name"] = "ctc"            # the encoder is all that is used (bci.py:125); head kept for checkpoint parity
        nk.setdefault("vocab_size", 41); nk.setdefault("blank_id", 0); nk.setdefault("zero_infinity", True)
        self.ndt1 = NDT1(config["ndt1"], **nk)
        if pt_path is not None:
            pc = torch.load(os.path.join(pt_path, "projector_config.pth"), weights_only=False)
            config["projector"] = update_config(config.projector, pc)
        pj = DictConfig(config["projector"])
        self.stacking = pj.stacking
        H = self.ndt1._ccfg.hidden   # the reference sizes the projector by transformer.hidden_size (bci.py:91,96) ...
        if self.ndt1._ccfg.factors_size not in (0, H):   # ... so a factors projection of another width cannot feed it there either
            raise ValueError("BCI: encoder.factors.size must equal transformer.hidden_size (the projector reads hidden_size inputs)")
        self.projector = Projector(H * self.stacking, pj.inter_size, llm.config.hidden_size, pj.bias, pj.act)
        if pt_path is not None:
            self.projector.load_state_dict(torch.load(os.path.join(pt_path, "projector.bin")))
        self.config = config
        self._native = None          # flat-buffer state of the native train step (built lazily, on the device)
        self._nat = None             # per-step state between _run_forward and _run_backward
        self.loss_scale = 1.0

    @staticmethod
    def _add_lora(llm, lora):
        lc = DictConfig(lora)
        try:
            from peft import LoraConfig, get_peft_model
        except ImportError:   # peft is not in this image: LoRA by its published definition (llm_bci_amd/lora.py)
            from .lora import inject_lora
            if lc.get("modules_to_save"):
                raise NotImplementedError("lora.modules_to_save needs peft")
            return inject_lora(llm, lc.r, lc.alpha, lc.dropout, lc.target_modules)
        return get_peft_model(llm, LoraConfig(inference_mode=False, r=lc.r, lora_alpha=lc.alpha, lora_dropout=lc.dropout,
                                              target_modules=lc.target_modules, modules_to_save=lc.modules_to_save))

    # ------------------------------------------------------------------------------------------------ autograd route
    def prepare_embeds(self, input_ids, attention_mask, input_split, spikes, spikes_mask, spikes_timestamp, spikes_lengths,
                       block_idx=None, day_idx=None, targets=None):
        text_embeds = self.llm.get_input_embeddings()(input_ids)                      # stock HF embedding lookup
        batch = dict(spikes=spikes, spikes_mask=spikes_mask, spikes_timestamp=spikes_timestamp, spikes_lengths=spikes_lengths,
                     targets=None, targets_lengths=None, day_idx=day_idx, block_idx=block_idx)
        hidden, tmask = _EncodeFn.apply(self.ndt1, batch, *self.ndt1._param_list)     # (B,T',H), (B,T')
        x2d, valid, B, Ts = self._stack(hidden, tmask, attention_mask.dtype)
        proj = self.projector(x2d).view(B, Ts, -1)
        embeds, mask, tg = _SpliceFn.apply(text_embeds, proj, attention_mask, valid, targets, input_split)
        return embeds, mask.to(attention_mask.dtype), tg

    def _stack(self, hidden, tmask, mask_dtype):
        """zero-pad to a multiple of `stacking`, view (B*T'/s, H*s); a stacked feature is valid iff none of its tokens is padding
        (bci.py:127-141)."""
        B, T, H = hidden.shape
        s = self.stacking
        if T % s != 0:
            new_T = math.ceil(T / s) * s
            hidden = torch.cat((hidden, hidden.new_zeros(B, new_T - T, H)), 1)
            tmask = torch.cat((tmask, tmask.new_zeros(B, new_T - T)), 1)
            T = new_T
        valid = (tmask.view(B, T // s, s).sum(-1) == s).to(mask_dtype)
        return hidden.reshape(B * (T // s), H * s), valid, B, T // s

    def forward(self, input_ids, attention_mask, input_split, spikes, spikes_mask, spikes_timestamp, spikes_lengths,
                block_idx=None, day_idx=None, targets=None):
        embeds, attention_mask, targets = self.prepare_embeds(input_ids, attention_mask, input_split, spikes, spikes_mask,
                                                              spikes_timestamp, spikes_lengths, block_idx, day_idx, targets)
        logits = self.llm(inputs_embeds=embeds.to(self.llm.dtype), attention_mask=attention_mask, return_dict=True).logits
        loss = n_examples = None
        if targets is not None:                                                       # shifted CE, reduction sum (bci.py:201-212)
            loss, n_examples = _shifted_ce(logits, targets, self.llm_config.vocab_size)
        return BCIOutput(loss=loss, n_examples=n_examples, preds=logits, targets=targets)

    # ------------------------------------------------------------------------------------------------ native train step
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)   # .to(device): the ndt1 re-flattens itself; the joint buffer is rebuilt on next use
        self._native = None
        self._nat = None
        return out

    def _ensure_native(self):
        """Build the joint flat layout [ndt1 | projector | trainable LLM tensors] and re-home every parameter into it.
        ndt1 / projector parameters become views (as in NDT1 itself); the LLM's trainable tensors keep their own (fp16) storage
        and get f32 MASTER copies in the buffer, written back after every optimizer step (`_after_optimizer_step`). That is
        mixed-precision training of the adapters; the reference steps its fp16 adapter weights directly (bci.py:71 +
        trainer.py:229) — a deliberate difference, stated in DESIGN.md."""
        if self._native is not None:
            return
        nd = self.ndt1
        dev = nd._flat.device
        n0 = nd._total
        layout = [("ndt1." + nm, off, n, shape, seg) for (nm, off, n, shape, seg) in nd._layout]
        segments = list(nd._segments)
        cur = (n0 + 7) // 8 * 8
        pseg = len(segments)
        pentries = []
        for name, p in self.projector.named_parameters():
            cur = (cur + 7) // 8 * 8
            pentries.append((name, p, cur))
            layout.append(("projector." + name, cur, p.numel(), tuple(p.shape), pseg))
            cur += p.numel()
        cur = (cur + 7) // 8 * 8
        segments.append((segments[-1][1], cur))
        extras = [(n, p) for n, p in self.llm.named_parameters() if p.requires_grad]
        eentries = []
        if extras:
            eb = cur
            for name, p in extras:
                cur = (cur + 7) // 8 * 8
                eentries.append((name, p, cur))
                layout.append(("llm." + name, cur, p.numel(), tuple(p.shape), pseg + 1))
                cur += p.numel()
            cur = (cur + 7) // 8 * 8
            segments.append((eb, cur))
        flat = torch.zeros(cur, dtype=torch.float32, device=dev)
        flat[:n0] = nd._flat
        with torch.no_grad():
            nd._flat = flat[:n0]
            for (nm, off, n, shape, _s), p in zip(nd._layout, nd._param_list):
                p.data = flat[off:off + n].view(shape)
            nd._flat_lp = None
            for name, p, off in pentries:
                flat[off:off + p.numel()] = p.detach().reshape(-1).float()
                p.data = flat[off:off + p.numel()].view(p.shape)
            for name, p, off in eentries:
                flat[off:off + p.numel()] = p.detach().reshape(-1).float()
        self._native = dict(flat=flat, layout=layout, segments=segments, total=cur, pentries=pentries, eentries=eentries,
                            pseg=pseg, lseg=(pseg + 1) if extras else None, lp=None)

    # what NativeTrainer / GradReducer read
    @property
    def _flat(self):
        self._ensure_native()
        return self._native["flat"]

    @property
    def _total(self):
        self._ensure_native()
        return self._native["total"]

    @property
    def _segments(self):
        self._ensure_native()
        return self._native["segments"]

    @property
    def _layout(self):
        self._ensure_native()
        return self._native["layout"]

    @property
    def _embed_split(self):
        return self.ndt1._embed_split

    @property
    def compute_dtype(self):
        return self.ndt1.compute_dtype

    @property
    def _flat_lp(self):
        self._ensure_native()
        if self.compute_dtype != NBCI_BF16:
            return None
        if self._native["lp"] is None:
            self.refresh_lp()
        return self._native["lp"]

    @property
    def _step_seed(self):
        return self.ndt1._step_seed

    @_step_seed.setter
    def _step_seed(self, v):
        self.ndt1._step_seed = v

    def refresh_lp(self):
        self._ensure_native()
        if self.compute_dtype == NBCI_BF16:
            lp = self._native["flat"].to(torch.bfloat16)
            self._native["lp"] = lp
            self.ndt1._flat_lp = lp[:self.ndt1._total]   # the encoder reads ITS part of the same shadow (the fused AdamW refreshes it)

    def _after_optimizer_step(self):
        """f32 masters of the LLM's trainable tensors -> the tensors the stock LLM computes with (their dtype, usually fp16)."""
        ee = self._native["eentries"]
        if ee:
            flat = self._native["flat"]
            with torch.no_grad():
                torch._foreach_copy_([p.data for _n, p, _o in ee], [flat[o:o + p.numel()].view(p.shape) for _n, p, o in ee])

    def _proj_views(self, buf):
        """(w1, b1, w2, b2) as views of `buf` (the f32 flat buffer, its bf16 shadow, or the flat gradient buffer)."""
        v = {}
        for name, p, off in self._native["pentries"]:
            v[name] = buf[off:off + p.numel()].view(p.shape)
        if self.projector.two:
            return v["0.weight"], v.get("0.bias"), v["2.weight"], v.get("2.bias")
        return v["weight"], v.get("bias"), None, None

    def _run_forward(self, batch, want_grad, seed=None, grad_scale=1.0):
        """forward of one micro-batch for NativeTrainer. batch = the reference forward's keyword arguments (bci.py:173-184).
        Returns (loss as a 1-element f32 vector = the CE SUM, logits)."""
        self._ensure_native()
        nd, nat = self.ndt1, self._native
        spikes = batch["spikes"]
        B, T, _ = spikes.shape
        Tp = nd.tokens(T)
        bf = self.compute_dtype == NBCI_BF16
        dt = torch.bfloat16 if bf else torch.float32
        if bf:   # (re)build the joint shadow when it is missing or the encoder's view of it was replaced (load_state_dict, autograd route)
            lp = nat["lp"]
            if lp is None or nd._flat_lp is None or nd._flat_lp.data_ptr() != lp.data_ptr():
                self.refresh_lp()
        dev = spikes.device
        hidden = torch.empty(B, Tp, nd._ccfg.factors_size or nd._ccfg.hidden, dtype=dt, device=dev)
        tmask = torch.empty(B, Tp, dtype=torch.int32, device=dev)
        enc = dict(spikes=spikes, spikes_mask=batch["spikes_mask"], spikes_timestamp=batch["spikes_timestamp"],
                   spikes_lengths=batch["spikes_lengths"], targets=None, targets_lengths=None, day_idx=batch.get("day_idx"),
                   block_idx=batch.get("block_idx"))
        nd.train(self.training)
        nd._run_forward(enc, want_grad=want_grad, seed=seed, hidden_out=hidden, token_mask_out=tmask)
        amask = batch["attention_mask"]
        x2d, valid, _B, Ts = self._stack(hidden, tmask, amask.dtype)
        w1, _b1, w2, _b2 = self._proj_views(nat["lp"] if bf else nat["flat"])
        _w1, b1f, _w2, b2f = self._proj_views(nat["flat"])
        x2d = x2d.contiguous()
        proj, h, dact = _proj_forward(x2d, w1, b1f, w2, b2f, self.projector.act, want_grad)
        emb_mod = self.llm.get_input_embeddings()
        text_grad = bool(want_grad and emb_mod.weight.requires_grad)
        with torch.set_grad_enabled(text_grad):
            text = emb_mod(batch["input_ids"])
        targets = batch.get("targets")
        embeds, mask, tg, sp = _splice_forward(text.detach(), proj.view(B, Ts, -1), amask, valid, targets, batch["input_split"])
        leaf = embeds.to(self.llm.dtype).detach().requires_grad_(bool(want_grad))
        with torch.set_grad_enabled(bool(want_grad)):
            logits = self.llm(inputs_embeds=leaf, attention_mask=mask.to(amask.dtype), return_dict=True).logits
            if tg is not None:
                loss, n_ex = _shifted_ce(logits, tg, self.llm_config.vocab_size)
            else:
                loss, n_ex = logits.new_zeros(()), torch.zeros((), dtype=torch.int64, device=dev)
        self.last_n_examples = n_ex.reshape(1)
        self.last_targets = tg
        self._nat = dict(leaf=leaf, loss=loss, text=text if text_grad else None, sp=sp, x2d=x2d, h=h, dact=dact, B=B, Ts=Ts, Tp=Tp,
                         Lt=text.shape[1], Hl=proj.shape[-1], grad_scale=grad_scale, stage=0, d_hidden=None, want_grad=want_grad,
                         keep=(hidden, tmask, proj, embeds, mask))
        return loss.detach().float().reshape(1), logits.detach()

    def _run_backward(self, grads, seg_hi=None, seg_lo=0, embed_part=0):
        """Backward of the last _run_forward into the flat gradient buffer, segments seg_hi..seg_lo (high to low: trainable LLM
        tensors, projector, then the ndt1's own segments head..embedder), so that each finished range can go on the wire."""
        nat, st, nd = self._native, self._nat, self.ndt1
        if st is None or not st["want_grad"]:
            raise RuntimeError("backward called but the forward pass ran without want_grad")
        last = len(nat["segments"]) - 1
        if seg_hi is None:
            seg_hi = last
        pseg = nat["pseg"]
        dt = torch.bfloat16 if self.compute_dtype == NBCI_BF16 else torch.float32
        if seg_hi >= pseg and st["stage"] == 0:
            # stage 1: the stock LLM's own autograd, from the CE sum back to its input embeddings and its trainable tensors
            ee = nat["eentries"]
            torch.autograd.backward(st["loss"] * st["grad_scale"], inputs=[st["leaf"]] + [p for _n, p, _o in ee])
            d_text, d_sp = _splice_backward(st["leaf"].grad.to(dt), st["sp"], st["B"], st["Lt"], st["Ts"], st["Hl"], st["text"] is not None)
            if st["text"] is not None:   # a trainable embedding table (full fine-tune): its gradient is the text part of the splice
                st["text"].backward(d_text.to(st["text"].dtype))
            for _n, p, off in ee:        # accumulate into the f32 flat gradient buffer (+=: gradient accumulation)
                if p.grad is not None:
                    grads[off:off + p.numel()].add_(p.grad.reshape(-1))
                    p.grad = None
            st["d_sp"] = d_sp.view(st["B"] * st["Ts"], st["Hl"])
            st["leaf"].grad = None
            st["stage"] = 1
        if seg_hi >= pseg and seg_lo <= pseg and st["stage"] == 1:
            bf = self.compute_dtype == NBCI_BF16
            w1, _b1, w2, _b2 = self._proj_views(nat["lp"] if bf else nat["flat"])
            gw1, gb1, gw2, gb2 = self._proj_views(grads)
            dx = _proj_backward(st["d_sp"], st["x2d"], st["h"], st["dact"], w1, w2, gw1, gb1, gw2, gb2, need_dx=True)
            B, Ts, Tp = st["B"], st["Ts"], st["Tp"]
            H = dx.shape[1] // self.stacking
            st["d_hidden"] = dx.view(B, Ts * self.stacking, H)[:, :Tp].float().contiguous()   # padded stacking rows carry nothing
            st["stage"] = 2
        hi = min(seg_hi, pseg - 1)
        if hi >= seg_lo:
            if st["stage"] != 2:
                raise RuntimeError("BCI backward: encoder segments requested before the projector segment")
            nd._run_backward(grads, hi, seg_lo, d_hidden=st["d_hidden"], embed_part=embed_part)

    # ------------------------------------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, save_dir):
        """Reference bci.py:250-257: `llm.save_pretrained` (with peft that is an ADAPTER-ONLY checkpoint), the NDT1 files,
        projector.bin / projector_config.pth. Without peft the LoRA'd LLM is saved in the same adapter-only format
        (llm_bci_amd/lora.py save_adapter), never as a full model with wrapper key names."""
        from .lora import has_injected_lora, save_adapter
        if has_injected_lora(self.llm):
            save_adapter(self.llm, save_dir)
        else:
            self.llm.save_pretrained(save_dir)
        self.ndt1.save_checkpoint(save_dir)
        torch.save({k: v.detach().clone() for k, v in self.projector.state_dict().items()}, os.path.join(save_dir, "projector.bin"))
        torch.save(dict(self.config.projector), os.path.join(save_dir, "projector_config.pth"))

    @staticmethod
    def _load_llm(load_dir, current=None, base_path=None):
        """The LLM of a checkpoint directory (reference bci.py:262: `AutoModelForCausalLM.from_pretrained(load_dir)`). A directory
        holding an adapter-only checkpoint needs peft for that call; without peft the adapter is loaded into `current` (wrappers
        injected when absent), or into a base model built from `base_path` / the adapter's `base_model_name_or_path`."""
        from .lora import is_adapter_dir, load_adapter
        from transformers import AutoModelForCausalLM
        if is_adapter_dir(load_dir):
            try:
                import peft  # noqa: F401
                return AutoModelForCausalLM.from_pretrained(load_dir)
            except ImportError:
                pass
            if current is None:
                import json
                base = base_path or json.load(open(os.path.join(load_dir, "adapter_config.json"))).get("base_model_name_or_path")
                if not base:
                    raise ValueError(f"{load_dir} is an adapter-only checkpoint: a base model path (llm_path) is needed to load it without peft")
                current = AutoModelForCausalLM.from_pretrained(base)
            return load_adapter(current, load_dir)
        return AutoModelForCausalLM.from_pretrained(load_dir)

    def load_checkpoint(self, load_dir):
        """Reference bci.py:259-264: the LLM (or its adapters) is reloaded too, then the NDT1 files and the projector. The joint flat
        buffer of the native train step is dropped: it is rebuilt on next use, which re-seeds the f32 masters of the LLM's trainable
        tensors from the tensors just loaded (NativeTrainer.load_checkpoint then restores the saved masters for an exact resume)."""
        dev, dt = next(self.llm.parameters()).device, self.llm.dtype
        self.llm = self._load_llm(load_dir, current=self.llm).to(dev).to(dt)
        self.llm_config = self.llm.config
        self.ndt1.load_checkpoint(load_dir)
        self.projector.load_state_dict(torch.load(os.path.join(load_dir, "projector.bin")))
        self._native = None
        self._nat = None

    # what NativeTrainer adds to trainer_state.pth for an exact resume: the f32 masters of the LLM's trainable tensors (the tensors
    # themselves are fp16 / bf16 roundings of them)
    def _resume_state(self):
        self._ensure_native()
        ee = self._native["eentries"]
        flat = self._native["flat"]
        return {"llm_masters": {n: flat[o:o + p.numel()].detach().cpu().clone() for n, p, o in ee}}

    def _load_resume_state(self, st):
        self._ensure_native()
        flat = self._native["flat"]
        masters = (st or {}).get("llm_masters", {})
        with torch.no_grad():
            for n, p, o in self._native["eentries"]:
                if n in masters:
                    flat[o:o + p.numel()] = masters[n].to(flat.device)
        self._native["lp"] = None
