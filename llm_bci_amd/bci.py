"""BCI coupler behind the reference's surface (models/bci.py): NDT1 encoder (HIP) -> `projector` MLP
(HIP GEMMs with fused bias / activation) -> splice into the LLM's token embeddings (HIP gather kernel)
-> stock Hugging Face causal LM (fp16, frozen or LoRA'd; NOT re-implemented) -> shifted CE sum.

Same constructor keywords, `prepare_embeds` / `forward` signatures, `BCIOutput` fields and checkpoint
files (`projector.bin`, `projector_config.pth`, NDT1 files, `llm.save_pretrained`) as the reference
(bci.py:31-264). The projector follows configs/bci.yaml (Linear -> act -> Linear, `inter_size: null` =
single Linear); the 41-input tanh/no-bias variant of configs/phoneme_coupler.yaml is the same kernel
with another config.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT, NBCI_BF16, NBCI_F32, check, lib
from .config import DictConfig, update_config
from .model_output import ModelOutput
from .flat import bridge_begin, bridge_check
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


class _LinearActFn(torch.autograd.Function):
    """y = act(x W^T + b) on nbci_gemm; the forward stores act'(pre) so backward is three plain GEMMs."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        M, K = x.shape
        N = w.shape[0]
        wl = w.to(x.dtype)
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        dact = torch.empty_like(y) if act else None
        bf = b.float().contiguous() if b is not None else None
        d = ops._dt(x)
        ops.gemm(M, N, K, ops.operand(x, K, True), ops.operand(wl, K, True), y, N, in_dtype=d, c_dtype=d, bias=bf, act=act,
                 C2=dact, c2_grad=1 if act else 0)
        ctx.save_for_backward(x, wl, dact)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wl, dact = ctx.saved_tensors
        M, K = x.shape
        N = wl.shape[0]
        g = gy.contiguous()
        if dact is not None:
            g = g * dact                                   # elementwise gate (tiny next to the GEMMs)
        d = ops._dt(x)
        gx = torch.empty(M, K, dtype=x.dtype, device=x.device)
        ops.gemm(M, K, N, ops.operand(g, N, True), ops.operand(wl, K, False), gx, K, in_dtype=d, c_dtype=d)
        gw = torch.zeros(N, K, dtype=torch.float32, device=x.device)
        ops.gemm(N, K, M, ops.operand(g, N, False), ops.operand(x, K, False), gw, K, in_dtype=d, c_dtype=NBCI_F32)
        gb = g.float().sum(0) if ctx.has_bias else None
        return gx, gw, gb, None


class _SpliceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text, spikes, text_mask, spikes_valid, targets, split):
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
        ctx.save_for_backward(sp)
        ctx.dims = (B, Lt, Ts, H, text.requires_grad)
        ctx.mark_non_differentiable(mask_out)
        if tg_out is not None:
            ctx.mark_non_differentiable(tg_out)
        return out, mask_out, tg_out

    @staticmethod
    def backward(ctx, g_out, _gm, _gt):
        (sp,) = ctx.saved_tensors
        B, Lt, Ts, H, _ = ctx.dims
        g_out = g_out.contiguous()
        d_text = torch.zeros(B, Lt, H, dtype=g_out.dtype, device=g_out.device)
        d_sp = torch.zeros(B, Ts, H, dtype=g_out.dtype, device=g_out.device)
        check(lib().nbci_coupler_splice_bwd(_ptr(g_out), _ptr(d_text), _ptr(d_sp), ops._dt(g_out), _ptr(sp), B, Lt, Ts, H, _stream()),
              "nbci_coupler_splice_bwd")
        return d_text, d_sp, None, None, None, None


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

    def forward(self, x2d):
        if self.two:
            a, b = self._modules["0"], self._modules["2"]
            return _LinearActFn.apply(_LinearActFn.apply(x2d, a.weight, a.bias, self.act), b.weight, b.bias, 0)
        return _LinearActFn.apply(x2d, self.weight, self.bias, 0)


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
                llm = AutoModelForCausalLM.from_pretrained(pt_path or llm_path)
            if lora is not None and pt_path is None:
                from peft import LoraConfig, get_peft_model
                lc = DictConfig(lora)
                llm = get_peft_model(llm, LoraConfig(inference_mode=False, r=lc.r, lora_alpha=lc.alpha, lora_dropout=lc.dropout,
                                                     target_modules=lc.target_modules, modules_to_save=lc.modules_to_save))
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

    def prepare_embeds(self, input_ids, attention_mask, input_split, spikes, spikes_mask, spikes_timestamp, spikes_lengths,
                       block_idx=None, day_idx=None, targets=None):
        text_embeds = self.llm.get_input_embeddings()(input_ids)                      # stock HF embedding lookup
        batch = dict(spikes=spikes, spikes_mask=spikes_mask, spikes_timestamp=spikes_timestamp, spikes_lengths=spikes_lengths,
                     targets=None, targets_lengths=None, day_idx=day_idx, block_idx=block_idx)
        hidden, tmask = _EncodeFn.apply(self.ndt1, batch, *self.ndt1._param_list)     # (B,T',H), (B,T')
        B, T, H = hidden.shape
        s = self.stacking
        if T % s != 0:                                                                # zero-pad to a multiple of `stacking` (bci.py:130-134)
            new_T = math.ceil(T / s) * s
            hidden = torch.cat((hidden, hidden.new_zeros(B, new_T - T, H)), 1)
            tmask = torch.cat((tmask, tmask.new_zeros(B, new_T - T)), 1)
            T = new_T
        proj = self.projector(hidden.reshape(B * (T // s), H * s)).view(B, T // s, -1)
        valid = (tmask.view(B, T // s, s).sum(-1) == s).to(attention_mask.dtype)      # only features without padding (bci.py:140-141)
        embeds, mask, tg = _SpliceFn.apply(text_embeds, proj, attention_mask, valid, targets, input_split)
        return embeds, mask.to(attention_mask.dtype), tg

    def forward(self, input_ids, attention_mask, input_split, spikes, spikes_mask, spikes_timestamp, spikes_lengths,
                block_idx=None, day_idx=None, targets=None):
        embeds, attention_mask, targets = self.prepare_embeds(input_ids, attention_mask, input_split, spikes, spikes_mask,
                                                              spikes_timestamp, spikes_lengths, block_idx, day_idx, targets)
        logits = self.llm(inputs_embeds=embeds.to(self.llm.dtype), attention_mask=attention_mask, return_dict=True).logits
        loss = n_examples = None
        if targets is not None:                                                       # shifted CE, reduction sum (bci.py:201-212)
            sl = logits[..., :-1, :].contiguous().view(-1, self.llm_config.vocab_size)
            st = targets[..., 1:].contiguous().view(-1).to(sl.device)
            loss = nn.functional.cross_entropy(sl, st, reduction="sum")
            n_examples = (st != -100).sum()
        return BCIOutput(loss=loss, n_examples=n_examples, preds=logits, targets=targets)

    def save_checkpoint(self, save_dir):
        self.llm.save_pretrained(save_dir)
        self.ndt1.save_checkpoint(save_dir)
        torch.save({k: v.detach().clone() for k, v in self.projector.state_dict().items()}, os.path.join(save_dir, "projector.bin"))
        torch.save(dict(self.config.projector), os.path.join(save_dir, "projector_config.pth"))

    def load_checkpoint(self, load_dir):
        self.ndt1.load_checkpoint(load_dir)
        self.projector.load_state_dict(torch.load(os.path.join(load_dir, "projector.bin")))
