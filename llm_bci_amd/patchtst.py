"""PatchTST behind the reference's plugin surface, running on libnbci.so.

Drop-in for `models.patchtst.PatchTSTForSpikingActivity` (reference models/patchtst.py:160-266; registry key "PatchTST")
for methods "ctc" (PredictHead, mean pooling) and "mlm" (PretrainHead): same constructor `(config, **kwargs)`, same forward
keyword names, returns `PatchTSTOutput`, same state-dict keys — including the encoder keys of the HF `PatchTSTModel` the
reference wraps (`encoder.encoder.layers.N.self_attn.q_proj.weight`, `...norm_sublayer1.batchnorm.running_mean`,
`encoder.encoder.positional_encoder.position_enc`, ...) — and the same checkpoint files.

Supported configuration = configs/patchtst.yaml's: share_embedding, share_projection, channel_attention false, norm_type
batchnorm, pre_norm, sincos positions, scaling null, mask_type random, pooling mean, head_dropout 0. Anything else raises.
Random patch masks come from the library's counter RNG (reproducible from the step seed). No CPU path.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import ACT, LOSS_KIND, NBCI_BF16, NBCI_F32, PtstConfig, PtstIO, check, lib
from .config import DictConfig, patchtst_config, update_config
from .flat import FlatParamModule, LayoutBuilder, bridge_begin, bridge_check, bridge_stamp, _Box, _ptr, _stream
from .model_output import ModelOutput
from .patchtst_init import reference_order_init


@dataclass
class PatchTSTOutput(ModelOutput):
    mask: Optional[torch.Tensor] = None
    preds: Optional[torch.Tensor] = None
    targets: Optional[torch.Tensor] = None
    patch_input: Optional[torch.Tensor] = None


def _saved_config(path):
    """save_checkpoint writes the config dicts with torch.save under a .yaml name (patchtst.py:259,261); accept that and real yaml."""
    try:
        return dict(torch.load(path, weights_only=False))
    except Exception:
        return path


def layout_of(D, F, pl, L, nout, mlp_decoder):
    b = LayoutBuilder()
    b.add("encoder.encoder.embedder.input_embedding.weight", (D, pl), 0); b.add("encoder.encoder.embedder.input_embedding.bias", (D,), 0)
    b.end_segment()
    for l in range(L):
        pre = f"encoder.encoder.layers.{l}."
        b.add(pre + "norm_sublayer1.batchnorm.weight", (D,), l + 1); b.add(pre + "norm_sublayer1.batchnorm.bias", (D,), l + 1)
        for nm in ("q_proj", "k_proj", "v_proj"):
            b.add(pre + f"self_attn.{nm}.weight", (D, D), l + 1)
        for nm in ("q_proj", "k_proj", "v_proj"):
            b.add(pre + f"self_attn.{nm}.bias", (D,), l + 1)
        b.add(pre + "self_attn.out_proj.weight", (D, D), l + 1); b.add(pre + "self_attn.out_proj.bias", (D,), l + 1)
        b.add(pre + "norm_sublayer3.batchnorm.weight", (D,), l + 1); b.add(pre + "norm_sublayer3.batchnorm.bias", (D,), l + 1)
        b.add(pre + "ff.0.weight", (F, D), l + 1); b.add(pre + "ff.0.bias", (F,), l + 1)
        b.add(pre + "ff.3.weight", (D, F), l + 1); b.add(pre + "ff.3.bias", (D,), l + 1)
        b.end_segment()
    hs = L + 1
    if mlp_decoder:
        b.add("decoder.projection.0.weight", (D, D), hs); b.add("decoder.projection.0.bias", (D,), hs)
        b.add("decoder.projection.2.weight", (nout, D), hs); b.add("decoder.projection.2.bias", (nout,), hs)
    else:
        b.add("decoder.projection.weight", (nout, D), hs); b.add("decoder.projection.bias", (nout,), hs)
    b.end_segment()
    return b


class _PtstFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, batch, *params):
        loss, preds = model._run_forward(batch, want_grad=True)
        ctx.model, ctx.fwd_id = model, model._fwd_id
        ctx.mark_non_differentiable(preds)
        return loss.sum(), preds

    @staticmethod
    def backward(ctx, g_loss, _g_preds):
        m = ctx.model
        bridge_check(m, ctx.fwd_id, "_PtstFunction")
        grads = torch.zeros_like(m._flat)
        m._run_backward(grads)
        grads.mul_(g_loss.to(grads.dtype))
        out = [None, None]
        for (_, off, numel, shape, _seg) in m._layout:
            out.append(grads[off:off + numel].view(shape))
        return tuple(out)


class PatchTSTForSpikingActivity(FlatParamModule):
    """kwargs: method_name ("ctc": vocab_size, blank_id, zero_infinity | "mlm": loss, log_input) (patchtst.py:190-208);
    extra: compute_dtype ("bf16" | "fp32" | "fp8", default bf16). "fp8" = the bf16 path with the q / k / v projections of the forward
    pass on the block-scaled fp8 matrix instruction (MX e4m3 activations quantised by the BatchNorm pass; d_model % 128 == 0)."""

    def __init__(self, config, **kwargs):
        super().__init__()
        config = patchtst_config(config)
        self.method = kwargs["method_name"]
        enc_pt = config["encoder"].pop("from_pt", None)
        if enc_pt is not None:   # patchtst.py:171-174
            config["encoder"] = update_config(config.encoder, _saved_config(os.path.join(enc_pt, "encoder_config.yaml")))
        dec_pt = config["decoder"].pop("from_pt", None)
        if dec_pt is not None:
            config["decoder"] = update_config(config.decoder, _saved_config(os.path.join(dec_pt, "decoder_config.yaml")))
        enc, dec = DictConfig(config["encoder"]), DictConfig(config["decoder"])
        if self.method not in ("ctc", "mlm"):
            raise Exception(f"Method {self.method} not implemented yet for PatchTST")   # patchtst.py:206
        bad = []
        if not enc.get("share_embedding", True): bad.append("share_embedding: false")
        if enc.get("channel_attention", False): bad.append("channel_attention: true")
        if enc.get("norm_type", "batchnorm") != "batchnorm": bad.append("norm_type != batchnorm")
        if not enc.get("pre_norm", True): bad.append("pre_norm: false")
        if enc.get("positional_encoding_type", "sincos") != "sincos": bad.append("positional_encoding_type != sincos")
        if enc.get("scaling", None) not in (None, False): bad.append("scaling")
        if enc.get("use_cls_token", False): bad.append("use_cls_token")
        if enc.get("mask_type", "random") != "random" and enc.get("do_mask_input", False): bad.append("mask_type != random")
        if enc.get("unmasked_channel_indices", None) is not None: bad.append("unmasked_channel_indices")
        if not enc.get("bias", True): bad.append("bias: false")
        if not dec.get("share_projection", True): bad.append("share_projection: false")
        if dec.get("pooling_type", "mean") != "mean": bad.append("pooling_type != mean")
        if float(dec.get("head_dropout", 0.0)) > 0: bad.append("head_dropout > 0")
        if bad:
            raise Exception("PatchTST HIP path does not support: " + ", ".join(bad))
        if self.method == "mlm":
            assert enc.do_mask_input, "Can't pretrain with inactive masking"   # patchtst.py:193
            self.loss_name, self.log_input = kwargs["loss"], bool(kwargs["log_input"])
            if self.loss_name not in ("poisson_nll", "mse"):
                raise Exception(f"Loss {self.loss_name} not implemented yet for mlm")
        dtype_name = kwargs.get("compute_dtype", "bf16")
        self.compute_dtype = {"bf16": NBCI_BF16, "bfloat16": NBCI_BF16, "fp32": NBCI_F32, "float32": NBCI_F32, "fp8": NBCI_BF16}[dtype_name]
        self.fp8_qkv = dtype_name == "fp8"
        if self.fp8_qkv and (enc.d_model % 128 != 0 or enc.d_model > 512):   # (nbci_gemm_fp8: K = d_model in {128, 256, 384, 512})
            raise Exception(f"compute_dtype fp8 needs d_model in (128, 256, 384, 512), got {enc.d_model}")
        c = PtstConfig()
        c.num_input_channels, c.context_length = enc.num_input_channels, enc.context_length
        c.patch_length, c.patch_stride = enc.patch_length, enc.patch_stride
        c.num_hidden_layers, c.d_model, c.num_attention_heads, c.ffn_dim = enc.num_hidden_layers, enc.d_model, enc.num_attention_heads, enc.ffn_dim
        c.norm_eps = float(enc.get("norm_eps", 1e-5))
        c.attention_dropout, c.positional_dropout = float(enc.get("attention_dropout", 0.0)), float(enc.get("positional_dropout", 0.0))
        c.path_dropout, c.ff_dropout = float(enc.get("path_dropout", 0.0)), float(enc.get("ff_dropout", 0.0))
        c.act = ACT[enc.get("activation_function", "gelu")]
        c.do_mask_input = 1 if enc.get("do_mask_input", False) else 0
        c.random_mask_ratio = float(enc.get("random_mask_ratio", 0.5))
        c.channel_consistent_masking = 1 if enc.get("channel_consistent_masking", False) else 0
        c.mask_value = float(enc.get("mask_value", 0))
        c.method = 0 if self.method == "ctc" else 1
        c.vocab = kwargs.get("vocab_size", 0) if self.method == "ctc" else 0
        c.blank_id = kwargs.get("blank_id", 0) if self.method == "ctc" else 0
        c.zero_infinity = 1 if kwargs.get("zero_infinity", False) else 0
        c.mlp_decoder, c.dec_act = (1 if dec.get("mlp_decoder", False) else 0), ACT[dec.get("mlp_activation", "gelu")]
        c.loss = LOSS_KIND[(self.loss_name, self.log_input)] if self.method == "mlm" else 0
        c.dtype = self.compute_dtype
        c.fp8_qkv = 1 if self.fp8_qkv else 0
        # storage of the residual stream / its gradient stream between kernels (as NDT1's residual_dtype): "fp32" by default (parity: the
        # reference keeps these in f32 under autocast), "bf16" opt-in on the bf16 / fp8 paths
        res_name = kwargs.get("residual_dtype", None) or "fp32"
        self.residual_dtype = {"bf16": NBCI_BF16, "bfloat16": NBCI_BF16, "fp32": NBCI_F32, "float32": NBCI_F32}[res_name]
        if self.residual_dtype == NBCI_BF16 and self.compute_dtype != NBCI_BF16:
            raise Exception("residual_dtype 'bf16' needs compute_dtype 'bf16' or 'fp8'")
        c.residual_dtype = self.residual_dtype
        self._ccfg = c
        self.config = config
        T, pl, st = c.context_length, c.patch_length, c.patch_stride
        if T <= pl:
            raise ValueError(f"Sequence length ({T}) has to be greater than the patch length ({pl})")
        self.num_patches = (max(T, pl) - pl) // st + 1
        P, D, L = self.num_patches, c.d_model, c.num_hidden_layers
        nout = c.vocab if self.method == "ctc" else pl
        b = layout_of(D, c.ffn_dim, pl, L, nout, c.mlp_decoder)
        self._layout, self._segments, self._total = b.entries, b.segments, b.cur
        # aux buffer: position_enc (P,D) then per layer norm1 mean/var, norm3 mean/var; nbt: 2 counters per layer
        self._aux_layout = [("encoder.encoder.positional_encoder.position_enc", 0, P * D, (P, D))]
        off = P * D
        self._nbt_names = []
        for l in range(L):
            for nm in ("norm_sublayer1", "norm_sublayer3"):
                pre = f"encoder.encoder.layers.{l}.{nm}.batchnorm."
                self._aux_layout.append((pre + "running_mean", off, D, (D,))); off += D
                self._aux_layout.append((pre + "running_var", off, D, (D,))); off += D
                self._nbt_names.append(pre + "num_batches_tracked")
        self._aux_total = off
        init = reference_order_init(enc, dec, self.method, kwargs.get("vocab_size"))
        flat = torch.zeros(self._total, dtype=torch.float32)
        for (name, o, n, _s, _g) in self._layout:
            flat[o:o + n] = init[name].reshape(-1)
        aux = torch.zeros(self._aux_total, dtype=torch.float32)
        for (name, o, n, _s) in self._aux_layout:
            aux[o:o + n] = init[name].reshape(-1).float()
        self._aux = aux
        self._nbt = torch.zeros(max(1, 2 * L), dtype=torch.int64)
        self._plan = None
        self._adopt(flat)
        if enc_pt is not None:
            self.encoder.load_state_dict(torch.load(os.path.join(enc_pt, "encoder.bin")))
        if dec_pt is not None:
            self.decoder.load_state_dict(torch.load(os.path.join(dec_pt, "decoder.bin")))
        self._io_keepalive = None
        self._step_seed = 0
        self.mask_override = None      # (B,C,P) bool: replaces the random patch mask (tests / replay)
        self.last_n_examples = None

    # ------------------------------------------------------------------ parameters + buffers as views
    def _bind_parameters(self):
        super()._bind_parameters()
        self._bind_aux()

    def _node(self, name):
        parts = name.split(".")
        node = self
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, _Box())
            node = node._modules[part]
        return node, parts[-1]

    def _bind_aux(self):
        for i, (name, off, numel, shape) in enumerate(self._aux_layout):
            node, leaf = self._node(name)
            view = self._aux[off:off + numel].view(shape)
            if i == 0:   # position_enc: nn.Parameter(requires_grad=False) in transformers
                node._parameters.pop(leaf, None)
                node.register_parameter(leaf, nn.Parameter(view, requires_grad=False))
            else:
                node._buffers.pop(leaf, None)
                node.register_buffer(leaf, view)
        for i, name in enumerate(self._nbt_names):
            node, leaf = self._node(name)
            node._buffers.pop(leaf, None)
            node.register_buffer(leaf, self._nbt[i:i + 1].view(()))

    def _apply(self, fn, *a, **k):
        nn.Module._apply(self, fn, *a, **k)
        dev = self._param_list[0].device
        named = dict(self.named_parameters())
        flat = torch.zeros(self._total, dtype=torch.float32, device=dev)
        for (name, off, numel, _shape, _seg) in self._layout:
            flat[off:off + numel] = named[name].detach().reshape(-1).float()
        bufs = dict(self.named_buffers())
        aux = torch.zeros(self._aux_total, dtype=torch.float32, device=dev)
        for i, (name, off, numel, _shape) in enumerate(self._aux_layout):
            src = named[name] if i == 0 else bufs[name]
            aux[off:off + numel] = src.detach().reshape(-1).float()
        nbt = torch.zeros(max(1, len(self._nbt_names)), dtype=torch.int64, device=dev)
        for i, name in enumerate(self._nbt_names):
            nbt[i] = bufs[name].detach().reshape(()).long()
        self._flat, self._aux, self._nbt = flat, aux, nbt
        self._flat_lp = None
        self._ws = None
        with torch.no_grad():
            for (name, off, numel, shape, _seg), p in zip(self._layout, self._param_list):
                p.data = flat[off:off + numel].view(shape)
                p.grad = None
        self._bind_aux()
        return self

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)   # copies into the views in place
        self._flat_lp = None
        return out

    # ------------------------------------------------------------------ plan / buffers
    def _ensure_plan(self):
        if self._plan is not None:
            return
        plan = C.c_void_p()
        check(lib().nbci_ptst_plan_create(C.byref(self._ccfg), C.byref(plan)), "nbci_ptst_plan_create")
        self._plan = plan
        self._check_layout("nbci_ptst_", plan)
        if lib().nbci_ptst_aux_floats(plan) != self._aux_total or lib().nbci_ptst_num_patches(plan) != self.num_patches:
            raise _lib.NbciError("aux layout mismatch between llm_bci_amd/patchtst.py and csrc/patchtst.hip")

    def __del__(self):
        try:
            if getattr(self, "_plan", None) is not None:
                lib().nbci_ptst_plan_destroy(self._plan)
        except Exception:
            pass

    def _workspace(self, B, S):
        need = lib().nbci_ptst_workspace_bytes(self._plan, B, S)
        if need < 0:
            check(-1, "nbci_ptst_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._flat.device)
        return self._ws, need

    # ------------------------------------------------------------------ forward / backward
    def _run_forward(self, batch, want_grad, seed=None, grad_scale=1.0, hidden_out=None):
        spikes = batch["spikes"]
        if not spikes.is_cuda:
            raise _lib.NbciUnavailable("PatchTST (HIP path) needs tensors on a ROCm device; there is no CPU fallback")
        self._ensure_plan()
        bridge_stamp(self)
        if self.compute_dtype == NBCI_BF16 and self._flat_lp is None:
            self.refresh_lp()
        c = self._ccfg
        dev = spikes.device
        B, T, Cn = spikes.shape
        if T != c.context_length:
            raise ValueError(f"Input sequence length ({T}) doesn't match model configuration ({c.context_length}).")
        if Cn != c.num_input_channels:
            raise ValueError(f"The defined number of input channels ({c.num_input_channels}) in the config has to be the same as the "
                             f"number of channels in the batch input ({Cn})")
        P, pl = self.num_patches, c.patch_length
        spikes = spikes.contiguous().float()
        smask = batch["spikes_mask"].contiguous().long()
        lens = batch.get("spikes_lengths")
        tg, tl, S = batch.get("targets"), batch.get("targets_lengths"), 0
        ctc = self.method == "ctc"
        if ctc:
            lens = lens.reshape(-1).contiguous().long()
            if tg is not None:
                tg = tg.contiguous().long(); tl = tl.reshape(-1).contiguous().long(); S = tg.shape[1]
        ext = None
        if self.mask_override is not None and c.do_mask_input:
            ext = self.mask_override.to(dev).to(torch.uint8).contiguous()
        if seed is None:
            self._step_seed = (self._step_seed * 1664525 + 1013904223) & 0xFFFFFFFF
            seed = self._step_seed
        ws, need = self._workspace(B, S)
        io = PtstIO()
        io.B, io.S = B, S
        io.spikes, io.spikes_mask, io.spikes_lengths = _ptr(spikes), _ptr(smask), _ptr(lens) if ctc else None
        io.targets, io.targets_lengths = (_ptr(tg), _ptr(tl)) if ctc else (None, None)
        io.ext_mask = _ptr(ext)
        io.train, io.want_grad = (1 if self.training else 0), (1 if want_grad else 0)
        io.seed, io.grad_scale = seed, grad_scale
        io.aux, io.nbt = _ptr(self._aux), _ptr(self._nbt)
        nex = torch.zeros(1, dtype=torch.int64, device=dev)
        patch = mask_out = argmax = None
        if ctc:
            preds = torch.empty(B, P, c.vocab, dtype=torch.float32, device=dev)
            loss = torch.zeros(B, dtype=torch.float32, device=dev)
            argmax = torch.empty(B, P, dtype=torch.int32, device=dev)
            nex.fill_(B)
        else:
            preds = torch.empty(B, Cn, P, pl, dtype=torch.float32, device=dev)
            patch = torch.empty(B, Cn, P, pl, dtype=torch.float32, device=dev)
            mask_out = torch.empty(B, Cn, P, dtype=torch.uint8, device=dev)
            loss = torch.zeros(1, dtype=torch.float32, device=dev)
        io.preds, io.patch_input, io.mask_out, io.loss, io.n_examples, io.argmax = (_ptr(preds), _ptr(patch), _ptr(mask_out), _ptr(loss),
                                                                                    _ptr(nex), _ptr(argmax))
        io.hidden_out = _ptr(hidden_out)
        io.workspace, io.workspace_bytes = _ptr(ws), need
        check(lib().nbci_ptst_forward(self._plan, _ptr(self._flat), _ptr(self._flat_lp), C.byref(io), _stream()), "nbci_ptst_forward")
        # layout of the first 7 slots is shared with NDT1 (NativeTrainer._per reads targets / lengths at [5], [6])
        self._io_keepalive = (io, spikes, smask, lens, ext, tg, tl, ws, preds, loss, nex, patch, mask_out, argmax, hidden_out)
        if ctc:
            self.last_argmax = argmax
            self.last_n_examples = None
        else:
            self.last_n_examples = nex
        self.last_mask = mask_out.bool() if mask_out is not None else None
        self.last_patch_input = patch
        return loss, preds

    def _run_backward(self, grads, seg_hi=None, seg_lo=0):
        io = self._io_keepalive[0]
        if not io.want_grad:
            raise RuntimeError("backward called but the forward pass ran without want_grad")
        if seg_hi is None:
            seg_hi = self._ccfg.num_hidden_layers + 1
        check(lib().nbci_ptst_backward(self._plan, _ptr(self._flat), _ptr(self._flat_lp), C.byref(io), _ptr(grads), seg_hi, seg_lo,
                                       _stream()), "nbci_ptst_backward")

    def forward(self, spikes, spikes_mask, spikes_lengths=None, targets=None, targets_lengths=None):
        batch = dict(spikes=spikes, spikes_mask=spikes_mask, spikes_lengths=spikes_lengths, targets=targets, targets_lengths=targets_lengths)
        has_loss = self.method == "mlm" or targets is not None
        bridge_begin(self)   # an external optimizer may have stepped the f32 views since the bf16 shadow was taken
        if torch.is_grad_enabled() and has_loss and any(p.requires_grad for p in self._param_list):
            loss, preds = _PtstFunction.apply(self, batch, *self._param_list)
        else:
            loss_vec, preds = self._run_forward(batch, want_grad=False)
            loss = loss_vec.sum() if has_loss else None
        if self.method == "mlm":
            return PatchTSTOutput(loss=loss, n_examples=self.last_n_examples.reshape(()), mask=self.last_mask, preds=preds,
                                  targets=self.last_patch_input, patch_input=self.last_patch_input)
        n = torch.tensor(spikes.size(0), device=spikes.device, dtype=torch.long)   # patchtst.py:241 (len(targets))
        return PatchTSTOutput(loss=loss, n_examples=n, preds=preds, targets=targets)

    # ------------------------------------------------------------------ checkpoints (patchtst.py:257-266)
    def save_checkpoint(self, save_dir):
        torch.save({k: v.detach().clone() for k, v in self.encoder.state_dict().items()}, os.path.join(save_dir, "encoder.bin"))
        torch.save(dict(self.config.encoder), os.path.join(save_dir, "encoder_config.yaml"))
        torch.save({k: v.detach().clone() for k, v in self.decoder.state_dict().items()}, os.path.join(save_dir, "decoder.bin"))
        torch.save(dict(self.config.decoder), os.path.join(save_dir, "decoder_config.yaml"))

    def load_checkpoint(self, load_dir):
        self.encoder.load_state_dict(torch.load(os.path.join(load_dir, "encoder.bin")))
        self.decoder.load_state_dict(torch.load(os.path.join(load_dir, "decoder.bin")))
        self.refresh_lp()
