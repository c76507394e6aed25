"""NDT1 behind the reference's plugin surface, running on libnbci.so.

Drop-in for `models.ndt1.NDT1` (reference models/ndt1.py:455-692) for method "ctc":
same constructor `(config, **kwargs)`, same forward keyword names, returns `NDT1Output`
(sum-loss, n_examples, preds = (B,T',V) log-probs, targets), same state-dict keys and the
same checkpoint files. Underneath, all parameters are views into ONE flat f32 buffer (plus a
bf16 shadow in bf16 mode) so the C side sees a single pointer, AdamW is one fused launch and
each backward segment's gradients are one contiguous RCCL bucket.
Config options of configs/ndt1.yaml beyond the defaults that run on the HIP path: context spans, RoPE,
`embedder.adapt` (day-specific embed layers), `embedder.day_token / block_token` (learned prefix tokens),
`factors.active` (NeuralFactorsProjection); inputs `day_idx` / `block_idx` as in the reference's forward.

There is no CPU execution path: forward raises NbciUnavailable without the HIP library / a GPU.
"""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import ACT, NBCI_BF16, NBCI_F32, NDT1IO, NDT1Config, check, lib
from .config import DictConfig, ndt1_config, update_config
from .flat import bridge_begin, bridge_check, bridge_stamp
from .model_output import NDT1Output


class _Box(nn.Module):
    """Parameter container; only exists so state_dict() keys match the reference's module tree."""


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _NDT1Function(torch.autograd.Function):
    """Autograd bridge for callers that drive the model with loss.backward() (the reference's
    Trainer via accelerate, trainer.py:336-339). The native train step bypasses autograd."""

    @staticmethod
    def forward(ctx, model, batch, *params):
        loss_vec, preds = model._run_forward(batch, want_grad=True)
        ctx.model, ctx.fwd_id = model, model._fwd_id
        ctx.mark_non_differentiable(preds)
        return loss_vec.sum(), preds

    @staticmethod
    def backward(ctx, g_loss, _g_preds):
        m = ctx.model
        bridge_check(m, ctx.fwd_id, "_NDT1Function")
        grads = torch.zeros_like(m._flat)
        m._run_backward(grads)
        grads.mul_(g_loss.to(grads.dtype))
        out = [None, None]
        for (_, off, numel, shape, _seg) in m._layout:
            out.append(grads[off:off + numel].view(shape))
        return tuple(out)


class NDT1(nn.Module):
    """See module docstring. kwargs: method_name ("ctc"), vocab_size, blank_id, zero_infinity
    (reference ndt1.py:465,489,517); extra: compute_dtype ("bf16" | "fp32", default bf16), residual_dtype ("bf16" | "fp32": storage
    of the residual stream and its gradient stream between kernels; default "fp32" = what the reference's bf16 autocast keeps in f32,
    ndt1.py:325,328; "bf16" is opt-in and needs compute_dtype bf16)."""

    _supports_aux_stream = True   # _run_backward(aux=...): weight gradients / fold on a second stream (NativeTrainer, small batches)

    def __init__(self, config, **kwargs):
        super().__init__()
        config = ndt1_config(config)
        self.method = kwargs["method_name"]
        if self.method not in ("ctc", "endtoend"):   # the reference treats the two alike everywhere (ndt1.py:488,498,516,580)
            raise Exception(f"Method {self.method} not implemented yet for NDT1 on the HIP path "
                            "(only 'ctc' / 'endtoend'; mlm / autoregressive stay on the reference implementation)")
        enc = config["encoder"]
        pt_path = enc.pop("from_pt", None)
        if pt_path is not None:  # warm start (ndt1.py:468-476)
            enc_cfg = torch.load(os.path.join(pt_path, "encoder_config.pth"), weights_only=False)
            config["encoder"] = update_config(config.encoder, enc_cfg)
        enc = DictConfig(config["encoder"])
        for m in enc.masker.values():
            if m.get("active", False):
                raise Exception("active Masker is only meaningful for mlm; not supported on the ctc HIP path")
        emb, tr = enc.embedder, enc.transformer
        if not emb.stack.active:
            raise Exception("HIP path supports embedder.stack.active=true")
        if (emb.day_token or emb.block_token) and tr.use_rope:
            raise Exception("use_rope with day / block tokens: the reference hands T' timestamps to T'+n tokens (ndt1.py:181,441) and fails")
        fac = enc.factors
        if fac.active and float(fac.dropout) != 0.0:
            raise Exception("HIP path supports factors.dropout = 0 (the yaml default) when factors.active")
        if fac.active and fac.act not in ACT:
            raise Exception(f"factors.act {fac.act} is not available on the HIP path")
        if not (emb.bias and tr.attention_bias and tr.mlp_bias):
            raise Exception("HIP path expects bias=true in embedder / attention / mlp")
        dtype_name = kwargs.get("compute_dtype", "bf16")
        self.compute_dtype = {"bf16": NBCI_BF16, "bfloat16": NBCI_BF16, "fp32": NBCI_F32, "float32": NBCI_F32}[dtype_name]
        sn = enc.smooth_and_noise
        c = NDT1Config()
        c.n_channels, c.input_dim = emb.n_channels, emb.input_dim
        c.stack_size, c.stack_stride = emb.stack.size, emb.stack.stride
        c.hidden, c.n_layers, c.n_heads, c.inter = tr.hidden_size, tr.n_layers, tr.n_heads, tr.inter_size
        c.vocab, c.max_F = kwargs["vocab_size"], emb.max_F
        c.smooth_sd = float(sn.smooth_sd) if sn.smooth_sd is not None else 0.0
        c.noise = 1 if sn.noise else 0
        c.white_noise_sd = float(sn.white_noise_sd) if sn.white_noise_sd is not None else 0.0
        c.constant_offset_sd = float(sn.constant_offset_sd) if sn.constant_offset_sd is not None else 0.0
        c.embed_act, c.mlp_act = ACT[emb.act], ACT[tr.act]
        c.embed_dropout, c.dropout = float(emb.dropout), float(tr.dropout)
        c.use_rope, c.rope_theta = (1 if tr.use_rope else 0), float(tr.rope_theta)
        c.context_forward, c.context_backward = enc.context.forward, enc.context.backward
        c.pos = 1 if emb.pos else 0
        c.blank_id, c.zero_infinity = kwargs["blank_id"], 1 if kwargs["zero_infinity"] else 0
        c.dtype = self.compute_dtype
        # storage of the residual stream / its gradient stream between kernels: "fp32" (the default: what bf16 autocast keeps in f32 in the
        # reference, ndt1.py:325,328 - the parity setting, like comm_dtype "fp32") or "bf16" (opt-in, bf16 path only: every kernel still
        # adds / normalises in f32 and rounds once at its store; about twice the logit error, profiles/r03_ab_residual.txt)
        res_name = kwargs.get("residual_dtype", None) or "fp32"
        self.residual_dtype = {"bf16": NBCI_BF16, "bfloat16": NBCI_BF16, "fp32": NBCI_F32, "float32": NBCI_F32}[res_name]
        if self.residual_dtype == NBCI_BF16 and self.compute_dtype != NBCI_BF16:
            raise Exception("residual_dtype 'bf16' needs compute_dtype 'bf16'")
        c.residual_dtype = self.residual_dtype
        # NeuralFactorsProjection (ndt1.py:348-373): encoder output = act(Linear(hidden -> size)) when active
        c.factors_size = int(fac.size) if fac.active else 0
        c.factors_act = ACT[fac.act] if fac.active else 0
        c.factors_bias = 1 if (fac.active and fac.bias) else 0
        # embedder.adapt: one embed_spikes Linear per recording day (ndt1.py:124-129), picked per sample by day_idx
        c.adapt_days = int(emb.n_days) if emb.adapt else 0
        # learned prefix tokens (ndt1.py:151-155,192-201): [day, block, spike tokens...], dropped after out_norm
        c.day_token_days = int(emb.n_days) if emb.day_token else 0
        c.block_token_blocks = int(emb.n_blocks) if emb.block_token else 0
        self._ccfg = c
        self.config = config
        self.vocab_size = kwargs["vocab_size"]
        self._plan = None
        self._layout = self._python_layout()
        self._total = self._layout_total
        # --- parameters: reference init order / RNG consumption (ndt1.py:388-405,494)
        flat = torch.zeros(self._total, dtype=torch.float32)
        self._init_reference_order(flat)
        self._flat = flat
        self._flat_lp = None
        self._bind_parameters()
        if pt_path is not None:
            self.encoder.load_state_dict(torch.load(os.path.join(pt_path, "encoder.bin")))
            self.decoder.load_state_dict(torch.load(os.path.join(pt_path, "decoder.bin")))
        self._ws = None
        self._io_keepalive = None
        self._step_seed = 0
        self.loss_scale = 1.0
        self._rope = None

    # ------------------------------------------------------------------ layout
    def _python_layout(self):
        """Same placement rule as csrc/ndt1.hip build_layout (checked against the plan on first GPU
        use): tensors in canonical order, each aligned to 8 elements, segments aligned too."""
        c = self._ccfg
        H, I, D = c.hidden, c.inter, c.input_dim
        out, cur = [], 0

        def add(name, shape, seg):
            nonlocal cur
            cur = (cur + 7) // 8 * 8
            n = int(math.prod(shape))
            out.append((name, cur, n, tuple(shape), seg))
            cur += n

        if c.adapt_days > 0:
            for d in range(c.adapt_days):
                add(f"encoder.embedder.embed_spikes.{d}.weight", (D, c.n_channels), 0)
                add(f"encoder.embedder.embed_spikes.{d}.bias", (D,), 0)
        else:
            add("encoder.embedder.embed_spikes.weight", (D, c.n_channels), 0)
            add("encoder.embedder.embed_spikes.bias", (D,), 0)
        add("encoder.embedder.stack_projection.weight", (H, D * c.stack_size), 0)
        add("encoder.embedder.stack_projection.bias", (H,), 0)
        if c.pos:
            add("encoder.embedder.embed_pos.weight", (c.max_F, H), 0)
        if c.block_token_blocks > 0:
            add("encoder.embedder.block_embedding.weight", (c.block_token_blocks, H), 0)
        if c.day_token_days > 0:
            add("encoder.embedder.day_embedding.weight", (c.day_token_days, H), 0)
        cur = (cur + 7) // 8 * 8
        self._segments = [(0, cur)]
        # segment 0's backward can run in two parts (nbci_ndt1_io.embed_part): [_embed_split, end) is finished by part 1
        self._embed_split = next(o for (nm, o, _n, _s, _g) in out if nm == "encoder.embedder.stack_projection.weight")
        for l in range(c.n_layers):
            b, pre = cur, f"encoder.layers.{l}."
            add(pre + "ln1.weight", (H,), l + 1); add(pre + "ln1.bias", (H,), l + 1)
            for nm in ("query", "key", "value"):
                add(pre + f"attn.{nm}.weight", (H, H), l + 1)
            for nm in ("query", "key", "value"):
                add(pre + f"attn.{nm}.bias", (H,), l + 1)
            add(pre + "attn.out_proj.weight", (H, H), l + 1); add(pre + "attn.out_proj.bias", (H,), l + 1)
            add(pre + "ln2.weight", (H,), l + 1); add(pre + "ln2.bias", (H,), l + 1)
            add(pre + "mlp.up_proj.weight", (I, H), l + 1); add(pre + "mlp.up_proj.bias", (I,), l + 1)
            add(pre + "mlp.down_proj.weight", (H, I), l + 1); add(pre + "mlp.down_proj.bias", (H,), l + 1)
            cur = (cur + 7) // 8 * 8
            self._segments.append((b, cur))
        b, hs = cur, c.n_layers + 1
        add("encoder.out_norm.weight", (H,), hs); add("encoder.out_norm.bias", (H,), hs)
        if c.factors_size > 0:
            add("encoder.out_proj.proj.0.weight", (c.factors_size, H), hs)
            if c.factors_bias:
                add("encoder.out_proj.proj.0.bias", (c.factors_size,), hs)
        add("decoder.0.weight", (c.vocab, c.factors_size if c.factors_size > 0 else H), hs); add("decoder.0.bias", (c.vocab,), hs)
        cur = (cur + 7) // 8 * 8
        self._segments.append((b, cur))
        self._layout_total = cur
        return out

    def _init_reference_order(self, flat):
        """Draw initial weights exactly as the reference's constructors do, in their order, so
        torch.manual_seed(s) yields the same model (fixup_initialization: ndt1.py:332-344)."""
        c = self._ccfg
        H, I, D, L = c.hidden, c.inter, c.input_dim, c.n_layers
        by_name = {n: (o, k, s) for (n, o, k, s, _) in self._layout}

        def put(name, t):
            o, k, s = by_name[name]
            flat[o:o + k] = t.detach().reshape(-1)

        def linear(prefix, fan_in, fan_out, wscale=None, pre=None):
            m = nn.Linear(fan_in, fan_out)
            w = m.weight
            if pre is not None:
                w = w * pre
            if wscale is not None:
                w = wscale * w
            put(prefix + ".weight", w)
            put(prefix + ".bias", m.bias)

        if c.adapt_days > 0:
            for d in range(c.adapt_days):
                linear(f"encoder.embedder.embed_spikes.{d}", c.n_channels, D)
        else:
            linear("encoder.embedder.embed_spikes", c.n_channels, D)
        linear("encoder.embedder.stack_projection", D * c.stack_size, H)
        if c.pos:
            put("encoder.embedder.embed_pos.weight", nn.Embedding(c.max_F, H).weight)
        if c.block_token_blocks > 0:   # construction order of NeuralEmbeddingLayer.__init__ (ndt1.py:148-155)
            put("encoder.embedder.block_embedding.weight", nn.Embedding(c.block_token_blocks, H).weight)
        if c.day_token_days > 0:
            put("encoder.embedder.day_embedding.weight", nn.Embedding(c.day_token_days, H).weight)
        tr = self.config["encoder"]["transformer"]
        fix = 0.67 * (L ** (-1.0 / 4.0)) if tr["fixup_init"] and L > 0 else None
        vpre = (2 ** 0.5) if fix is not None else None
        for l in range(L):
            pre = f"encoder.layers.{l}."
            put(pre + "ln1.weight", torch.ones(H)); put(pre + "ln1.bias", torch.zeros(H))
            linear(pre + "attn.query", H, H)
            linear(pre + "attn.key", H, H)
            linear(pre + "attn.value", H, H, fix, vpre)
            linear(pre + "attn.out_proj", H, H, fix)
            put(pre + "ln2.weight", torch.ones(H)); put(pre + "ln2.bias", torch.zeros(H))
            linear(pre + "mlp.up_proj", H, I, fix)
            linear(pre + "mlp.down_proj", I, H, fix)
        put("encoder.out_norm.weight", torch.ones(H)); put("encoder.out_norm.bias", torch.zeros(H))
        if c.factors_size > 0:   # NeuralFactorsProjection.__init__ (ndt1.py:360-369): Linear, then the optional re-initialisation
            fac = self.config["encoder"]["factors"]
            m = nn.Linear(H, c.factors_size, bool(fac["bias"]))
            if fac["fixup_init"]:
                m.weight.data.uniform_(-fac["init_range"], fac["init_range"])
                if fac["bias"]:
                    m.bias.data.zero_()
            put("encoder.out_proj.proj.0.weight", m.weight)
            if fac["bias"]:
                put("encoder.out_proj.proj.0.bias", m.bias)
        linear("decoder.0", c.factors_size if c.factors_size > 0 else H, c.vocab)

    def _bind_parameters(self):
        """(Re)create nn.Parameters as views into self._flat under the reference's key names."""
        for top in ("encoder", "decoder"):
            if top in self._modules:
                del self._modules[top]
        self._param_list = []
        for (name, off, numel, shape, _seg) in self._layout:
            parts = name.split(".")
            node = self
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Box())
                node = node._modules[part]
            p = nn.Parameter(self._flat[off:off + numel].view(shape))
            node.register_parameter(parts[-1], p)
            self._param_list.append(p)

    def _apply(self, fn, *a, **k):
        """.to()/.cuda() move every Parameter separately; re-flatten afterwards so the C side keeps
        seeing one buffer."""
        super()._apply(fn, *a, **k)
        first = self._param_list[0]
        flat = torch.zeros(self._total, dtype=torch.float32, device=first.device)
        named = dict(self.named_parameters())
        for (name, off, numel, _shape, _seg) in self._layout:
            flat[off:off + numel] = named[name].detach().reshape(-1).float()
        self._flat = flat
        self._flat_lp = None
        self._ws = None
        self._rope = None
        with torch.no_grad():
            for (name, off, numel, shape, _seg), p in zip(self._layout, self._param_list):
                p.data = flat[off:off + numel].view(shape)
                p.grad = None
        return self

    # ------------------------------------------------------------------ C plan / buffers
    def _ensure_plan(self):
        if self._plan is not None:
            return
        l = lib()
        plan = C.c_void_p()
        check(l.nbci_ndt1_plan_create(C.byref(self._ccfg), C.byref(plan)), "nbci_ndt1_plan_create")
        self._plan = plan
        total = l.nbci_ndt1_param_count(plan)
        n = l.nbci_ndt1_num_params(plan)
        buf = C.create_string_buffer(160)
        off, numel, rows, cols, seg = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        mine = {nm: (o, k, sg) for (nm, o, k, _s, sg) in self._layout}
        if total != self._total or n != len(self._layout):
            raise _lib.NbciError("parameter layout mismatch between llm_bci_amd/ndt1.py and csrc/ndt1.hip")
        for i in range(n):
            check(l.nbci_ndt1_param_info(plan, i, buf, 160, C.byref(off), C.byref(numel), C.byref(rows), C.byref(cols),
                                         C.byref(seg)), "nbci_ndt1_param_info")
            if mine.get(buf.value.decode()) != (off.value, numel.value, seg.value):
                raise _lib.NbciError(f"parameter layout mismatch for {buf.value.decode()}")

    def __del__(self):
        try:
            if getattr(self, "_plan", None) is not None:
                lib().nbci_ndt1_plan_destroy(self._plan)
        except Exception:
            pass

    def refresh_lp(self):
        """(Re)build the bf16 shadow of the flat parameters (after load_state_dict / manual edits;
        the fused AdamW keeps it in sync on its own)."""
        if self.compute_dtype == NBCI_BF16:
            self._flat_lp = self._flat.to(torch.bfloat16)

    def _workspace(self, B, T, S):
        need = lib().nbci_ndt1_workspace_bytes(self._plan, B, T, S)
        if need < 0:
            check(-1, "nbci_ndt1_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._flat.device)
        return self._ws, need

    def tokens(self, T):
        self._ensure_plan()
        return lib().nbci_ndt1_tokens(self._plan, T)

    # ------------------------------------------------------------------ forward / backward
    def _run_forward(self, batch, want_grad, seed=None, grad_scale=1.0, hidden_out=None, token_mask_out=None):
        spikes = batch["spikes"]
        if not spikes.is_cuda:
            raise _lib.NbciUnavailable("NDT1 (HIP path) needs tensors on a ROCm device; there is no CPU fallback")
        self._ensure_plan()
        bridge_stamp(self)
        if self.compute_dtype == NBCI_BF16 and self._flat_lp is None:
            self.refresh_lp()
        dev = spikes.device
        B, T, N = spikes.shape
        if N != self._ccfg.n_channels:
            raise ValueError(f"expected {self._ccfg.n_channels} channels, got {N}")
        spikes = spikes.contiguous().float()
        mask = batch["spikes_mask"].contiguous().long()
        ts = batch["spikes_timestamp"].contiguous().long()
        lens = batch["spikes_lengths"].reshape(-1).contiguous().long()
        tg = batch.get("targets")
        tl = batch.get("targets_lengths")
        S = 0
        if tg is not None:
            tg = tg.contiguous().long()
            tl = tl.reshape(-1).contiguous().long()
            S = tg.shape[1]
        Tp = lib().nbci_ndt1_tokens(self._plan, T)
        if Tp <= 0:
            raise ValueError("sequence shorter than the stacking window")
        ws, need = self._workspace(B, T, S)
        preds = torch.empty(B, Tp, self.vocab_size, dtype=torch.float32, device=dev)
        loss = torch.zeros(B, dtype=torch.float32, device=dev)
        argmax = torch.empty(B, Tp, dtype=torch.int32, device=dev)
        io = NDT1IO()
        io.B, io.T, io.S = B, T, S
        io.spikes, io.spikes_mask, io.spikes_timestamp, io.spikes_lengths = _ptr(spikes), _ptr(mask), _ptr(ts), _ptr(lens)
        io.targets, io.targets_lengths = _ptr(tg), _ptr(tl)
        if self._ccfg.use_rope:
            if self._rope is None:
                hd = self._ccfg.hidden // self._ccfg.n_heads
                inv = 1.0 / (self._ccfg.rope_theta ** (torch.arange(0, hd, 2, device=dev).float() / hd))
                fr = torch.einsum("i,j->ij", torch.arange(self._ccfg.max_F, device=dev).float(), inv)
                emb = torch.cat((fr, fr), -1)
                self._rope = (emb.cos().contiguous(), emb.sin().contiguous())
            io.rope_cos, io.rope_sin = _ptr(self._rope[0]), _ptr(self._rope[1])
        io.train = 1 if self.training else 0
        io.want_grad = 1 if (want_grad and (tg is not None or hidden_out is not None)) else 0
        if seed is None:
            self._step_seed = (self._step_seed * 1664525 + 1013904223) & 0xFFFFFFFF
            seed = self._step_seed
        io.seed = seed
        io.grad_scale = grad_scale
        io.preds, io.loss, io.argmax = _ptr(preds), _ptr(loss), _ptr(argmax)
        io.hidden_out = _ptr(hidden_out)
        io.token_mask_out = _ptr(token_mask_out)
        io.d_hidden = None
        io.workspace, io.workspace_bytes = _ptr(ws), need
        def _index(name, need, why):
            t = batch.get(name)
            if not need:
                return None
            if t is None:
                raise ValueError(f"{why}: forward needs {name} (one entry per sample)")
            t = t.reshape(-1).contiguous().long()
            if t.numel() != B:
                raise ValueError(f"{name} has {t.numel()} entries for a batch of {B}")
            return t
        days = _index("day_idx", self._ccfg.adapt_days > 0 or self._ccfg.day_token_days > 0, "embedder.adapt / day_token is on")
        blocks = _index("block_idx", self._ccfg.block_token_blocks > 0, "embedder.block_token is on")
        io.day_idx, io.block_idx = _ptr(days), _ptr(blocks)
        check(lib().nbci_ndt1_forward(self._plan, _ptr(self._flat), _ptr(self._flat_lp), C.byref(io), _stream()),
              "nbci_ndt1_forward")
        # keep every borrowed tensor alive until the backward of this step has been queued
        self.last_rows = B * Tp
        self._io_keepalive = (io, spikes, mask, ts, lens, tg, tl, ws, preds, loss, argmax, hidden_out, token_mask_out, days, blocks)
        self.last_argmax = argmax
        return loss, preds

    def _run_backward(self, grads, seg_hi=None, seg_lo=0, d_hidden=None, embed_part=0, aux=None):
        """aux: an optional second torch.cuda.Stream (nbci_ndt1_io.aux_stream): weight gradients + the fold of the small-vector
        gradients are queued there and the covered segments' gradients are complete on THAT stream."""
        io = self._io_keepalive[0]
        io.aux_stream = C.c_void_p(aux.cuda_stream) if aux is not None else None
        io.d_hidden = _ptr(d_hidden)   # f32 (B,T',H): backward starts from the encoder output instead of the CTC head
        io.embed_part = embed_part     # segment 0 only: 1 = stack-projection/position/token gradients, 2 = the rest, 0 = both
        if not io.want_grad:
            raise RuntimeError("backward called but the forward pass ran without want_grad/targets")
        if seg_hi is None:
            seg_hi = self._ccfg.n_layers + 1
        check(lib().nbci_ndt1_backward(self._plan, _ptr(self._flat), _ptr(self._flat_lp), C.byref(io), _ptr(grads),
                                       seg_hi, seg_lo, _stream()), "nbci_ndt1_backward")

    def forward(self, spikes, spikes_mask, spikes_timestamp, spikes_lengths, targets=None, targets_lengths=None,
                block_idx=None, day_idx=None):
        batch = dict(spikes=spikes, spikes_mask=spikes_mask, spikes_timestamp=spikes_timestamp,
                     spikes_lengths=spikes_lengths, targets=targets, targets_lengths=targets_lengths, day_idx=day_idx, block_idx=block_idx)
        bridge_begin(self)   # an external optimizer may have stepped the f32 views since the bf16 shadow was taken
        if torch.is_grad_enabled() and targets is not None and any(p.requires_grad for p in self._param_list):
            loss, preds = _NDT1Function.apply(self, batch, *self._param_list)
        else:
            loss_vec, preds = self._run_forward(batch, want_grad=False)
            loss = loss_vec.sum() if targets is not None else None
        n_examples = torch.tensor(spikes.size(0), device=spikes.device, dtype=torch.int64)
        return NDT1Output(loss=loss, n_examples=n_examples, preds=preds, targets=targets)

    # ------------------------------------------------------------------ checkpoints (ndt1.py:685-692)
    def save_checkpoint(self, save_dir):
        enc = {k: v.detach().clone() for k, v in self.encoder.state_dict().items()}
        dec = {k: v.detach().clone() for k, v in self.decoder.state_dict().items()}
        torch.save(enc, os.path.join(save_dir, "encoder.bin"))
        torch.save(dict(self.config.encoder), os.path.join(save_dir, "encoder_config.pth"))
        torch.save(dec, os.path.join(save_dir, "decoder.bin"))

    def load_checkpoint(self, load_dir):
        self.encoder.load_state_dict(torch.load(os.path.join(load_dir, "encoder.bin")))
        self.decoder.load_state_dict(torch.load(os.path.join(load_dir, "decoder.bin")))
        self.refresh_lp()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._flat_lp = None
        return out
