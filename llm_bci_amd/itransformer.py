"""iTransformer (SSL / mlm) behind the reference's plugin surface, running on libnbci.so.

Drop-in for `models.itransformer.iTransformer` (reference models/itransformer.py:212-375) for method "mlm" with either embedder
(`mlp`, :108-118, or `transformer` = UnivariateTransformer, :40-93,119-124) and the channel / region / depth embeddings
(:126-150,189-202): same constructor `(config, **kwargs)` (kwargs: method_name, loss, log_input), same forward keyword
names, returns `iTransformerOutput(loss = masked sum, n_examples = #masked bins, mask, preds (B,T,N), targets)`, same
state-dict keys (`encoder.embed.0.0.weight`, `encoder.transformer.layers.N.self_attn.in_proj_weight`, ...,
`decoder.2.bias`) and checkpoint files. The maskers (models/masker.py) run on the device through `nbci_masker`.

Differences, on purpose: the reference's Masker mutates the caller's `spikes` tensor in place (masker.py:96,102);
this module leaves it untouched and works on a private copy. Random draws come from the library's counter RNG
(reproducible from the step seed), not torch's generator. No CPU path: raises NbciUnavailable without a GPU.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import ACT, LOSS_KIND, MASK_MODE, NBCI_BF16, NBCI_F32, ItrConfig, ItrIO, MaskerDesc, check, lib
from .config import DictConfig, itransformer_config, update_config
from .flat import FlatParamModule, LayoutBuilder, bridge_begin, bridge_check, bridge_stamp, _ptr, _stream
from .model_output import ModelOutput

SITE_MASKER = 64   # + 8 * masker index: +0 mask, +1 zero, +2 random-select, +3 random values, +4 timespan


@dataclass
class iTransformerOutput(ModelOutput):
    mask: Optional[torch.Tensor] = None
    preds: Optional[torch.Tensor] = None
    targets: Optional[torch.Tensor] = None


def _mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def _rng_u32(seed, site, idx):
    """host copy of csrc/nbci_common.h rng_u32 (for the one host-side draw: the temporal masker's timespan)."""
    h = _mix32(idx ^ ((seed * 0x9E3779B9 + 0x85EBCA6B) & 0xFFFFFFFF))
    return _mix32(h ^ ((site * 0xC2B2AE35 + 0x27D4EB2F) & 0xFFFFFFFF))


def masker_timespan(seed, site, expand_prob, max_timespan):
    """masker.py:55-59: with probability expand_prob the temporal mask is widened to randint(1, max_timespan)."""
    u = np.float32(_rng_u32(seed, site + 4, 0) >> 8) * np.float32(1.0 / 16777216.0)
    if u < np.float32(expand_prob):
        return 1 + _rng_u32(seed, site + 4, 1) % int(max_timespan)
    return 1


def region_sample(seed, site, regions, n):
    """`random.sample(regions, n)` of "masker copy.py":91,99 with the counter RNG (site + 5): a partial Fisher-Yates shuffle, so the
    choice is a function of (seed, site) like every other draw (the reference uses Python's global `random`)."""
    pool = list(regions)
    n = int(n)
    if n > len(pool):
        raise ValueError("Sample larger than population or is negative")   # what random.sample raises
    for j in range(n):
        k = j + _rng_u32(seed, site + 5, j) % (len(pool) - j)
        pool[j], pool[k] = pool[k], pool[j]
    return pool[:n]


def _layer_entries(b, pre, H, seg):
    """one torch.nn.TransformerEncoderLayer in state-dict order"""
    b.add(pre + "self_attn.in_proj_weight", (3 * H, H), seg); b.add(pre + "self_attn.in_proj_bias", (3 * H,), seg)
    b.add(pre + "self_attn.out_proj.weight", (H, H), seg); b.add(pre + "self_attn.out_proj.bias", (H,), seg)
    b.add(pre + "linear1.weight", (4 * H, H), seg); b.add(pre + "linear1.bias", (4 * H,), seg)
    b.add(pre + "linear2.weight", (H, 4 * H), seg); b.add(pre + "linear2.bias", (H,), seg)
    b.add(pre + "norm1.weight", (H,), seg); b.add(pre + "norm1.bias", (H,), seg)
    b.add(pre + "norm2.weight", (H,), seg); b.add(pre + "norm2.bias", (H,), seg)


def layout_of(T, H, L, C_, R, use_cls, mlp_decoder, embed_depth=False, embedder=None):
    """embedder: None = `mlp`; dict(h, nh, L) = the UnivariateTransformer embedder (itransformer.py:40-93) + embed_proj (:119-124)"""
    b = LayoutBuilder()
    if embedder is None:
        b.add("encoder.embed.0.0.weight", (H, T), 0); b.add("encoder.embed.0.0.bias", (H,), 0)
        b.add("encoder.embed.0.3.weight", (H, H), 0); b.add("encoder.embed.0.3.bias", (H,), 0)
        b.add("encoder.embed.1.weight", (H,), 0); b.add("encoder.embed.1.bias", (H,), 0)
    else:
        h = embedder["h"]
        b.add("encoder.embed.embed_spikes.0.weight", (h, 1), 0); b.add("encoder.embed.embed_spikes.0.bias", (h,), 0)
        b.add("encoder.embed.embed_spikes.2.weight", (h, h), 0); b.add("encoder.embed.embed_spikes.2.bias", (h,), 0)
        b.add("encoder.embed.embed_pos.weight", (T, h), 0)
        b.add("encoder.embed.cls_embed.weight", (1, h), 0)
        for l in range(embedder["L"]):
            _layer_entries(b, f"encoder.embed.transformer.layers.{l}.", h, 0)
        b.add("encoder.embed.transformer.norm.weight", (h,), 0); b.add("encoder.embed.transformer.norm.bias", (h,), 0)
        b.add("encoder.embed_proj.0.weight", (H, h), 0); b.add("encoder.embed_proj.0.bias", (H,), 0)
        b.add("encoder.embed_proj.1.weight", (H,), 0); b.add("encoder.embed_proj.1.bias", (H,), 0)
    if C_:
        b.add("encoder.channel_embeddings.0.weight", (C_, H), 0)
        b.add("encoder.channel_embeddings.1.weight", (H,), 0); b.add("encoder.channel_embeddings.1.bias", (H,), 0)
    if R:
        b.add("encoder.region_embeddings.0.weight", (R, H), 0)
        b.add("encoder.region_embeddings.1.weight", (H,), 0); b.add("encoder.region_embeddings.1.bias", (H,), 0)
    if embed_depth:
        b.add("encoder.depth_embeddings.0.weight", (H, 1), 0); b.add("encoder.depth_embeddings.0.bias", (H,), 0)
        b.add("encoder.depth_embeddings.2.weight", (H, H), 0); b.add("encoder.depth_embeddings.2.bias", (H,), 0)
        b.add("encoder.depth_embeddings.3.weight", (H,), 0); b.add("encoder.depth_embeddings.3.bias", (H,), 0)
    if use_cls:
        b.add("encoder.cls_embed.weight", (1, H), 0)
    b.end_segment()
    for l in range(L):
        _layer_entries(b, f"encoder.transformer.layers.{l}.", H, l + 1)
        b.end_segment()
    hs = L + 1
    b.add("encoder.transformer.norm.weight", (H,), hs); b.add("encoder.transformer.norm.bias", (H,), hs)
    if mlp_decoder:
        b.add("decoder.0.weight", (H, H), hs); b.add("decoder.0.bias", (H,), hs)
        b.add("decoder.2.weight", (T, H), hs); b.add("decoder.2.bias", (T,), hs)
    else:
        b.add("decoder.0.weight", (T, H), hs); b.add("decoder.0.bias", (T,), hs)
    b.end_segment()
    return b


def reference_order_init(cfg_shapes, seed=None, n_regions=0, dropout=0.0, embed_depth=False, embedder=None, embedder_dropout=0.0):
    """Initial weights drawn exactly as the reference's constructors draw them, in their order (itransformer.py:107-173,
    264-279), using torch's own layer constructors (pure CPU; torch is the init plumbing, nothing from the reference):
    the embedder (the MLP's two Linears - or the UnivariateTransformer's embed_spikes Linears, embed_pos / cls Embeddings, ONE
    TransformerEncoderLayer deep-copied into its layers, then embed_proj), channel / region Embedding tables, the depth MLP, the CLS
    Embedding, ONE nn.TransformerEncoderLayer that nn.TransformerEncoder deep-copies into every layer (so all layers start equal),
    then the decoder Linears. With torch.manual_seed(s) beforehand the model equals the reference's bit for bit."""
    T, H, L, nh, C_ = (cfg_shapes[k] for k in ("T", "H", "L", "nh", "C"))
    if seed is not None:
        torch.manual_seed(seed)
    p = {}

    def lin(name, i, o):
        m = nn.Linear(i, o)
        p[name + ".weight"], p[name + ".bias"] = m.weight.detach(), m.bias.detach()

    if embedder is None:
        lin("encoder.embed.0.0", T, H)
        lin("encoder.embed.0.3", H, H)
        p["encoder.embed.1.weight"], p["encoder.embed.1.bias"] = torch.ones(H), torch.zeros(H)
    else:   # UnivariateTransformer.__init__ (itransformer.py:48-73), then embed_proj (:121-124)
        h = embedder["h"]
        lin("encoder.embed.embed_spikes.0", 1, h)
        lin("encoder.embed.embed_spikes.2", h, h)
        p["encoder.embed.embed_pos.weight"] = nn.Embedding(T, h).weight.detach()
        p["encoder.embed.cls_embed.weight"] = nn.Embedding(1, h).weight.detach()
        elayer = nn.TransformerEncoderLayer(d_model=h, nhead=embedder["nh"], dim_feedforward=4 * h, dropout=embedder_dropout, batch_first=True)
        for l in range(embedder["L"]):
            for k, v in elayer.state_dict().items():
                p[f"encoder.embed.transformer.layers.{l}.{k}"] = v.detach().clone()
        p["encoder.embed.transformer.norm.weight"], p["encoder.embed.transformer.norm.bias"] = torch.ones(h), torch.zeros(h)
        lin("encoder.embed_proj.0", h, H)
        p["encoder.embed_proj.1.weight"], p["encoder.embed_proj.1.bias"] = torch.ones(H), torch.zeros(H)
    if C_:
        p["encoder.channel_embeddings.0.weight"] = nn.Embedding(C_, H).weight.detach()
        p["encoder.channel_embeddings.1.weight"], p["encoder.channel_embeddings.1.bias"] = torch.ones(H), torch.zeros(H)
    if n_regions:
        p["encoder.region_embeddings.0.weight"] = nn.Embedding(n_regions, H).weight.detach()
        p["encoder.region_embeddings.1.weight"], p["encoder.region_embeddings.1.bias"] = torch.ones(H), torch.zeros(H)
    if embed_depth:   # nn.Sequential(Linear(1,H), act, Linear(H,H), LayerNorm(H)) (itransformer.py:145-150)
        lin("encoder.depth_embeddings.0", 1, H)
        lin("encoder.depth_embeddings.2", H, H)
        p["encoder.depth_embeddings.3.weight"], p["encoder.depth_embeddings.3.bias"] = torch.ones(H), torch.zeros(H)
    if cfg_shapes["use_cls"]:
        p["encoder.cls_embed.weight"] = nn.Embedding(1, H).weight.detach()
    layer = nn.TransformerEncoderLayer(d_model=H, nhead=nh, dim_feedforward=4 * H, dropout=dropout, batch_first=True)
    for l in range(L):
        for k, v in layer.state_dict().items():
            p[f"encoder.transformer.layers.{l}.{k}"] = v.detach().clone()
    p["encoder.transformer.norm.weight"], p["encoder.transformer.norm.bias"] = torch.ones(H), torch.zeros(H)
    if cfg_shapes["mlp_decoder"]:
        lin("decoder.0", H, H)
        lin("decoder.2", H, T)
    else:
        lin("decoder.0", H, T)
    return p


class _ItrFunction(torch.autograd.Function):
    """Autograd bridge for callers that drive the model with loss.backward() (the reference's Trainer via accelerate)."""

    @staticmethod
    def forward(ctx, model, batch, *params):
        loss, preds = model._run_forward(batch, want_grad=True)
        ctx.model, ctx.fwd_id = model, model._fwd_id
        ctx.mark_non_differentiable(preds)
        return loss.sum(), preds

    @staticmethod
    def backward(ctx, g_loss, _g_preds):
        m = ctx.model
        bridge_check(m, ctx.fwd_id, "_ItrFunction")
        grads = torch.zeros_like(m._flat)
        m._run_backward(grads)
        grads.mul_(g_loss.to(grads.dtype))
        out = [None, None]
        for (_, off, numel, shape, _seg) in m._layout:
            out.append(grads[off:off + numel].view(shape))
        return tuple(out)


class iTransformer(FlatParamModule):
    """See module docstring. kwargs: method_name ("mlm"), loss ("poisson_nll" | "mse"), log_input (bool)
    (itransformer.py:221,287-295); extra: compute_dtype ("bf16" | "fp32", default bf16)."""

    def __init__(self, config, **kwargs):
        super().__init__()
        self.method = kwargs["method_name"]
        config = itransformer_config(config)
        enc_pt = config["encoder"].pop("from_pt", None)
        if enc_pt is not None:   # itransformer.py:226-229
            config["encoder"] = update_config(config.encoder, torch.load(os.path.join(enc_pt, "encoder_config.pth"), weights_only=False))
        dec_pt = config["decoder"].pop("from_pt", None)
        if dec_pt is not None:
            config["decoder"] = update_config(config.decoder, torch.load(os.path.join(dec_pt, "decoder_config.pth"), weights_only=False))
        if self.method != "mlm":
            raise Exception(f"Method {self.method} not implemented on the iTransformer HIP path (only 'mlm'; ctc / dyn_behaviour / "
                            "stat_behaviour stay on the reference implementation)")
        enc, dec = DictConfig(config["encoder"]), DictConfig(config["decoder"])
        if enc.embedder.mode not in ("mlp", "transformer"):
            raise Exception(f"embedder.mode {enc.embedder.mode} not implemented (itransformer.py:108-124 knows mlp | transformer)")
        self.embedder = None
        if enc.embedder.mode == "transformer":   # UnivariateTransformer(config.embedder) (itransformer.py:120)
            if enc.embedder.activation != "relu":
                raise Exception("iTransformer HIP path supports embedder.activation: relu")
            self.embedder = dict(h=int(enc.embedder.hidden_size), nh=int(enc.embedder.n_heads), L=int(enc.embedder.n_layers))
        if not enc.bias:
            raise Exception("iTransformer HIP path expects bias: true")
        self.loss_name, self.log_input = kwargs["loss"], bool(kwargs.get("log_input", True))
        if self.loss_name not in ("poisson_nll", "mse"):
            raise Exception(f"Loss {self.loss_name} not implemented yet for mlm")   # itransformer.py:295
        self.regions = list(enc.regions) if (enc.embed_region and enc.regions is not None) else None
        if enc.embed_region and self.regions is None:
            raise Exception("embed_region: true needs encoder.regions (itransformer.py:135-137)")
        self.region_to_indx = {r: i for i, r in enumerate(self.regions)} if self.regions else {}
        dtype_name = kwargs.get("compute_dtype", "bf16")
        self.compute_dtype = {"bf16": NBCI_BF16, "bfloat16": NBCI_BF16, "fp32": NBCI_F32, "float32": NBCI_F32}[dtype_name]
        c = ItrConfig()
        c.max_n_bins, c.hidden, c.n_heads, c.n_layers = enc.embedder.max_n_bins, enc.hidden_size, enc.n_heads, enc.n_layers
        c.max_n_channels, c.n_regions = enc.max_n_channels, len(self.regions) if self.regions else 0
        c.act, c.dec_act = ACT[enc.activation], ACT[dec.activation]
        c.embed_dropout, c.dropout = float(enc.embedder.dropout), float(enc.dropout)
        c.use_cls, c.mlp_decoder = (1 if dec.use_cls else 0), (1 if dec.mlp_decoder else 0)
        c.loss = LOSS_KIND[(self.loss_name, self.log_input)]
        c.dtype = self.compute_dtype
        # storage of the LayerNorm inputs and the gradient streams between kernels (as NDT1's residual_dtype): "fp32" by default
        # (parity: the reference keeps these in f32 under autocast), "bf16" opt-in on the bf16 path
        res_name = kwargs.get("residual_dtype", None) or "fp32"
        self.residual_dtype = {"bf16": NBCI_BF16, "bfloat16": NBCI_BF16, "fp32": NBCI_F32, "float32": NBCI_F32}[res_name]
        if self.residual_dtype == NBCI_BF16 and self.compute_dtype != NBCI_BF16:
            raise Exception("residual_dtype 'bf16' needs compute_dtype 'bf16'")
        c.residual_dtype = self.residual_dtype
        c.embed_depth = 1 if enc.embed_depth else 0
        if self.embedder is not None:
            c.emb_mode, c.emb_hidden, c.emb_heads, c.emb_layers = 1, self.embedder["h"], self.embedder["nh"], self.embedder["L"]
        self._ccfg = c
        self.config = config
        self.use_cls = bool(dec.use_cls)
        self.masker_cfg = [(k, DictConfig(m)) for k, m in config["masker"].items()]
        b = layout_of(c.max_n_bins, c.hidden, c.n_layers, c.max_n_channels, c.n_regions, c.use_cls, c.mlp_decoder,
                      embed_depth=bool(c.embed_depth), embedder=self.embedder)
        self._layout, self._segments, self._total = b.entries, b.segments, b.cur
        flat = torch.zeros(self._total, dtype=torch.float32)
        init = reference_order_init(dict(T=c.max_n_bins, H=c.hidden, L=c.n_layers, nh=c.n_heads, C=c.max_n_channels,
                                         use_cls=c.use_cls, mlp_decoder=c.mlp_decoder), n_regions=c.n_regions, dropout=c.dropout,
                                    embed_depth=bool(c.embed_depth), embedder=self.embedder, embedder_dropout=c.embed_dropout)
        for (name, off, numel, _shape, _seg) in self._layout:
            flat[off:off + numel] = init[name].reshape(-1)
        self._plan = None
        self._adopt(flat)
        if enc_pt is not None:
            self.encoder.load_state_dict(torch.load(os.path.join(enc_pt, "encoder.bin")))
        if dec_pt is not None:
            self.decoder.load_state_dict(torch.load(os.path.join(dec_pt, "decoder.bin")))
        self._io_keepalive = None
        self._step_seed = 0
        self.mask_override = None      # (B,T,N) int64: replaces the maskers' draws (tests / replay of a recorded mask)
        self.last_n_examples = None

    # ------------------------------------------------------------------ plan / buffers
    def _ensure_plan(self):
        if self._plan is not None:
            return
        plan = C.c_void_p()
        check(lib().nbci_itr_plan_create(C.byref(self._ccfg), C.byref(plan)), "nbci_itr_plan_create")
        self._plan = plan
        self._check_layout("nbci_itr_", plan)

    def __del__(self):
        try:
            if getattr(self, "_plan", None) is not None:
                lib().nbci_itr_plan_destroy(self._plan)
        except Exception:
            pass

    def _workspace(self, B, N):
        need = lib().nbci_itr_workspace_bytes(self._plan, B, N)
        if need < 0:
            check(-1, "nbci_itr_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._flat.device)
        return self._ws, need

    # ------------------------------------------------------------------ maskers (models/masker.py:44-104)
    def _apply_maskers(self, spikes, neuron_regions, seed):
        """Returns (masked spikes, OR of the masks) — itransformer.py:322-326 — leaving `spikes` untouched."""
        B, T, N = spikes.shape
        dev = spikes.device
        mask = torch.zeros(B, T, N, dtype=torch.int64, device=dev)
        masked, keep = spikes, []
        scratch = torch.empty(1, dtype=torch.int32, device=dev)
        first = True
        for k, (name, mc) in enumerate(self.masker_cfg):
            active = mc.get("active", True)   # configs/itransformer.yaml carries no `active` key; the mlm recipe means "on"
            if not active or (not self.training and not mc.get("force_active", False)):
                continue
            d = MaskerDesc()
            d.B, d.T, d.N = B, T, N
            mode = mc["mode"]
            d.ratio, d.timespan = float(mc["ratio"]), 1
            d.zero_ratio, d.random_ratio = float(mc["zero_ratio"]), float(mc["random_ratio"])
            d.seed, d.site = seed, SITE_MASKER + 8 * k
            if self.mask_override is not None:
                ext = self.mask_override.to(dev).long().contiguous()
                d.mode, d.ext_mask = MASK_MODE["given"], _ptr(ext)
                keep.append(ext)
            elif mode == "temporal":
                span = masker_timespan(seed, d.site, mc.get("expand_prob", 0.0), mc.get("max_timespan", 1))
                d.mode, d.timespan, d.ratio = MASK_MODE["temporal"], span, float(mc["ratio"]) / span
            elif mode in ("neuron", "random"):
                d.mode = MASK_MODE[mode]
            elif mode == "region":
                assert neuron_regions is not None, "Can't mask region without brain region information"   # masker.py:68
                assert mc.get("regions") is not None, "No regions to mask"
                pr = np.isin(np.asarray(neuron_regions), list(mc["regions"])).astype(np.float32)
                probs = torch.from_numpy(pr).to(dev).contiguous()
                d.mode, d.probs = MASK_MODE["region"], _ptr(probs)
                keep.append(probs)
            elif mode == "co-smooth":
                assert mc.get("channels") is not None, "No channels to mask"
                pr = np.zeros(N, np.float32)
                pr[list(mc["channels"])] = 1
                probs = torch.from_numpy(pr).to(dev)
                d.mode, d.probs = MASK_MODE["co-smooth"], _ptr(probs)
                keep.append(probs)
            elif mode == "forward-pred":          # "masker copy.py":81-85: a fixed set of time steps
                assert mc.get("timesteps") is not None, "No time steps to mask"
                pr = np.zeros(T, np.float32)
                pr[list(mc["timesteps"])] = 1
                probs = torch.from_numpy(pr).to(dev)
                d.mode, d.probs = MASK_MODE["table_t"], _ptr(probs)
                keep.append(probs)
            elif mode == "inter-region":          # :86-94: `ratio` of the neurons of n_mask_regions sampled regions; targets = the mask
                assert neuron_regions is not None, "Can't mask region without brain region information"
                assert mc.get("mask_regions") is not None, "No regions to mask"
                chosen = region_sample(seed, d.site, mc["mask_regions"], mc.get("n_mask_regions", 1))
                pr = np.isin(np.asarray(neuron_regions), chosen).astype(np.float32) * np.float32(mc["ratio"])
                probs = torch.from_numpy(pr).to(dev).contiguous()
                d.mode, d.probs = MASK_MODE["region"], _ptr(probs)
                keep.append(probs)
            elif mode == "intra-region":          # :95-104,133: everything outside the sampled target regions is masked, `ratio` inside;
                assert neuron_regions is not None, "Can't mask region without brain region information"   # targets = masked bins inside
                assert mc.get("target_regions") is not None, "No target regions"
                chosen = region_sample(seed, d.site, mc["target_regions"], mc.get("n_mask_regions", 1))
                tgt = np.isin(np.asarray(neuron_regions), chosen)
                pr = np.where(tgt, np.float32(mc["ratio"]), np.float32(1.0)).astype(np.float32)
                probs = torch.from_numpy(pr).to(dev).contiguous()
                target = torch.from_numpy(tgt.astype(np.float32)).to(dev).contiguous()
                d.mode, d.probs, d.target_bn = MASK_MODE["region"], _ptr(probs), _ptr(target)
                keep += [probs, target]
            else:
                raise Exception(f"Masking mode {mode} not implemented")
            if first:
                out = torch.empty_like(spikes)
            else:
                out = masked
            d.in_, d.out, d.mask = _ptr(masked), _ptr(out), _ptr(mask)
            d.accumulate = 0 if first else 1
            d.scratch = _ptr(scratch)
            check(lib().nbci_masker(C.byref(d), _stream()), "nbci_masker")
            masked, first = out, False
        keep.append(scratch)
        return masked, mask, keep

    # ------------------------------------------------------------------ forward / backward
    def _run_forward(self, batch, want_grad, seed=None, grad_scale=1.0, hidden_out=None):
        spikes = batch["spikes"]
        if not spikes.is_cuda:
            raise _lib.NbciUnavailable("iTransformer (HIP path) needs tensors on a ROCm device; there is no CPU fallback")
        self._ensure_plan()
        bridge_stamp(self)
        if self.compute_dtype == NBCI_BF16 and self._flat_lp is None:
            self.refresh_lp()
        dev = spikes.device
        B, T, N = spikes.shape
        if T != self._ccfg.max_n_bins:
            raise ValueError(f"expected {self._ccfg.max_n_bins} time bins (embedder.max_n_bins), got {T}")
        spikes = spikes.contiguous().float()
        smask = batch["spikes_mask"].contiguous().long()
        ss = batch.get("spikes_spacestamp")
        if ss is not None:
            ss = ss.long().expand(B, N).contiguous() if ss.dim() == 1 else ss.contiguous().long()
        regions = batch.get("neuron_regions")
        ridx = None
        if self._ccfg.n_regions:   # itransformer.py:196: region_to_indx of every neuron's region name
            if regions is None:
                raise ValueError("embed_region: true needs neuron_regions in the batch (itransformer.py:195-198)")
            ridx = torch.from_numpy(np.array([[self.region_to_indx[str(r)] for r in row] for row in np.asarray(regions)], dtype=np.int64)
                                    .reshape(B, N)).to(dev)
        depths = batch.get("neuron_depths")
        if self._ccfg.embed_depth:
            if depths is None:
                raise ValueError("embed_depth: true needs neuron_depths in the batch (itransformer.py:200-202)")
            depths = torch.as_tensor(depths, dtype=torch.float32).to(dev).reshape(B, N).contiguous()
        else:
            depths = None
        ts = batch.get("spikes_timestamp") if self._ccfg.emb_mode == 1 else None
        if ts is not None:
            ts = ts.to(dev).long().reshape(B, T).contiguous()
        if seed is None:
            self._step_seed = (self._step_seed * 1664525 + 1013904223) & 0xFFFFFFFF
            seed = self._step_seed
        masked, mask, keep = self._apply_maskers(spikes, regions, seed)
        ws, need = self._workspace(B, N)
        preds = torch.empty(B, T, N, dtype=torch.float32, device=dev)
        mask_out = torch.empty(B, T, N, dtype=torch.int64, device=dev)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        nex = torch.zeros(1, dtype=torch.int64, device=dev)
        io = ItrIO()
        io.B, io.N = B, N
        io.spikes, io.masked, io.mask, io.spikes_mask = _ptr(spikes), _ptr(masked), _ptr(mask), _ptr(smask)
        io.spikes_spacestamp, io.region_idx = _ptr(ss), _ptr(ridx)
        io.spikes_timestamp, io.neuron_depths = _ptr(ts), _ptr(depths)
        io.train = 1 if self.training else 0
        io.want_grad = 1 if want_grad else 0
        io.seed, io.grad_scale = seed, grad_scale
        io.preds, io.mask_out, io.loss, io.n_examples = _ptr(preds), _ptr(mask_out), _ptr(loss), _ptr(nex)
        io.hidden_out = _ptr(hidden_out)
        io.workspace, io.workspace_bytes = _ptr(ws), need
        check(lib().nbci_itr_forward(self._plan, _ptr(self._flat), _ptr(self._flat_lp), C.byref(io), _stream()), "nbci_itr_forward")
        self._io_keepalive = (io, spikes, masked, mask, smask, ss, ridx, ws, preds, mask_out, loss, nex, hidden_out, keep, ts, depths)
        self.last_n_examples = nex
        self.last_mask = mask_out
        self.last_targets = spikes
        return loss, preds

    def _run_backward(self, grads, seg_hi=None, seg_lo=0):
        io = self._io_keepalive[0]
        if not io.want_grad:
            raise RuntimeError("backward called but the forward pass ran without want_grad")
        if seg_hi is None:
            seg_hi = self._ccfg.n_layers + 1
        check(lib().nbci_itr_backward(self._plan, _ptr(self._flat), _ptr(self._flat_lp), C.byref(io), _ptr(grads), seg_hi, seg_lo,
                                      _stream()), "nbci_itr_backward")

    def forward(self, spikes, spikes_mask, spikes_timestamp, spikes_spacestamp=None, spikes_lengths=None, targets=None,
                targets_lengths=None, neuron_regions=None, neuron_depths=None):
        batch = dict(spikes=spikes, spikes_mask=spikes_mask, spikes_spacestamp=spikes_spacestamp, neuron_regions=neuron_regions,
                     neuron_depths=neuron_depths, spikes_timestamp=spikes_timestamp)
        bridge_begin(self)   # an external optimizer may have stepped the f32 views since the bf16 shadow was taken
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list):
            loss, preds = _ItrFunction.apply(self, batch, *self._param_list)
        else:
            loss_vec, preds = self._run_forward(batch, want_grad=False)
            loss = loss_vec.sum()
        return iTransformerOutput(loss=loss, n_examples=self.last_n_examples.reshape(()), mask=self.last_mask, preds=preds,
                                  targets=self.last_targets)

    # ------------------------------------------------------------------ checkpoints (itransformer.py:367-375)
    def save_checkpoint(self, save_dir):
        torch.save({k: v.detach().clone() for k, v in self.encoder.state_dict().items()}, os.path.join(save_dir, "encoder.bin"))
        torch.save(dict(self.config.encoder), os.path.join(save_dir, "encoder_config.pth"))
        torch.save({k: v.detach().clone() for k, v in self.decoder.state_dict().items()}, os.path.join(save_dir, "decoder.bin"))
        torch.save(dict(self.config.decoder), os.path.join(save_dir, "decoder_config.pth"))

    def load_checkpoint(self, load_dir):
        self.encoder.load_state_dict(torch.load(os.path.join(load_dir, "encoder.bin")))
        self.decoder.load_state_dict(torch.load(os.path.join(load_dir, "decoder.bin")))
        self.refresh_lp()
