"""Flat-parameter plumbing shared by the HIP-backed modules: every nn.Parameter is a view into ONE f32 buffer
(plus a bf16 shadow in bf16 mode), so the C side sees a single pointer, AdamW is one fused launch and each
backward segment's gradients are one contiguous RCCL bucket. State-dict keys stay the reference's."""
import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib
from ._lib import NBCI_BF16, check, lib


class _Box(nn.Module):
    """Parameter container; only exists so state_dict() keys match the reference's module tree."""


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bridge_begin(model):
    """Top of every autograd-bridge forward. The bf16 shadow is only kept in sync by the fused AdamW of NativeTrainer; on the
    registry-swap route (loss.backward() + torch.optim.* on model.parameters()) the optimizer writes the f32 views in place, so the
    shadow is rebuilt whenever any parameter's version counter moved since it was taken."""
    if model.compute_dtype == NBCI_BF16:
        ver = sum(p._version for p in model._param_list)
        if model._flat_lp is None or getattr(model, "_lp_version", None) != ver:
            model.refresh_lp()
            model._lp_version = ver


def bridge_stamp(model):
    """Every forward gets a number; a bridge backward refuses to run on another forward's saved activations (the workspace and
    the borrowed io tensors are single-slot state of the most recent forward)."""
    model._fwd_id = getattr(model, "_fwd_id", 0) + 1
    return model._fwd_id


def bridge_check(model, fwd_id, what):
    if getattr(model, "_fwd_id", None) != fwd_id:
        raise RuntimeError(f"{what}: backward of forward #{fwd_id} requested, but the model has since run forward "
                           f"#{getattr(model, '_fwd_id', None)}; its saved activations are gone (one grad-enabled forward per backward)")


class LayoutBuilder:
    """Same placement rule as the C++ plans: tensors in canonical order, each aligned to 8 elements, segments too."""

    def __init__(self):
        self.entries, self.segments, self.cur, self._seg_begin = [], [], 0, 0

    def add(self, name, shape, seg):
        self.cur = (self.cur + 7) // 8 * 8
        n = int(math.prod(shape))
        self.entries.append((name, self.cur, n, tuple(shape), seg))
        self.cur += n

    def end_segment(self):
        self.cur = (self.cur + 7) // 8 * 8
        self.segments.append((self._seg_begin, self.cur))
        self._seg_begin = self.cur


class FlatParamModule(nn.Module):
    """Subclasses set self._layout [(name, off, numel, shape, seg)], self._segments, self._total, self.compute_dtype, fill a
    CPU flat tensor and call _adopt(flat). `_top_modules` lists the top-level submodule names owned by the layout."""
    _top_modules = ("encoder", "decoder")

    def _adopt(self, flat):
        self._flat = flat
        self._flat_lp = None
        self._ws = None
        self._bind_parameters()

    def _bind_parameters(self):
        for top in self._top_modules:
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
        """.to()/.cuda() move every Parameter separately; re-flatten afterwards so the C side keeps seeing one buffer."""
        super()._apply(fn, *a, **k)
        first = self._param_list[0]
        flat = torch.zeros(self._total, dtype=torch.float32, device=first.device)
        named = dict(self.named_parameters())
        for (name, off, numel, _shape, _seg) in self._layout:
            flat[off:off + numel] = named[name].detach().reshape(-1).float()
        self._flat = flat
        self._flat_lp = None
        self._ws = None
        with torch.no_grad():
            for (name, off, numel, shape, _seg), p in zip(self._layout, self._param_list):
                p.data = flat[off:off + numel].view(shape)
                p.grad = None
        return self

    def refresh_lp(self):
        """(Re)build the bf16 shadow (after load_state_dict / manual edits; the fused AdamW keeps it in sync on its own)."""
        if self.compute_dtype == NBCI_BF16:
            self._flat_lp = self._flat.to(torch.bfloat16)

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._flat_lp = None
        return out

    def _check_layout(self, prefix, plan):
        """The Python layout must equal the C++ plan's (names, offsets, sizes, segments)."""
        l = lib()
        total = getattr(l, prefix + "param_count")(plan)
        n = getattr(l, prefix + "num_params")(plan)
        if total != self._total or n != len(self._layout):
            raise _lib.NbciError(f"parameter layout mismatch between the Python module and the {prefix}* plan")
        buf = C.create_string_buffer(160)
        off, numel, rows, cols, seg = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        mine = {nm: (o, k, sg) for (nm, o, k, _s, sg) in self._layout}
        for i in range(n):
            check(getattr(l, prefix + "param_info")(plan, i, buf, 160, C.byref(off), C.byref(numel), C.byref(rows), C.byref(cols),
                                                    C.byref(seg)), prefix + "param_info")
            if mine.get(buf.value.decode()) != (off.value, numel.value, seg.value):
                raise _lib.NbciError(f"parameter layout mismatch for {buf.value.decode()}")
