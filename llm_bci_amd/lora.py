"""Minimal LoRA injection for the stock HF LLM inside BCI, used only when `peft` is not importable (it is absent from this image).

The reference builds its adapters with peft (models/bci.py:10,56-62: LoraConfig(r, lora_alpha, lora_dropout, target_modules) +
get_peft_model). peft is a third-party dependency that is not under /root/reference and is not pinned by it; what is restated here
is LoRA's published definition as peft's `lora.Linear` implements it:
    y = base(x) + (lora_alpha / r) * B(A(dropout(x))),   A: Linear(in, r, bias=False) ~ kaiming_uniform(a=sqrt(5)),
                                                          B: Linear(r, out, bias=False) = 0,
every non-adapter parameter frozen. Module / parameter names follow peft's (`base_layer`, `lora_A.default.weight`,
`lora_B.default.weight`) so that an adapter state dict looks the same.
"""
import math

import torch
import torch.nn as nn


class LoRALinear(nn.Module):
    def __init__(self, base, r, alpha, dropout):
        super().__init__()
        self.base_layer = base
        self.lora_dropout = nn.ModuleDict({"default": nn.Dropout(p=dropout) if dropout > 0.0 else nn.Identity()})
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False)})
        nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B["default"].weight)
        self.scaling = alpha / r
        self.to(base.weight.device)

    def forward(self, x):
        a = self.lora_A["default"]
        y = self.lora_B["default"](a(self.lora_dropout["default"](x).to(a.weight.dtype)))
        return self.base_layer(x) + (y * self.scaling).to(x.dtype)


def inject_lora(model, r, alpha, dropout, target_modules):
    """Wrap every nn.Linear whose attribute name is in `target_modules`; freeze everything that is not an adapter weight."""
    for p in model.parameters():
        p.requires_grad = False
    targets = set(target_modules)
    n = 0
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            if name in targets and isinstance(child, nn.Linear):
                setattr(parent, name, LoRALinear(child, r, alpha, dropout))
                n += 1
    if n == 0:
        raise ValueError(f"LoRA: no nn.Linear named any of {sorted(targets)} in the model")
    return model
