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


# ---------------------------------------------------------------------------------------------------------------------
# adapter-only checkpoints in peft's on-disk format (what `PeftModel.save_pretrained` writes for the reference, models/bci.py:252):
#   adapter_config.json        peft's LoraConfig fields (the ones that define the adapter)
#   adapter_model.safetensors  ONLY the lora_A / lora_B tensors, keyed `base_model.model.<module path>.lora_A.weight`
#                              (peft drops the adapter name "default" from the keys it saves)
# so a directory written here loads with peft where peft exists, and a directory written by peft loads here.
# ---------------------------------------------------------------------------------------------------------------------
ADAPTER_CONFIG, ADAPTER_WEIGHTS = "adapter_config.json", "adapter_model.safetensors"


def lora_modules(model):
    return [(n, m) for n, m in model.named_modules() if isinstance(m, LoRALinear)]


def has_injected_lora(model):
    return any(True for _ in lora_modules(model))


def adapter_state_dict(model):
    out = {}
    for name, m in lora_modules(model):
        out[f"base_model.model.{name}.lora_A.weight"] = m.lora_A["default"].weight.detach().contiguous().cpu()
        out[f"base_model.model.{name}.lora_B.weight"] = m.lora_B["default"].weight.detach().contiguous().cpu()
    return out


def save_adapter(model, save_dir):
    """Adapter-only checkpoint of a model wrapped by inject_lora (module docstring above)."""
    import json
    import os
    from safetensors.torch import save_file
    mods = lora_modules(model)
    if not mods:
        raise ValueError("save_adapter: the model carries no LoRA wrappers")
    m0 = mods[0][1]
    r = m0.lora_A["default"].weight.shape[0]
    drop = m0.lora_dropout["default"]
    alpha = float(m0.scaling * r)
    cfg = {"peft_type": "LORA", "task_type": None, "inference_mode": True, "r": int(r),
           "lora_alpha": int(round(alpha)) if abs(alpha - round(alpha)) < 1e-9 else alpha,   # (peft writes an int when it is one)
           "lora_dropout": float(drop.p) if isinstance(drop, nn.Dropout) else 0.0, "bias": "none", "fan_in_fan_out": False,
           "target_modules": sorted({n.rsplit(".", 1)[-1] for n, _ in mods}), "modules_to_save": None,
           "base_model_name_or_path": getattr(getattr(model, "config", None), "_name_or_path", None) or None, "init_lora_weights": True}
    os.makedirs(save_dir, exist_ok=True)
    with open(os.path.join(save_dir, ADAPTER_CONFIG), "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)
    save_file(adapter_state_dict(model), os.path.join(save_dir, ADAPTER_WEIGHTS), metadata={"format": "pt"})


def is_adapter_dir(path):
    import os
    return os.path.exists(os.path.join(path, ADAPTER_CONFIG))


def load_adapter(model, load_dir):
    """Load an adapter-only checkpoint (written by save_adapter or by peft) into `model`: wrappers are injected first when the model
    has none (from adapter_config.json), then every lora_A / lora_B tensor of the file must find its module, and every wrapper its
    two tensors — nothing is silently re-initialised."""
    import json
    import os
    from safetensors.torch import load_file
    cfg = json.load(open(os.path.join(load_dir, ADAPTER_CONFIG)))
    if cfg.get("peft_type", "LORA") != "LORA":
        raise ValueError(f"load_adapter: peft_type {cfg.get('peft_type')} is not LoRA")
    # Anything that changes the adapter's scaling or structure beyond plain LoRA (W + (alpha / r) B A on `target_modules`) would load
    # with the wrong arithmetic: refuse it by name instead.
    unsupported = [k for k, bad in (("use_rslora", bool(cfg.get("use_rslora"))), ("use_dora", bool(cfg.get("use_dora"))),
                                    ("rank_pattern", bool(cfg.get("rank_pattern"))), ("alpha_pattern", bool(cfg.get("alpha_pattern"))),
                                    ("bias", cfg.get("bias", "none") not in ("none", None)),
                                    ("modules_to_save", bool(cfg.get("modules_to_save"))),
                                    ("fan_in_fan_out", bool(cfg.get("fan_in_fan_out"))),
                                    ("layers_to_transform", cfg.get("layers_to_transform") is not None)) if bad]
    if unsupported:
        raise ValueError(f"load_adapter: {os.path.join(load_dir, ADAPTER_CONFIG)} uses {', '.join(unsupported)}, which this loader does not "
                         "implement (plain LoRA only: r, lora_alpha, lora_dropout, target_modules); load it through peft instead")
    if not has_injected_lora(model):
        inject_lora(model, int(cfg["r"]), float(cfg["lora_alpha"]), float(cfg.get("lora_dropout", 0.0)), cfg["target_modules"])
    wpath = os.path.join(load_dir, ADAPTER_WEIGHTS)
    sd = load_file(wpath) if os.path.exists(wpath) else torch.load(os.path.join(load_dir, "adapter_model.bin"), map_location="cpu")
    mods = dict(lora_modules(model))
    seen = set()
    def locate(key):
        """(module path, "lora_A" | "lora_B") of a saved tensor name; peft drops the adapter name, a raw state dict keeps it."""
        name = key[len("base_model.model."):] if key.startswith("base_model.model.") else key
        for part in ("lora_A", "lora_B"):
            for suffix in (f".{part}.weight", f".{part}.default.weight"):
                if name.endswith(suffix):
                    return name[:-len(suffix)], part
        return None, None

    for k, v in sd.items():
        path, part = locate(k)
        if part is None:
            raise KeyError(f"load_adapter: unexpected tensor {k} in an adapter checkpoint")
        if path not in mods:
            raise KeyError(f"load_adapter: no LoRA wrapper for {k}")
        w = getattr(mods[path], part)["default"].weight
        if tuple(w.shape) != tuple(v.shape):
            raise ValueError(f"load_adapter: {k} has shape {tuple(v.shape)}, the wrapper expects {tuple(w.shape)}")
        with torch.no_grad():
            w.copy_(v.to(w.dtype))
        seen.add((path, part))
    missing = [(n, p) for n in mods for p in ("lora_A", "lora_B") if (n, p) not in seen]
    if missing:
        raise KeyError(f"load_adapter: checkpoint has no tensors for {missing[:4]}{' ...' if len(missing) > 4 else ''}")
    return model
