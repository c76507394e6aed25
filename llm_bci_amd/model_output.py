"""Return contract of every model behind the Trainer plugin surface (reference:
models/model_output.py:11-17, models/ndt1.py:20-26): `loss` is a SUM over examples,
`n_examples` a separate count; `.to_dict()` is what metric functions receive."""
from dataclasses import dataclass, fields
from typing import Optional

import torch


@dataclass
class ModelOutput:
    loss: Optional[torch.Tensor] = None
    n_examples: Optional[torch.Tensor] = None

    def to_dict(self):
        return {f.name: getattr(self, f.name) for f in fields(self)}


@dataclass
class NDT1Output(ModelOutput):
    mask: Optional[torch.Tensor] = None
    preds: Optional[torch.Tensor] = None
    targets: Optional[torch.Tensor] = None
