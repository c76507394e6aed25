"""AdamW + schedulers as the reference's trainer drives them (models/trainer.py:229,233-253,
340-343). Test infrastructure only.

The arithmetic lives in torch.optim (torch 2.10 in the build container), not under
/root/reference; restated from its documented update rule and pinned by the golden fixture
"adamw_*" produced from torch.optim.AdamW + OneCycleLR themselves.
"""
import math

import numpy as np


def onecycle(step, total_steps, max_lr, pct_start=0.0, div_factor=25.0, final_div_factor=1e4,
             base_momentum=0.85, max_momentum=0.95):
    """(lr, beta1) that OneCycleLR(anneal='cos', cycle_momentum=True, three_phase=False) has set
    when optimizer.step() number `step` (0-based) runs (trainer.py:240-246)."""
    initial_lr = max_lr / div_factor
    min_lr = initial_lr / final_div_factor
    phases = [
        (float(pct_start * total_steps) - 1, initial_lr, max_lr, max_momentum, base_momentum),
        (total_steps - 1, max_lr, min_lr, base_momentum, max_momentum),
    ]

    def cos_anneal(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)

    start = 0.0
    for i, (end, lr0, lr1, m0, m1) in enumerate(phases):
        if step <= end or i == len(phases) - 1:
            pct = (step - start) / (end - start)
            return cos_anneal(lr0, lr1, pct), cos_anneal(m0, m1, pct)
        start = end
    raise AssertionError


def linear_warmup(step, warmup_steps, total_steps, lr):
    """transformers.get_linear_schedule_with_warmup (trainer.py:234-238)."""
    if step < warmup_steps:
        return lr * step / max(1, warmup_steps)
    return lr * max(0.0, (total_steps - step) / max(1, total_steps - warmup_steps))


def adamw_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0):
    """One torch.optim.AdamW step (amsgrad=False), in place on float32 arrays; t is 1-based.
    bias_correction1 uses the CURRENT beta1 ** t, as PyTorch does under cycle_momentum."""
    f = np.float32
    p *= f(1.0 - lr * wd)
    m *= f(beta1); m += f(1.0 - beta1) * g
    v *= f(beta2); v += f(1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    denom = np.sqrt(v) / f(math.sqrt(bc2)) + f(eps)
    p -= f(lr / bc1) * (m / denom)
    return p, m, v
