"""Learning-rate / momentum schedules the reference's Trainer can select
(models/trainer.py:233-253): "cosine" = torch OneCycleLR(anneal cos, cycle_momentum: beta1 moves
0.95 -> 0.85 -> 0.95 inversely to lr), "linear" = HF linear warmup/decay, "step" = StepLR per epoch.
Returned values are what optimizer.step() number `step` (0-based) uses."""
import math


def _cos(a, b, pct):
    return b + (a - b) * 0.5 * (1.0 + math.cos(math.pi * pct))


class OneCycle:
    def __init__(self, total_steps, max_lr, pct_start=0.0, div_factor=25.0, final_div_factor=1e4,
                 base_momentum=0.85, max_momentum=0.95):
        if total_steps <= 0:
            raise ValueError("Expected positive integer total_steps")
        self.total = total_steps
        self.max_lr = max_lr
        self.init_lr = max_lr / div_factor
        self.min_lr = self.init_lr / final_div_factor
        self.m_lo, self.m_hi = base_momentum, max_momentum
        self.knee = float(pct_start * total_steps) - 1.0

    def at(self, step):
        if step >= self.total:
            raise ValueError(f"Tried to step {step + 1} times. The specified number of total steps is {self.total}")
        if step <= self.knee:  # warm-up leg
            pct = step / self.knee if self.knee > 0 else 1.0
            return _cos(self.init_lr, self.max_lr, pct), _cos(self.m_hi, self.m_lo, pct)
        pct = (step - self.knee) / ((self.total - 1) - self.knee)
        return _cos(self.max_lr, self.min_lr, pct), _cos(self.m_lo, self.m_hi, pct)


class LinearWarmup:
    def __init__(self, total_steps, lr, warmup_steps, beta1=0.9):
        self.total, self.lr, self.warm, self.beta1 = total_steps, lr, warmup_steps, beta1

    def at(self, step):
        if step < self.warm:
            f = step / max(1, self.warm)
        else:
            f = max(0.0, (self.total - step) / max(1, self.total - self.warm))
        return self.lr * f, self.beta1


class StepDecay:
    """StepLR(step_size=1, gamma) stepped once per EPOCH (trainer.py:248-251,418-419)."""

    def __init__(self, lr, gamma, beta1=0.9):
        self.lr, self.gamma, self.beta1, self.epoch = lr, gamma, beta1, 0

    def at(self, step):
        return self.lr * (self.gamma ** self.epoch), self.beta1

    def end_epoch(self):
        self.epoch += 1
