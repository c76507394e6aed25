"""One reference train step on the CPU (forward -> CTC sum-loss -> backward -> AdamW), assembled
from the oracle pieces: what models/trainer.py:336-343 does per batch. Used by tests and by
bench.py's cpu_baseline leg (kind "port"). Test infrastructure only."""
import numpy as np

from . import ndt1 as O
from . import optim as OO


class CpuTrainer:
    def __init__(self, cfg, params, lr=1e-3, wd=5e-5, eps=1e-8, total_steps=100, pct_start=0.0, div_factor=25.0,
                 beta2=0.999, ga=1, world=1):
        self.cfg, self.p = cfg, {k: np.array(v, np.float32) for k, v in params.items()}
        self.m = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.lr, self.wd, self.eps, self.beta2 = lr, wd, eps, beta2
        self.total, self.pct, self.div = total_steps, pct_start, div_factor
        self.ga, self.world = ga, world
        self.global_step, self.opt_step = 1, 0
        self.acc = None

    def step(self, batch, train=True, seed=0, reduce_fn=None):
        """reduce_fn(grads_dict) -> grads_dict summed over ranks (DDP), applied on sync steps."""
        sync = ((self.global_step - 1) % self.ga == 0)
        out, cache = O.forward(self.cfg, self.p, batch, train=train, seed=seed)
        g = O.backward(cache, grad_scale=1.0 / self.ga)
        if self.acc is None:
            self.acc = g
        else:
            for k in g:
                self.acc[k] += g[k]
        if sync:
            grads = self.acc if reduce_fn is None else reduce_fn(self.acc)
            lr, b1 = OO.onecycle(self.opt_step, self.total, self.lr, self.pct, self.div)
            t = self.opt_step + 1
            for k in self.p:
                OO.adamw_step(self.p[k], grads[k] * np.float32(1.0 / self.world), self.m[k], self.v[k], t, lr, b1, self.beta2,
                              self.eps, self.wd)
            self.acc = None
            self.opt_step += 1
        self.global_step += 1
        return out
