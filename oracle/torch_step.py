"""PyTorch-CPU restatement of the reference's NDT1-CTC train step (SURVEY §8(d) "CPU baseline"): the same sequence of
torch.nn.functional calls the reference's eager path dispatches (models/ndt1.py:92-107 smoothing conv + noise, :160-203 embed /
softsign / unfold / stack projection / position table, :266-292 three Linears + SDPA with an explicit boolean mask, :224-227 MLP,
:317-330 pre-LN residual layers, :442 out_norm, :494-499 decoder + log-softmax, :517,581 CTCLoss sum) -> autograd backward
(trainer.py:339) -> torch.optim.AdamW + OneCycleLR step (trainer.py:229,240-246,340-343), dropout / noise ON as the recipe has
them (torch's own RNG: the draws differ from the HIP kernels', the WORK is the same).

Test infrastructure: only tests/ and bench.py's cpu_baseline leg import it. It exists because the reference itself cannot travel
to the GPU box; it is pinned to the reference's own outputs by tests/test_oracle_torch_step.py (g_c1 / g_c2: log-probs, loss,
gradients, two AdamW + OneCycle steps). Parameters are a dict keyed by the reference's state-dict names.
"""
import math

import torch
import torch.nn.functional as F


def default_hparams():
    return dict(stack_size=32, stack_stride=4, n_heads=8, n_layers=5, smooth_sd=2, white_noise_sd=1.0, constant_offset_sd=0.2,
                embed_dropout=0.2, dropout=0.4, blank_id=0)


def _taps(sd, dtype):
    n = 1 + 6 * sd                                      # ndt1.py:87 (signal.gaussian(1 + 6 sd, sd), normalised, built in f64)
    k = torch.exp(-0.5 * ((torch.arange(n, dtype=torch.float64) - (n - 1) / 2.0) / sd) ** 2)
    return (k / k.sum()).to(dtype)


def forward(p, batch, hp, train):
    """returns (loss sum, log-probs (B,T',V), token lengths)."""
    x = batch["spikes"]
    B, T, N = x.shape
    L, nh, S, st = hp["n_layers"], hp["n_heads"], hp["stack_size"], hp["stack_stride"]
    k = _taps(hp["smooth_sd"], x.dtype)
    x = F.conv1d(x.transpose(1, 2), k.view(1, 1, -1).expand(N, 1, -1), padding="same", groups=N).transpose(1, 2)
    if train and hp.get("noise", True):
        x = x + hp["white_noise_sd"] * torch.randn_like(x)
        x = x + hp["constant_offset_sd"] * torch.randn(B, 1, N, dtype=x.dtype)
    y = F.softsign(F.linear(x, p["encoder.embedder.embed_spikes.weight"], p["encoder.embedder.embed_spikes.bias"]))
    D = y.shape[-1]
    win = F.unfold(y.unsqueeze(1), kernel_size=(S, D), stride=(st, 1)).transpose(1, 2)            # (B, T', S*D), row-major windows
    Tp = win.shape[1]
    h = F.linear(win, p["encoder.embedder.stack_projection.weight"], p["encoder.embedder.stack_projection.bias"])
    m = batch["spikes_mask"].to(x.dtype)
    tmask = F.unfold(m.view(B, 1, T, 1), kernel_size=(S, 1), stride=(st, 1)).prod(1) > 0.5        # all S source bins valid
    ts = batch["spikes_timestamp"][:, :Tp]                                                       # first T' stamps (ndt1.py:181)
    h = h + F.embedding(ts, p["encoder.embedder.embed_pos.weight"])
    h = F.dropout(h, hp["embed_dropout"], train)
    H = h.shape[-1]
    attn = torch.eye(Tp, dtype=torch.bool).unsqueeze(0) | tmask.unsqueeze(1)                      # eye | (ctx & key valid), ctx = all
    attn = attn.unsqueeze(1).expand(B, nh, Tp, Tp)
    for l in range(L):
        pre = f"encoder.layers.{l}."
        a = F.layer_norm(h, (H,), p[pre + "ln1.weight"], p[pre + "ln1.bias"])
        q, kk, v = (F.linear(a, p[pre + f"attn.{n}.weight"], p[pre + f"attn.{n}.bias"]).view(B, Tp, nh, H // nh).transpose(1, 2)
                    for n in ("query", "key", "value"))
        o = F.scaled_dot_product_attention(q, kk, v, attn_mask=attn, dropout_p=hp["dropout"] if train else 0.0, is_causal=False)
        o = F.dropout(o.transpose(1, 2).reshape(B, Tp, H), hp["dropout"], train)
        h = h + F.linear(o, p[pre + "attn.out_proj.weight"], p[pre + "attn.out_proj.bias"])
        a = F.layer_norm(h, (H,), p[pre + "ln2.weight"], p[pre + "ln2.bias"])
        u = F.gelu(F.linear(a, p[pre + "mlp.up_proj.weight"], p[pre + "mlp.up_proj.bias"]))
        h = h + F.dropout(F.linear(u, p[pre + "mlp.down_proj.weight"], p[pre + "mlp.down_proj.bias"]), hp["dropout"], train)
    h = F.layer_norm(h, (H,), p["encoder.out_norm.weight"], p["encoder.out_norm.bias"])
    lp = F.log_softmax(F.linear(h, p["decoder.0.weight"], p["decoder.0.bias"]), -1)
    lens = (1 + (batch["spikes_lengths"].reshape(-1) - S) / st).to(torch.int64)                   # float divide, truncating cast
    loss = F.ctc_loss(lp.transpose(0, 1), batch["targets"], lens, batch["targets_lengths"].reshape(-1), blank=hp["blank_id"],
                      reduction="none", zero_infinity=True).sum()
    return loss, lp, lens


class TorchCpuTrainer:
    """AdamW over ALL parameters + per-step OneCycle (cosine, cycle_momentum) exactly as trainer.py:229,240-246 configures them."""

    def __init__(self, params, hp=None, lr=1e-3, wd=5e-5, eps=1e-8, total_steps=100, pct_start=0.0, div_factor=25.0):
        self.hp = dict(default_hparams(), **(hp or {}))
        self.p = {k: torch.nn.Parameter(torch.as_tensor(v).clone().float()) for k, v in params.items()}
        self.opt = torch.optim.AdamW(list(self.p.values()), lr=lr, weight_decay=wd, eps=eps)
        self.sched = torch.optim.lr_scheduler.OneCycleLR(self.opt, total_steps=total_steps, max_lr=lr, pct_start=pct_start,
                                                         div_factor=div_factor)

    def step(self, batch, train=True):
        loss, lp, _ = forward(self.p, batch, self.hp, train)
        loss.backward()
        self.opt.step()
        self.sched.step()
        self.opt.zero_grad()
        return loss.detach(), lp.detach()


def host_cpu_description():
    """{'model', 'sockets', 'physical_cores', 'logical_cpus'} from /proc/cpuinfo (printed beside the baseline)."""
    model, phys, cores_per, logical = "unknown", set(), {}, 0
    try:
        pid = None
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                continue
            k, v = [s.strip() for s in line.split(":", 1)]
            if k == "processor":
                logical += 1
            elif k == "model name":
                model = v
            elif k == "physical id":
                pid = v
                phys.add(v)
            elif k == "cpu cores" and pid is not None:
                cores_per[pid] = int(v)
    except OSError:
        pass
    physical = sum(cores_per.values()) if cores_per else logical
    return {"model": model, "sockets": max(1, len(phys)), "physical_cores": physical or logical, "logical_cpus": logical}
