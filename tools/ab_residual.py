"""In-box A/B of NDT1's residual_dtype ("fp32" = f32 residual / gradient streams, "bf16" = both stored in bf16) at C2:
(1) error of each bf16 variant against the fp32 HIP path on the same weights, inputs and dropout / noise draws: eval log-probs
    (max abs), train-mode loss, gradient L1 ratio per parameter (worst and total);
(2) train-step time, modes interleaved in one process, median of windows.
python tools/ab_residual.py [--batches 8 64] [--steps 20] [--windows 7]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_batch  # noqa: E402
from llm_bci_amd.ndt1 import NDT1  # noqa: E402
from llm_bci_amd.trainer import NativeTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, nargs="*", default=[8, 64])
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--windows", type=int, default=7)
ap.add_argument("--no-error", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)


def build(dtype, residual):
    torch.manual_seed(1)
    return NDT1({}, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype=dtype, residual_dtype=residual).to(dev)


def fwd_bwd(m, b, train, seed=7):
    m.train(train)
    loss, preds = m._run_forward(b, want_grad=True, seed=seed)
    g = torch.zeros_like(m._flat)
    m._run_backward(g)
    torch.cuda.synchronize()
    return loss.double().cpu(), preds.double().cpu(), g.double().cpu()


if not args.no_error:
    ref = build("fp32", "fp32")
    _, b = make_batch(16, 600, 256, 60, 41, dev, seed=0)
    out = {}
    for name, (dt, rd) in {"fp32": ("fp32", "fp32"), "bf16 / f32 streams": ("bf16", "fp32"), "bf16 / bf16 streams": ("bf16", "bf16")}.items():
        m = ref if name == "fp32" else build(dt, rd)
        if m is not ref:
            m.load_state_dict(ref.state_dict())
        out[name] = (fwd_bwd(m, b, False), fwd_bwd(m, b, True))
    (l0e, p0e, g0e), (l0t, p0t, g0t) = out["fp32"]
    lens = b["spikes_lengths"].cpu()
    tl = (1 + (lens - 32) // 4).clamp(min=0)   # valid tokens per sample (stack 32 / stride 4)
    valid = torch.arange(p0e.shape[1])[None, :] < tl[:, None]
    for name in list(out)[1:]:
        (le, pe, ge), (lt, pt, gt) = out[name]
        dp = (pe - p0e).abs()[valid].max().item()
        flips = (pe.argmax(-1) != p0e.argmax(-1))[valid].sum().item()
        worst, tot_n, tot_d = 0.0, 0.0, 0.0
        for (pn, off, numel, shape, _s) in ref._layout:
            a, r = gt[off:off + numel], g0t[off:off + numel]
            n, d = (a - r).abs().sum().item(), r.abs().sum().item()
            tot_n += n; tot_d += d
            if d > 0:
                worst = max(worst, n / d)
        print(f"{name:22s} eval log-probs max |d| {dp:.4f}  argmax flips {flips}/{int(valid.sum())}  eval loss rel {((le - l0e).abs().sum() / l0e.abs().sum()).item():.2e}"
              f"  train loss rel {((lt - l0t).abs().sum() / l0t.abs().sum()).item():.2e}  grad L1 ratio total {tot_n / tot_d:.4f} worst param {worst:.4f}", flush=True)
    del ref, out

models = {rd: build("bf16", rd) for rd in ("fp32", "bf16")}
trs = {rd: NativeTrainer(m, total_steps=1_000_000) for rd, m in models.items()}
for B in args.batches:
    _, b = make_batch(B, 600, 256, 60, 41, dev, seed=0)
    res = {rd: [] for rd in trs}
    for rd, tr in trs.items():
        for i in range(3):
            tr.train_step(b, seed=i)
    for w in range(args.windows):
        for rd, tr in trs.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                tr.train_step(b, seed=100 * w + i)
            torch.cuda.synchronize()
            res[rd].append((time.perf_counter() - t0) / args.steps * 1e3)
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    print(f"B={B:3d}  f32 streams {med['fp32']:.3f} ms (min {min(res['fp32']):.3f})   bf16 streams {med['bf16']:.3f} ms (min {min(res['bf16']):.3f})"
          f"   ratio {med['bf16'] / med['fp32']:.3f}", flush=True)
