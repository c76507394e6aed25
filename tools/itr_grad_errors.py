"""Per-tensor bf16-vs-fp32 gradient error of an iTransformer fixture (what tests/test_itr_gpu.py bounds by its worst tensor).
python tools/itr_grad_errors.py g_itr_uni_c3 [bf16|fp32 streams]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_itr_gpu import _dev, _grads_of, _model  # noqa: E402
from test_oracle_itr_golden import itr_batch, load  # noqa: E402

name = sys.argv[1]
streams = sys.argv[2] if len(sys.argv) > 2 else "bf16"
fx = load(name)
batch = _dev(itr_batch(fx))
res = {}
for dt in ("fp32", "bf16"):
    m = _model(fx, dtype=dt, **({"residual_dtype": streams} if dt == "bf16" else {})).to("cuda")
    m.mask_override = torch.from_numpy(fx["raw_mask_step0"])
    loss, preds, res[dt] = _grads_of(m, batch)
    res[dt]["__preds"] = preds.float().cpu().numpy()
    res[dt]["__loss"] = np.asarray(loss.float().cpu().numpy())
rows = sorted(((np.abs(res["bf16"][k] - res["fp32"][k]).sum() / (np.abs(res["fp32"][k]).sum() + 1e-6), k, res["fp32"][k].size,
               float(np.abs(res["fp32"][k]).sum())) for k in res["fp32"]), reverse=True)
for r in rows[:8]:
    print(f"{r[0]:.4f}  {r[1]}  n={r[2]}  sum|g|={r[3]:.3e}")
if len(sys.argv) > 3:   # save both gradient sets for an offline diff of two runs
    np.savez(sys.argv[3], **{"bf16:" + k: v for k, v in res["bf16"].items()}, **{"fp32:" + k: v for k, v in res["fp32"].items()})
