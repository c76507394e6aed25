"""Which tensors differ between two gradient dumps of tools/itr_grad_errors.py (python tools/diff_grads.py a.npz b.npz): L1-relative difference,
ratio of the norms, cosine"""
import sys

import numpy as np

a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
rows = []
for k in a.files:
    x, y = a[k].astype(np.float64).ravel(), b[k].astype(np.float64).ravel()
    d = np.abs(x - y).sum() / (np.abs(y).sum() + 1e-300)
    rows.append((d, k, np.linalg.norm(x) / (np.linalg.norm(y) + 1e-300), float(x @ y) / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-300)))
for d, k, r, c in sorted(rows, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{d:.3e}  |a|/|b| {r:.4f}  cos {c:.5f}  {k}")
print("identical:", sum(1 for d, *_ in rows if d == 0), "of", len(rows))
