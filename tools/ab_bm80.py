"""80-row GEMM tiles (three workgroups per CU) against the cost model's pick, per shape: the measurement build with NBCI_GEMM_BM80=0 / 2 in
child processes, hot loop of 20 launches, bf16 out, bias epilogue.  python tools/ab_bm80.py [child M N K]"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(9152, 1024, 1024), (9152, 3072, 1024), (9152, 1024, 3072), (4576, 1024, 1024), (4576, 3072, 1024), (2288, 1024, 1024), (2288, 3072, 1024),
          (24016, 2304, 768), (24016, 3072, 768), (24016, 768, 3072), (24016, 768, 768), (10704, 2304, 768), (10704, 3072, 768), (10704, 768, 3072),
          (419840, 256, 256), (419840, 1024, 256), (419840, 256, 1024), (419840, 768, 256)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from llm_bci_amd import ops
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") / 32).bfloat16()
        bias = torch.randn(N, device="cuda"); c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        f = lambda: ops.gemm(M, N, K, ops.operand(a, K, True), ops.operand(b, K, True), c, N, in_dtype=1, c_dtype=1, bias=bias)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f"{M} {N} {K} {e0.elapsed_time(e1) / 20 * 1e3:.1f}", flush=True)
        del a, b, c
    sys.exit(0)

res = {}
for mode in ("0", "2"):
    env = dict(os.environ, NBCI_LIB=os.path.join(ROOT, "build/measure/libnbci.so"), NBCI_GEMM_BM80=mode)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    for l in out.stdout.splitlines():
        p = l.split()
        if len(p) == 4:
            res[(tuple(int(x) for x in p[:3]), mode)] = float(p[3])
    if out.returncode:
        print(out.stderr[-800:])
print(f"{'M x N x K':>24s} {'model pick':>11s} {'80-row':>9s}  ratio   TF/s(80)")
for s in SHAPES:
    a, b = res.get((s, "0")), res.get((s, "2"))
    if a and b:
        print(f"{s[0]:>8d} x{s[1]:>5d} x{s[2]:>5d} {a:11.1f} {b:9.1f}  {b / a:5.2f}   {2.0 * s[0] * s[1] * s[2] / b / 1e6:7.0f}")
