"""LayerNorm forward / backward alone, through the C-ABI (nbci_layernorm_fwd_ex / _bwd_ex), at the train step's shapes: the stream dtypes
and cast outputs of nbci_ndt1_config.residual_dtype = f32 / bf16. Operands rotate through `--sets` buffer sets so a launch does not find its
inputs in the Infinity Cache (in the step they were written hundreds of MB earlier).   python tools/time_ln.py [--rows 9152 1144] [--sets 6]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llm_bci_amd._lib import check, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="*", default=[9152, 1144])
ap.add_argument("--hidden", type=int, default=1024)
ap.add_argument("--sets", type=int, default=6)
ap.add_argument("--reps", type=int, default=60)
args = ap.parse_args()
dev = torch.device("cuda", 0)
l = lib()
H = args.hidden


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us per launch (back-to-back launches: includes the ~2 us kernel boundary)


for M in args.rows:
    S = args.sets
    w = torch.randn(H, device=dev); b = torch.randn(H, device=dev)
    mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev)
    dw = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev); cs = torch.zeros(H, device=dev)
    x32 = [torch.randn(M, H, device=dev) for _ in range(S)]
    x16 = [t.bfloat16() for t in x32]
    dy16 = [torch.randn(M, H, device=dev).bfloat16() for _ in range(S)]
    dx32 = [torch.randn(M, H, device=dev) for _ in range(S)]
    dx16 = [t.bfloat16() for t in dx32]
    o16 = [torch.empty(M, H, device=dev, dtype=torch.bfloat16) for _ in range(S)]
    c16 = [torch.empty(M, H, device=dev, dtype=torch.bfloat16) for _ in range(S)]
    rows = []

    def fwd(xs, xdt):
        def f(i):
            k = i % S
            check(l.nbci_layernorm_fwd_ex(vp(xs[k]), xdt, vp(w), vp(b), vp(o16[k]), 1, vp(mean), vp(rstd), M, H, st()), "ln_fwd_ex")
        return f

    def bwd(xs, xdt, dxin, dxout, cast, p):
        def f(i):
            k = i % S
            check(l.nbci_layernorm_bwd_ex(vp(dy16[k]), 1, vp(xs[k]), xdt, vp(w), vp(mean), vp(rstd), vp(dxin[k]) if dxin else None, vp(dxout[k]),
                                          vp(dw), vp(db), M, H, vp(cast[k]) if cast else None, 1, p, 7, 18, vp(cs), st()), "ln_bwd_ex")
        return f

    mb = M * H / 1e6
    cases = [
        ("fwd  x f32 -> y bf16", fwd(x32, 0), 6 * mb),
        ("fwd  x bf16 -> y bf16", fwd(x16, 1), 4 * mb),
        ("bwd  f32 streams, in-place dx, masked bf16 cast (step: ln1 / head)", bwd(x32, 0, dx32, dx32, c16, 0.4), 16 * mb),
        ("bwd  f32 streams, in-place dx, plain bf16 cast (step: ln2)", bwd(x32, 0, dx32, dx32, c16, 0.0), 16 * mb),
        ("bwd  bf16 streams, dx_in -> dx_out, masked bf16 cast (ln1)", bwd(x16, 1, dx16, o16, c16, 0.4), 10 * mb),
        ("bwd  bf16 streams, dx_in -> dx_out, column sums only (ln2)", bwd(x16, 1, dx16, o16, None, 0.0), 8 * mb),
        ("bwd  bf16 streams, no dx_in, masked cast (head)", bwd(x16, 1, None, o16, c16, 0.4), 8 * mb),
    ]
    for name, fn, mbytes in cases:
        us = timed(fn, args.reps)
        print(f"M={M:5d} {name:72s} {us:7.1f} us  {mbytes:6.1f} MB  {mbytes / us:5.2f} TB/s", flush=True)
