"""Time the streaming attention kernels (HIP events). python tools/bench_flash.py NS nh S H [drop_p]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd._lib import check, lib  # noqa: E402

NS, nh, S, H = (int(x) for x in sys.argv[1:5])
p = float(sys.argv[5]) if len(sys.argv) > 5 else 0.4
dev = "cuda"
qkv = (torch.randn(NS * S, 3 * H, device=dev) * 0.5).bfloat16()
dout = torch.randn(NS * S, H, device=dev).bfloat16()
out = torch.empty(NS * S, H, dtype=torch.bfloat16, device=dev)
lse = torch.empty(NS * nh * S, device=dev); dsum = torch.empty_like(lse); dqkv = torch.empty_like(qkv)
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
l = lib()


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


f = lambda: check(l.nbci_attention_flash_fwd(P(qkv), P(out), P(lse), NS, nh, S, H, p, 1, 2, st), "f")
bw = lambda: check(l.nbci_attention_flash_bwd(P(qkv), P(out), P(dout), P(lse), P(dsum), P(dqkv), NS, nh, S, H, p, 1, 2, st), "b")
tf, tb = timeit(f), timeit(bw)
fl = 4.0 * NS * nh * S * S * (H // nh)
print(f"{os.environ.get('NBCI_LIB', 'default')[-16:]:>16}  NS={NS} nh={nh} S={S} H={H} p={p}: fwd {tf * 1e3:8.1f} us ({fl / tf / 1e9:6.1f} TF/s)   bwd(q+kv) {tb * 1e3:8.1f} us ({2.5 * fl / tb / 1e9:6.1f} TF/s)", flush=True)
