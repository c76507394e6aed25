"""Accuracy of the streaming attention kernels against an f32 torch evaluation of the same bf16 operands (no dropout), forward and backward.
python tools/check_flash.py NS nh S H     (NBCI_LIB / NBCI_FA_* select the library and, in measurement builds, the kernel family)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd._lib import check, lib  # noqa: E402

NS, nh, S, H = (int(x) for x in sys.argv[1:5])
SCALE = float(sys.argv[5]) if len(sys.argv) > 5 else 0.7   # standard deviation of q, k, v
MEAN = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
hd = H // nh
dev = "cuda"
torch.manual_seed(0)
qkv = (torch.randn(NS * S, 3 * H, device=dev) * SCALE + MEAN).bfloat16()
dout = torch.randn(NS * S, H, device=dev).bfloat16()
out = torch.empty(NS * S, H, dtype=torch.bfloat16, device=dev)
lse = torch.empty(NS * nh * S, device=dev); dsum = torch.empty_like(lse); dqkv = torch.zeros_like(qkv)
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
l = lib()
check(l.nbci_attention_flash_fwd(P(qkv), P(out), P(lse), NS, nh, S, H, 0.0, 1, 2, st), "f")
check(l.nbci_attention_flash_bwd(P(qkv), P(out), P(dout), P(lse), P(dsum), P(dqkv), NS, nh, S, H, 0.0, 1, 2, st), "b")
torch.cuda.synchronize()

x = qkv.float().view(NS, S, 3, nh, hd).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)   # (3, NS, nh, S, hd)
q, k, v = x[0], x[1], x[2]
sc = (q @ k.transpose(-1, -2)) / hd ** 0.5
ref_l = torch.logsumexp(sc, -1)
o = torch.softmax(sc, -1) @ v                                     # (NS, nh, S, hd)
o2 = o.permute(0, 2, 1, 3).reshape(NS * S, H)
o2.backward(dout.float())
gref = x.grad.permute(1, 3, 0, 2, 4).reshape(NS * S, 3 * H)


def err(a, b):
    return f"max {float((a - b).abs().max()):.3e}  l1rel {float((a - b).abs().sum() / b.abs().sum()):.3e}"


print(f"NS={NS} nh={nh} S={S} H={H} (head {hd}) std {SCALE} mean {MEAN}: out {err(out.float(), o2.detach())} | lse {err(lse.view(NS, nh, S), ref_l.detach())}")
for i, n in enumerate("qkv"):
    print(f"    d{n}: {err(dqkv.float()[:, i * H:(i + 1) * H], gref[:, i * H:(i + 1) * H])}")
print(f"    finite: out {bool(torch.isfinite(out.float()).all())} dqkv {bool(torch.isfinite(dqkv.float()).all())}", flush=True)
