"""GPU: the two score-tensor-free attention kernels (attn_small.hip: thread-per-query, f32 / bf16; attn_flash.hip:
wave-per-16-queries on MFMA, bf16) against a plain PyTorch fp32 reference of the same op, forward and backward, with
and without probability dropout (masks from the oracle's mirror of the counter RNG)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import rng as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(qkv, NS, nh, S, H, keep):
    hd = H // nh
    q, k, v = (t.reshape(NS, S, nh, hd).transpose(1, 2) for t in qkv.float().split(H, dim=1))
    s = (q @ k.transpose(-1, -2)) / hd ** 0.5
    p = torch.softmax(s, -1) * keep
    return (p @ v).transpose(1, 2).reshape(NS * S, H)


def _run(kind, dtype, NS, nh, S, H, drop_p, seed=5, site=9):
    from llm_bci_amd._lib import NBCI_BF16, NBCI_F32, check, lib
    g = torch.Generator(device="cpu").manual_seed(3)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    qkv = (torch.randn(NS * S, 3 * H, generator=g) * 0.7).to(td).to(DEV)
    dout = torch.randn(NS * S, H, generator=g).to(td).to(DEV)
    out = torch.empty(NS * S, H, dtype=td, device=DEV)
    lse = torch.empty(NS * nh * S, dtype=torch.float32, device=DEV)
    dsum = torch.empty_like(lse)
    dqkv = torch.zeros_like(qkv)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    l = lib()
    if kind == "small":
        dt = NBCI_BF16 if dtype == "bf16" else NBCI_F32
        check(l.nbci_attention_small_fwd(P(qkv), P(out), P(lse), dt, NS, nh, S, H, drop_p, seed, site, st), "small fwd")
        check(l.nbci_attention_small_bwd(P(qkv), P(out), P(dout), P(lse), P(dsum), P(dqkv), dt, NS, nh, S, H, drop_p, seed, site, st), "small bwd")
    else:
        check(l.nbci_attention_flash_fwd(P(qkv), P(out), P(lse), NS, nh, S, H, drop_p, seed, site, st), "flash fwd")
        check(l.nbci_attention_flash_bwd(P(qkv), P(out), P(dout), P(lse), P(dsum), P(dqkv), NS, nh, S, H, drop_p, seed, site, st), "flash bwd")
    torch.cuda.synchronize()
    keep = torch.from_numpy(R.keep_mask(seed, site, NS * nh * S * S, drop_p).reshape(NS, nh, S, S)).to(DEV)
    x = qkv.float().clone().requires_grad_(True)
    ref = _ref(x, NS, nh, S, H, keep)
    ref.backward(dout.float())
    return out.float(), ref.detach(), dqkv.float(), x.grad, lse


@pytest.mark.parametrize("kind,dtype,NS,nh,S,H", [
    ("small", "fp32", 3, 2, 37, 32), ("small", "fp32", 2, 4, 205, 128), ("small", "bf16", 5, 8, 205, 256), ("small", "bf16", 2, 2, 70, 128),
    ("flash", "bf16", 5, 8, 205, 256), ("flash", "bf16", 2, 8, 333, 768), ("flash", "bf16", 1, 2, 1501, 192), ("flash", "bf16", 3, 2, 16, 256),
    ("flash", "bf16", 2, 3, 97, 192)])
@pytest.mark.parametrize("drop_p", [0.0, 0.3])
def test_streaming_attention_matches_torch_reference(kind, dtype, NS, nh, S, H, drop_p):
    out, ref, dqkv, gref, lse = _run(kind, dtype, NS, nh, S, H, drop_p)
    if dtype == "fp32":
        np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), atol=2e-5)
        np.testing.assert_allclose(dqkv.cpu().numpy(), gref.cpu().numpy(), atol=2e-5 + 1e-4 * gref.abs().max().item())
    else:
        assert (out - ref).abs().max().item() < 0.03 * max(1.0, ref.abs().max().item())
        for i, nm in enumerate("qkv"):
            a, b = dqkv[:, i * H:(i + 1) * H], gref[:, i * H:(i + 1) * H]
            rel = (a - b).abs().sum().item() / (b.abs().sum().item() + 1e-9)
            assert rel < 0.03, (nm, rel)
    assert torch.isfinite(lse).all()
