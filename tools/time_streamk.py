"""A/B of nbci_gemm_grouped's two launch schemes (one workgroup per output tile vs K tiles dealt out evenly, gemm_streamk.hip) on the
weight-gradient groups of the three models' train steps. Usage: python tools/time_streamk.py [--iters 50]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_bci_amd import ops                      # noqa: E402
from llm_bci_amd._lib import GemmDesc, check, lib   # noqa: E402

GROUPS = {
    "NDT1 layer (B=64: 9152 tokens)": [(3072, 1024, 9152), (1024, 1024, 9152), (1024, 1024, 9152), (1024, 1024, 9152)],
    "NDT1 layer (B=128: 18304 tokens)": [(3072, 1024, 18304), (1024, 1024, 18304), (1024, 1024, 18304), (1024, 1024, 18304)],
    "iTransformer layer (768 wide, 16 x 1501 tokens = 24016 rows)": [(2304, 768, 24016), (768, 768, 24016), (2048, 768, 24016), (768, 2048, 24016)],
    "two square problems (2 x 64 tiles)": [(1024, 1024, 9152), (1024, 1024, 9152)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    l = lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, probs in GROUPS.items():
        descs = (GemmDesc * len(probs))()
        keep = []
        fl = 0.0
        for i, (M, N, K) in enumerate(probs):
            ad = torch.randn(K, M, device="cuda").to(torch.bfloat16)
            bd = torch.randn(K, N, device="cuda").to(torch.bfloat16)
            out = torch.zeros(M, N, device="cuda")
            d = descs[i]
            d.M, d.N, d.K, d.in_dtype = M, N, K, ops.NBCI_BF16
            d.A, d.B = ops.operand(ad, M, False), ops.operand(bd, N, False)
            d.C, d.ldc, d.c_dtype, d.batch, d.zdiv, d.splitk, d.alpha, d.beta = out.data_ptr(), N, ops.NBCI_F32, 1, 1, 1, 1.0, 1.0
            keep += [ad, bd, out]
            fl += 2.0 * M * N * K
        tiles = sum(((M + 127) // 128) * ((N + 127) // 128) for M, N, K in probs)
        res = {}
        for mode in (0, 2):
            check(l.nbci_debug_gemm_streamk(mode), "mode")
            for _ in range(5):
                check(l.nbci_gemm_grouped(descs, len(probs), st), "grouped")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                check(l.nbci_gemm_grouped(descs, len(probs), st), "grouped")
            e1.record()
            torch.cuda.synchronize()
            res[mode] = e0.elapsed_time(e1) / a.iters * 1e3
        check(l.nbci_debug_gemm_streamk(1), "mode")
        print(f"{name:64s} {tiles:4d} tiles  per-tile {res[0]:7.1f} us ({fl / res[0] / 1e6:6.1f} TF)   dealt-out {res[2]:7.1f} us ({fl / res[2] / 1e6:6.1f} TF)"
              f"   ratio {res[2] / res[0]:.3f}", flush=True)


if __name__ == "__main__":
    main()
