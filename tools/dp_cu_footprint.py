"""What an overlapped RCCL all-reduce does to the train step on ONE GPU: RCCL's channels park workgroups on
some CUs for the whole exchange, and every hot GEMM grid of the step is sized for exactly one or two rounds of 256 CUs. This
emulates that footprint: a kernel that parks k workgroups (one per CU: each claims the CU's whole LDS, or a light 8 KB footprint
that leaves the CU shareable) on a second stream for the duration of each step, and measures the step beside it - with the tile
cost model told nothing (available_cus = 256) and told the truth (nbci_set_available_cus(256 - k)).

    python tools/dp_cu_footprint.py            # B = 64, bf16, 10 steps per point
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch  # noqa: E402
from llm_bci_amd._lib import check, lib  # noqa: E402
from llm_bci_amd.ndt1 import NDT1  # noqa: E402
from llm_bci_amd.trainer import NativeTrainer  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1)
m = NDT1({}, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True, compute_dtype="bf16").to(dev)
tr = NativeTrainer(m, total_steps=2000)
_, batch = make_batch(64, 600, 256, 60, 41, dev, 0)
l = lib()
side = torch.cuda.Stream(device=dev)
STEPS = 10


def run(k, lds, avail):
    check(l.nbci_set_available_cus(avail), "available_cus")
    for i in range(3):
        tr.train_step(batch, seed=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        if k:
            side.wait_stream(torch.cuda.current_stream())     # parked from the start of the step to (about) its end
            check(l.nbci_debug_occupy_cus(k, lds, 4600.0, C.c_void_p(side.cuda_stream)), "occupy")
        tr.train_step(batch, seed=10 + i)
        if k:
            torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / STEPS


res = {"note": "ms per train step (B = 64, bf16) beside k parked workgroups; exclusive = 160 KB of LDS each (the CU is lost), "
               "light = 8 KB (the CU stays shareable, the parked waves only sleep)", "points": []}
base = run(0, 0, 256)
res["points"].append({"parked": 0, "ms_per_step": round(base, 3)})
for k in (8, 16, 32):
    for kind, lds in (("exclusive", 160 * 1024), ("light", 8 * 1024)):
        a = run(k, lds, 256)
        b = run(k, lds, 256 - k) if kind == "exclusive" else None
        res["points"].append({"parked": k, "footprint": kind, "ms_per_step_model_unaware": round(a, 3),
                              "ms_per_step_model_told": None if b is None else round(b, 3), "vs_alone": round(a / base, 3)})
        print(res["points"][-1], flush=True)
check(l.nbci_set_available_cus(256), "available_cus")
print(json.dumps(res))
