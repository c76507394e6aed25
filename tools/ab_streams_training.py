"""Does storing the residual / gradient streams in bf16 change how the NDT1-CTC step TRAINS? The same model, batches, seeds and schedule with
NDT1(residual_dtype="fp32") and ("bf16"): loss per sample and on-device PER at a few points of the run (four fixed synthetic batches, so
the loss falls as they are memorised).   python tools/ab_streams_training.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from llm_bci_amd.ndt1 import NDT1  # noqa: E402
from llm_bci_amd.trainer import NativeTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # independent repetitions (model init + dropout / noise draws)
dev = torch.device("cuda", 0)
marks = sorted({0, steps // 10, steps // 4, steps // 2, 3 * steps // 4, steps - 1})
cfgs = ((8, 600, True), (5, 332, True), (64, 600, True)) if nseeds == 1 else ((8, 600, True),)
for B, bins, ragged in [c for c in cfgs for _ in range(nseeds)]:
    rows = {}
    rep_id = getattr(sys.modules[__name__], "_rep", 0)
    sys.modules[__name__]._rep = rep_id + 1
    for rd in ("fp32", "bf16"):
        torch.manual_seed(1 + rep_id)
        model = NDT1({"encoder": {"embedder": {"n_channels": 256}}}, method_name="ctc", vocab_size=41, blank_id=0, zero_infinity=True,
                     compute_dtype="bf16", residual_dtype=rd).to(dev)
        tr = NativeTrainer(model, lr=1e-3, wd=5e-5, eps=1e-8, scheduler="cosine", total_steps=steps + 8, warmup_pct=0.0, div_factor=25)
        batches = [bench.make_batch(B, bins, 256, 60, 41, dev, seed=s, ragged=ragged)[1] for s in range(4)]
        out = []
        for i in range(steps):
            loss, _ = tr.train_step(batches[i % 4], seed=i + 100003 * rep_id)
            if i in marks:
                out.append(float(loss.sum().item()) / B)
        torch.cuda.synchronize()
        rows[rd] = (out, tr.read_stats())
    print(f"B={B} bins={bins} ragged={ragged} repetition {rep_id}: loss per sample at steps {marks}")
    for rd, (out, stt) in rows.items():
        print(f"   {rd:4s} streams: " + "  ".join(f"{v:8.2f}" for v in out) + f"    running PER {stt['PER']:.3f}", flush=True)
