"""In-box A/B of library BUILDS (two commits, or a commit and a measurement build): bench.py once per (library, configuration), the libraries
alternating, `--rounds` times; prints every run's median ms / step and the median over rounds. One process per run (the library is loaded once per
process: NBCI_LIB), all on the same GPU back to back - never compare numbers of different boxes.

    python tools/ab_libs.py base=build/base/libnbci.so new=llm_bci_amd/csrc/libnbci.so [--rounds 3] [--configs b64_bf16 b64_f32 b8_bf16]
A variant may carry environment switches of a measurement build: name=path:KEY=VAL,KEY2=VAL2.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {"b64_bf16": ["--batch", "64", "--residual-dtype", "bf16"], "b64_f32": ["--batch", "64", "--residual-dtype", "fp32"],
           "b8_bf16": ["--batch", "8", "--residual-dtype", "bf16"], "b64_bf16_1s": ["--batch", "64", "--residual-dtype", "bf16", "--side-stream", "off"],
           "b8_f32": ["--batch", "8", "--residual-dtype", "fp32"], "b32_bf16": ["--batch", "32", "--residual-dtype", "bf16"],
           "b16_bf16": ["--batch", "16", "--residual-dtype", "bf16"], "b4_bf16": ["--batch", "4", "--residual-dtype", "bf16"]}

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+", help="name=path")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--configs", nargs="*", default=["b64_bf16", "b64_f32", "b8_bf16"])
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
libs, envs = [], {}
for x in a.libs:
    n, rest = x.split("=", 1)
    p, _, ev = rest.partition(":")
    libs.append([n, p])
    envs[n] = dict(kv.split("=", 1) for kv in ev.split(",") if kv)
res = {(n, c): [] for n, _ in libs for c in a.configs}
for r in range(a.rounds):
    for c in a.configs:
        for n, p in libs:
            env = dict(os.environ, NBCI_LIB=os.path.join(ROOT, p), **envs[n])
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra-points", "--no-roofline", "--steps",
                                  str(a.steps)] + CONFIGS[c], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(f"round {r} {c} {n}: FAILED\n{out.stderr[-600:]}", flush=True)
                continue
            d = json.loads(line[-1])
            res[(n, c)].append(d["ms_per_step"])
            print(f"round {r} {c:12s} {n:8s} {d['ms_per_step']:.3f} ms/step (min {d['ms_per_step_min']:.3f} max {d['ms_per_step_max']:.3f})", flush=True)
print()
for c in a.configs:
    base = None
    for n, _ in libs:
        v = res[(n, c)]
        if not v:
            continue
        med = statistics.median(v)
        base = base or med
        print(f"{c:12s} {n:8s} median {med:.3f} ms/step  runs {['%.3f' % x for x in v]}  vs {libs[0][0]} {med / base:.4f}")
