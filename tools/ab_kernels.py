"""Per-kernel A/B of two library builds in one box: bench.py's HIP-event per-kernel table (one stream) once per library, printed side by side.
    python tools/ab_kernels.py base=build/base/libnbci.so new=llm_bci_amd/csrc/libnbci.so [bench args...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [x.split("=", 1) for x in sys.argv[1:3]]
extra = sys.argv[3:]
tabs = {}
for n, p in libs:
    env = dict(os.environ, NBCI_LIB=os.path.join(ROOT, p))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra-points", "--steps", "20"] + extra, env=env,
                         capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(n, "FAILED", out.stderr[-500:]); sys.exit(1)
    d = json.loads(line[-1])
    tabs[n] = ({r["kernel"]: r for r in d["roofline"]["per_kernel"]}, d["ms_per_step"])
(a, ams), (b, bms) = tabs[libs[0][0]], tabs[libs[1][0]]
print(f"step: {libs[0][0]} {ams:.3f} ms   {libs[1][0]} {bms:.3f} ms")
print(f"{'kernel':72s} {'n/step':>6s} {libs[0][0]:>9s} {libs[1][0]:>9s} {'d us/step':>10s}")
tot = 0.0
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k) or b.get(k))["ms_per_step"]):
    ra, rb = a.get(k), b.get(k)
    n = (ra or rb)["launches_per_step"]
    ua = ra["avg_launch_us"] if ra else float("nan")
    ub = rb["avg_launch_us"] if rb else float("nan")
    dd = ((rb["ms_per_step"] if rb else 0) - (ra["ms_per_step"] if ra else 0)) * 1e3
    tot += dd
    print(f"{k[:72]:72s} {n:6.1f} {ua:9.2f} {ub:9.2f} {dd:10.1f}")
print(f"sum of differences: {tot:.1f} us/step")
