"""Summarise one train step from a rocprofv3 --kernel-trace CSV (per-kernel totals)."""
import collections
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if "per_kernel" in r["Kernel_Name"]]
mid = len(idx) // 2   # a step from the middle of the run (the last ones may belong to bench.py's event-instrumented pass)
step = rows[idx[mid - 1] + 1: idx[mid] + 1]
agg, tot = collections.OrderedDict(), 0.0
for r in step:
    nm = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    short = nm.replace("nbci::", "").replace("void ", "").split("(")[0][:60]
    if "gemm" in nm:
        short += " g=%sx%s" % (int(r["Grid_Size_X"]) // 256, r["Grid_Size_Y"])
    a = agg.setdefault(short, [0, 0.0]); a[0] += 1; a[1] += d; tot += d
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-84s n=%3d tot=%8.1f us avg=%7.1f" % (k, v[0], v[1], v[1] / v[0]))
print("sum kernel time per step (us): %.1f  span: %.1f" % (tot, (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3))
