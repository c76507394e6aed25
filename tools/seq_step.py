"""List one train step launch by launch from a rocprofv3 --kernel-trace CSV (start offset, name, grid, duration)."""
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "per_kernel" in r["Kernel_Name"]]
mid = len(idx) // 2   # a step from the middle of the run (the last ones may belong to bench.py's event-instrumented pass)
step = rows[idx[mid - 1] + 1: idx[mid] + 1]
t0 = int(step[0]["Start_Timestamp"])
for r in step:
    nm = r["Kernel_Name"].replace("nbci::", "").replace("void ", "").split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f -> %9.1f  q=%-3s %-62s g=%5dx%-3s %8.1f us" % ((s - t0) / 1e3, (e - t0) / 1e3, r.get("Queue_Id", "?"), nm,
          int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], (e - s) / 1e3))
print("span %.1f us" % ((int(step[-1]["End_Timestamp"]) - t0) / 1e3))
