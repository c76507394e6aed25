"""usage: python tools/per_kernel_print.py bench_a.json [bench_b.json ...] — the roofline.per_kernel rows of bench.py lines, one table per file"""
import json
import sys

for f in sys.argv[1:]:
    d = json.load(open(f))
    print(f, "ms_per_step", d["ms_per_step"])
    for r in d["roofline"]["per_kernel"]:
        print("  %-72s n=%5.1f avg %7.1f us  %8.3f ms/step" % (r["kernel"][:72], r["launches_per_step"], r["avg_launch_us"], r["ms_per_step"]))
