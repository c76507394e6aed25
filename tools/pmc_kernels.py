"""HBM-side bytes per launch of every kernel, by symbol, from two rocprofv3 counter passes (own runs: FETCH_SIZE and WRITE_SIZE do
not fit one pass; each with --kernel-trace only, the program directly after `--`):
    python tools/pmc_kernels.py --fetch <pmc_fetch.db> --write <pmc_write.db> --out profiles/r03_pmc_kernels.json
FETCH_SIZE is doubled (gfx950 tallies a 128-byte request as 64 B: MI355X_MICROARCH.md §HBM), both are KiB -> bytes; the counters
count fabric-side requests (Infinity-Cache hits included). bench.py reads `hbm_bytes_per_launch[symbol]` as roofline.traffic."""
import argparse
import collections
import json
import re
import sqlite3


def symbol(name):
    n = re.sub(r"\(.*", "", name.replace("void ", "").replace("nbci::", "")).strip()
    m = re.match(r"_ZN4nbci(\d+)([A-Za-z_0-9]+)", n)
    return m.group(2)[:int(m.group(1))] if m else n


def table(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall()
    per = collections.defaultdict(float)
    names = {}
    for name, cn, v, did in rows:
        if cn == counter:
            per[did] += float(v)
            names[did] = name
    acc = collections.defaultdict(list)
    for did, v in per.items():
        acc[symbol(names[did])].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


ap = argparse.ArgumentParser()
ap.add_argument("--fetch", required=True); ap.add_argument("--write", required=True); ap.add_argument("--out", required=True)
ap.add_argument("--note", default="")
ap.add_argument("--config", default="", help="JSON of the bench.py configuration the counter passes ran (batch, bins, channels, dtype, residual_dtype, side_stream): "
                "bench.py reports roofline.traffic only for a run of the same configuration")
a = ap.parse_args()
f, w = table(a.fetch, "FETCH_SIZE"), table(a.write, "WRITE_SIZE")
out = {"note": a.note or "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) of bench.py; FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 bytes, mean per launch",
       "config": json.loads(a.config) if a.config else None,
       "hbm_bytes_per_launch": {}, "fetch_bytes_per_launch": {}, "write_bytes_per_launch": {}, "launches_seen": {}}
for k in sorted(set(f) | set(w)):
    fb = f.get(k, (0.0, 0))[0] * 2 * 1024
    wb = w.get(k, (0.0, 0))[0] * 1024
    out["hbm_bytes_per_launch"][k] = round(fb + wb)
    out["fetch_bytes_per_launch"][k] = round(fb)
    out["write_bytes_per_launch"][k] = round(wb)
    out["launches_seen"][k] = max(f.get(k, (0, 0))[1], w.get(k, (0, 0))[1])
json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)
for k, v in sorted(out["hbm_bytes_per_launch"].items(), key=lambda kv: -kv[1])[:25]:
    print(f"{k[:70]:70s} {v / 1e6:9.1f} MB  (fetch {out['fetch_bytes_per_launch'][k] / 1e6:8.1f}  write {out['write_bytes_per_launch'][k] / 1e6:8.1f})  n={out['launches_seen'][k]}")
