#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_flash.sh OUT_DIR NS nh S H p  - two rocprofv3 counter passes over tools/bench_flash.py (own runs, --kernel-trace only)
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$out/p1 -- python3 $R/tools/bench_flash.py "$@" > $R/$out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $R/$out/p2 -- python3 $R/tools/bench_flash.py "$@" > $R/$out/p2.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void nbci::", "")[:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "fattn" not in k: continue
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
