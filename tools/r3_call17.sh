#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/c17; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/ptst" -o t -- python3 $R/tools/bench_ptst.py --dtype fp8 --steps 5 --warmup 2 > "$out/ptst.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/itr" -o t -- python3 $R/tools/bench_itr.py --channels 1500 --steps 5 --warmup 2 > "$out/itr.log" 2>&1
cd $R
python3 tools/db_stats.py "$(find $out/ptst -name '*.db' | head -1)" > $out/ptst_stats.csv
python3 tools/db_stats.py "$(find $out/itr -name '*.db' | head -1)" > $out/itr_stats.csv
rm -rf $out/ptst $out/itr
head -22 $out/ptst_stats.csv; head -16 $out/itr_stats.csv
