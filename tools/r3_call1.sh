#!/bin/bash
# round-3 first GPU call: this box's baseline + where a B = 8 step goes (launch by launch)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/c1
mkdir -p "$out"
true
true
cd /tmp && export TMPDIR=/tmp
for B in 8 64; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace$B" -o t -- python3 $R/bench.py --no-cpu-baseline --no-extra-points --no-roofline --repeats 1 --steps 10 --warmup 3 --batch $B > "$out/trace$B.log" 2>&1 || { tail "$out/trace$B.log"; exit 1; }
  python3 $R/tools/seq_step.py "$out/trace$B" > "$out/b${B}_seq.txt"
  python3 $R/tools/prof_step.py "$out/trace$B" 60 > "$out/b${B}_breakdown.txt"
  tail -3 "$out/b${B}_breakdown.txt"
  rm -rf "$out/trace$B"
done
cd "$R"
true
