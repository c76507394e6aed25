#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/c3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace8" -o t -- python3 $R/bench.py --no-cpu-baseline --no-extra-points --no-roofline --repeats 1 --steps 10 --warmup 3 --batch 8 --side-stream on > "$out/trace8.log" 2>&1 || { tail "$out/trace8.log"; exit 1; }
python3 $R/tools/seq_step.py "$out/trace8" > "$out/b8_side_seq.txt"
rm -rf "$out/trace8"
cd $R
for cfg in "1 512" "1 256" "1 1024" "0 0"; do
  set -- $cfg
  echo "== per-seg adamw $1, blocks $2" >> $out/ab.txt
  NBCI_SIDE_ADAMW=$1 NBCI_SIDE_BLOCKS=$2 timeout -k 10 200 python tools/ab_side_stream.py --batches 8 32 --windows 5 >> $out/ab.txt 2>&1
done
cat $out/ab.txt
