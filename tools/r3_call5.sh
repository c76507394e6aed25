#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/c5; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_attention_f64_gpu.py -q -x -s > $out/pytest_attn.log 2>&1; tail -12 $out/pytest_attn.log
timeout -k 10 900 python -m pytest tests/test_ndt1_gpu.py -q -x > $out/pytest_ndt1.log 2>&1; tail -4 $out/pytest_ndt1.log
timeout -k 10 200 python tools/ab_side_stream.py --batches 8 64 --windows 5 > $out/ab.txt 2>&1; cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
for B in 8 64; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace$B" -o t -- python3 $R/bench.py --no-cpu-baseline --no-extra-points --no-roofline --repeats 1 --steps 10 --warmup 3 --batch $B --side-stream off > "$out/trace$B.log" 2>&1 || { tail "$out/trace$B.log"; exit 1; }
  python3 $R/tools/prof_step.py "$out/trace$B" 60 > "$out/b${B}_breakdown.txt"
  rm -rf "$out/trace$B"
done
head -16 $out/b64_breakdown.txt
