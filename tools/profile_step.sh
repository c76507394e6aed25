#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_step.sh <outdir under gpurun_out/>
# rocprofv3 of bench.py: (1) kernel trace + stats of the DEFAULT command (side stream on: what the timed windows run), (2) the same with
# --side-stream off (every kernel alone on the chip: what bench.py's own per-kernel HIP-event pass measures; their averages must agree),
# (3) B = 8 (the reference recipe's per-rank batch on 8 GPUs) with span vs kernel sum, (4) two counter passes (FETCH_SIZE, WRITE_SIZE: they do not
# fit one) -> per-kernel HBM bytes. Summaries only are kept.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-extra-points --no-roofline --repeats 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o t -- python3 $B --steps 20 --warmup 3 > "$out/trace.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace1" -o t -- python3 $B --steps 20 --warmup 3 --side-stream off > "$out/trace1.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/csv64" -o t -- python3 $B --steps 10 --warmup 3 > "$out/csv64.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/csv8" -o t -- python3 $B --steps 10 --warmup 3 --batch 8 > "$out/csv8.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/csv8s" -o t -- python3 $B --steps 10 --warmup 3 --batch 8 --side-stream off > "$out/csv8s.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o t -- python3 $B --steps 4 --warmup 1 --side-stream off > "$out/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o t -- python3 $B --steps 4 --warmup 1 --side-stream off > "$out/pmc_write.log" 2>&1
cd "$R"
db() { find "$out/$1" -name '*.db' | head -1; }
python3 tools/db_stats.py "$(db trace)" > "$out/kernel_stats.csv"
python3 tools/db_stats.py "$(db trace1)" > "$out/kernel_stats_one_stream.csv"
python3 tools/pmc_kernels.py --fetch "$(db pmc_fetch)" --write "$(db pmc_write)" --out "$out/pmc_kernels.json" \
  --config '{"batch": 64, "bins": 600, "channels": 256, "dtype": "bf16", "residual_dtype": "bf16", "side_stream": "off"}' > "$out/pmc_kernels.txt"
for v in 64 8 8s; do
  python3 tools/seq_step.py "$out/csv$v" > "$out/b${v}_step_sequence.txt"
  python3 tools/prof_step.py "$out/csv$v" 60 > "$out/b${v}_step_breakdown.txt"
done
grep '^{' "$out/trace.log" > "$out/bench_under_rocprof.json" || true
rm -rf "$out/trace" "$out/trace1" "$out/csv64" "$out/csv8" "$out/csv8s" "$out/pmc_fetch" "$out/pmc_write"
head -12 "$out/kernel_stats_one_stream.csv"; tail -2 "$out/b8_step_breakdown.txt"; tail -2 "$out/b64_step_breakdown.txt"
