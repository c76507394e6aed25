#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_step.sh <outdir under gpurun_out/>
# One kernel-trace run of bench.py plus three counter passes (own runs: FETCH_SIZE and WRITE_SIZE do not fit one pass), each with
# the library's GEMM launch log, then tools/roofline_report.py -> <outdir>/roofline.json and the --stats summary as CSV.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-extra-points --no-roofline --repeats 1"
NBCI_GEMM_LOG=$out/log_trace.txt timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o t -- python3 $B --steps 20 --warmup 3 > "$out/trace.log" 2>&1
NBCI_GEMM_LOG=$out/log_fetch.txt timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o t -- python3 $B --steps 4 --warmup 1 > "$out/pmc_fetch.log" 2>&1
NBCI_GEMM_LOG=$out/log_write.txt timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o t -- python3 $B --steps 4 --warmup 1 > "$out/pmc_write.log" 2>&1
NBCI_GEMM_LOG=$out/log_mfma.txt timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d "$out/pmc_mfma" -o t -- python3 $B --steps 4 --warmup 1 > "$out/pmc_mfma.log" 2>&1
cd "$R"
db() { find "$out/$1" -name '*.db' | head -1; }
python3 tools/roofline_report.py --trace "$(db trace)" --log "$out/log_trace.txt" \
  --pmc "$(db pmc_fetch)" "$(db pmc_write)" "$(db pmc_mfma)" --pmc-log "$out/log_fetch.txt" "$out/log_write.txt" "$out/log_mfma.txt" \
  --out "$out/roofline.json" | tee "$out/roofline.txt"
python3 tools/db_stats.py "$(db trace)" > "$out/kernel_stats.csv"
grep '^{' "$out/trace.log" > "$out/bench_under_rocprof.json" || true
# the raw databases are large: keep only the summaries in gpurun_out
rm -rf "$out/trace" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_mfma"
