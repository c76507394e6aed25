# usage (on the GPU box, from the repo root): bash tools/final_pass.sh — the end-of-round pass behind profiles/rNN_*: GPU tests, soak, smoke(), bench.py, tools/profile_step.sh
set -o pipefail
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/final/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/final/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/final/pytest_gpu.log
timeout -k 10 400 python tools/soak.py 200 > gpurun_out/final/soak.log 2>&1 || { tail -20 gpurun_out/final/soak.log; exit 1; }
tail -1 gpurun_out/final/soak.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1 || { tail gpurun_out/final/smoke.log; exit 1; }
tail -1 gpurun_out/final/smoke.log
timeout -k 10 300 python tools/rccl_smoke.py > gpurun_out/final/rccl_smoke.txt 2>&1 || { tail gpurun_out/final/rccl_smoke.txt; exit 1; }
tail -1 gpurun_out/final/rccl_smoke.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || { tail gpurun_out/final/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/final/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "repeats", "ms_per_step_min", "ms_per_step_max")}, "B8", d["extra_points"]["B8_full_length"])
r = d["roofline"]; print({k: v for k, v in r.items() if k != "per_kernel"})
PY
timeout -k 10 300 python tools/ab_side_stream.py --batches 8 16 32 64 > gpurun_out/final/side_stream_ab.txt 2>&1; grep -v amdgpu gpurun_out/final/side_stream_ab.txt
timeout -k 10 1100 bash tools/profile_step.sh final/prof > gpurun_out/final/profile.log 2>&1 || { tail -20 gpurun_out/final/profile.log; exit 1; }
tail -12 gpurun_out/final/profile.log
# the two secondary models: step time + per-kernel rocprofv3 stats
( cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && \
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/itr -o t -- python3 $R/tools/bench_itr.py --channels 1500 --steps 5 > $R/gpurun_out/final/itr_n1500.log 2>&1 && \
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/ptst -o t -- python3 $R/tools/bench_ptst.py --steps 3 > $R/gpurun_out/final/ptst_c5.log 2>&1 )
python3 tools/db_stats.py "$(find gpurun_out/final/itr -name '*.db' | head -1)" > gpurun_out/final/itr_n1500_kernel_stats.csv
python3 tools/db_stats.py "$(find gpurun_out/final/ptst -name '*.db' | head -1)" > gpurun_out/final/ptst_c5_kernel_stats.csv
rm -rf gpurun_out/final/itr gpurun_out/final/ptst
for n in 1500 668; do timeout -k 10 200 python tools/bench_itr.py --channels $n 2>&1 | grep -v amdgpu; done | tee gpurun_out/final/itr_steps.txt
timeout -k 10 200 python tools/bench_ptst.py 2>&1 | grep -v amdgpu | tee gpurun_out/final/ptst_step.txt
head -8 gpurun_out/final/itr_n1500_kernel_stats.csv
