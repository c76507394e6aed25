# usage (on the GPU box, from the repo root): bash tools/final_pass.sh — the end-of-round pass behind profiles/rNN_*: GPU tests, smoke(), bench.py, tools/profile_step.sh
set -o pipefail
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/final/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/final/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/final/pytest_gpu.log
timeout -k 10 400 python tools/soak.py 200 > gpurun_out/final/soak.log 2>&1 || { tail -20 gpurun_out/final/soak.log; exit 1; }
tail -1 gpurun_out/final/soak.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1 || { tail gpurun_out/final/smoke.log; exit 1; }
tail -1 gpurun_out/final/smoke.log
timeout -k 10 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || { tail gpurun_out/final/bench.err; exit 1; }
tail -c 2500 gpurun_out/final/bench.json
timeout -k 10 900 bash tools/profile_step.sh final/prof > gpurun_out/final/profile.log 2>&1 || { tail -20 gpurun_out/final/profile.log; exit 1; }
tail -20 gpurun_out/final/profile.log
