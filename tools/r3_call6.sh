#!/bin/bash
set -o pipefail
out=gpurun_out/c6; mkdir -p $out
( time timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err ) 2> $out/bench.time || { tail $out/bench.err; exit 1; }
tail -3 $out/bench.time; python - <<'PY'
import json
d=json.load(open('gpurun_out/c6/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','repeats','ms_per_step_min','ms_per_step_max')})
r=d['roofline']; print({k:v for k,v in r.items() if k!='per_kernel'})
for k in r['per_kernel'][:14]: print(k)
print(json.dumps(d['extra_points'],indent=0)[:3500])
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['points'])
PY
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log
