#!/bin/bash
set -o pipefail
out=gpurun_out/c16; mkdir -p $out
timeout -k 10 300 python tools/rccl_smoke.py > $out/rccl.txt 2>&1; tail -4 $out/rccl.txt
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py tests/test_dp_gpu_cabi.py tests/test_trainer_gpu.py -q -x > $out/pytest.log 2>&1; tail -5 $out/pytest.log
