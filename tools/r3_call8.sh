#!/bin/bash
set -o pipefail
out=gpurun_out/c8; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_ndt1_gpu.py tests/test_kernels_gpu.py tests/test_dp_gpu.py -q -x -s -k "benched_batch or adamw or two_rank" > $out/pytest.log 2>&1; tail -12 $out/pytest.log
