#!/bin/bash
set -o pipefail
out=gpurun_out/c7; mkdir -p $out
( time timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_reentrancy_gpu.py tests/test_ckpt_interchange.py tests/test_gemm_streamk_gpu.py -q -x -s --durations=5 ) > $out/pytest.log 2>&1; tail -30 $out/pytest.log
