#!/bin/bash
set -o pipefail
out=gpurun_out/c9; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_ndt1_gpu.py tests/test_trainer_gpu.py tests/test_gemm_gpu.py tests/test_kernels_gpu.py -q -x > $out/pytest.log 2>&1; tail -5 $out/pytest.log
timeout -k 10 300 python tools/ab_side_stream.py --batches 8 16 64 --windows 5 > $out/ab.txt 2>&1; cat $out/ab.txt
