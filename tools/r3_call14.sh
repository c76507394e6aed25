#!/bin/bash
out=gpurun_out/c14; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_attn_stream_gpu.py tests/test_itr_gpu.py tests/test_ptst_gpu.py tests/test_fp8_gpu.py tests/test_fullsize_gpu.py tests/test_ndt1_gpu.py -q -x > $out/pytest.log 2>&1; tail -5 $out/pytest.log
timeout -k 10 200 python tools/bench_itr.py --channels 1500 --steps 10 > $out/itr.txt 2>&1; tail -1 $out/itr.txt
timeout -k 10 200 python tools/bench_ptst.py --dtype fp8 > $out/ptst.txt 2>&1; tail -1 $out/ptst.txt
