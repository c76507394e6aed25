#!/bin/bash
out=gpurun_out/c19; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_ndt1_gpu.py -q -x -k "maximum_length" > $out/pytest.log 2>&1; tail -8 $out/pytest.log
