#!/bin/bash
set -o pipefail
out=gpurun_out/c2; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_trainer_gpu.py tests/test_bci_gpu.py -q -x > $out/pytest.log 2>&1; tail -5 $out/pytest.log
timeout -k 10 300 python tools/ab_side_stream.py > $out/ab.txt 2>&1; cat $out/ab.txt
