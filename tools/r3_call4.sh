#!/bin/bash
set -o pipefail
out=gpurun_out/c4; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_attention_f64_gpu.py -q -x -s > $out/pytest_attn.log 2>&1; tail -15 $out/pytest_attn.log
timeout -k 10 900 python -m pytest tests/test_ndt1_gpu.py tests/test_trainer_gpu.py -q -x > $out/pytest_ndt1.log 2>&1; tail -5 $out/pytest_ndt1.log
for v in 1 0; do echo "== NBCI_ATTN_BWD1=$v" >> $out/ab.txt; NBCI_ATTN_BWD1=$v timeout -k 10 200 python tools/ab_side_stream.py --batches 8 64 --windows 5 >> $out/ab.txt 2>&1; done; cat $out/ab.txt
