#!/bin/bash
out=gpurun_out/c12; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_attn_stream_gpu.py -q -x > $out/pytest.log 2>&1; tail -5 $out/pytest.log
for sh in 1 0; do
  for shape in "16 8 1501 768" "16 8 669 768" "2048 8 205 256" "64 8 593 1024"; do
    NBCI_LIB=build/measure/libnbci.so NBCI_FA_SHARED=$sh timeout -k 10 120 python tools/bench_flash.py $shape >> $out/flash.txt 2>&1
  done
done
grep -v amdgpu $out/flash.txt
