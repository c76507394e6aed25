#!/bin/bash
out=gpurun_out/c13; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_attn_stream_gpu.py -q -x > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for cfg in "1 0" "1 1" "0 0"; do
  set -- $cfg
  echo "== shared $1 nkv $2" >> $out/flash.txt
  for shape in "16 8 1501 768" "2048 8 205 256" "16 8 669 768"; do
    NBCI_LIB=build/measure/libnbci.so NBCI_FA_SHARED=$1 NBCI_FA_NKV=$2 timeout -k 10 120 python tools/bench_flash.py $shape >> $out/flash.txt 2>&1
  done
done
grep -v amdgpu $out/flash.txt
