#!/bin/bash
out=gpurun_out/c15; mkdir -p $out
for cfg in "0 0" "1 0" "0 4" "0 3" "0 2" "0 0"; do
  set -- $cfg
  echo "== notail $1 iters $2" >> $out/ab.txt
  NBCI_LIB=build/measure/libnbci.so NBCI_LNB_NOTAIL=$1 NBCI_LNB_ITERS=$2 timeout -k 10 120 python tools/ab_side_stream.py --batches 64 --windows 5 >> $out/ab.txt 2>&1
done
grep -v amdgpu $out/ab.txt
