#!/bin/bash
out=gpurun_out/c18; mkdir -p $out
for v in 0 1 2; do NBCI_LIB=build/measure/libnbci.so NBCI_GEMM_MS_BIG=$v timeout -k 10 200 python tools/bench_ptst.py --dtype fp8 --steps 5 >> $out/ptst.txt 2>&1; done; grep PatchTST $out/ptst.txt
