#!/bin/bash
out=gpurun_out/c11; mkdir -p $out
for p in 0 1 0 1; do echo "== NBCI_SIDE_PRIO=$p" >> $out/ab.txt; NBCI_SIDE_PRIO=$p timeout -k 10 200 python tools/ab_side_stream.py --batches 8 64 --windows 5 >> $out/ab.txt 2>&1; done; grep -v amdgpu $out/ab.txt
