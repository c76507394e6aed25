#!/bin/bash
# build/<name>/libnbci.so = the library with one source rebuilt under extra flags:  tools/build_variant.sh NAME FILE.hip -DFLAG=1 ...
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../llm_bci_amd/csrc"
mkdir -p ../../build/$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -mllvm -pragma-unroll-threshold=200000 -DNBCI_MEASURE "$@" -c $src -o ../../build/$name/${src%.hip}.o
objs=""
for o in *.o; do if [ "$o" = "${src%.hip}.o" ]; then objs="$objs ../../build/$name/$o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/$name/libnbci.so $objs
echo "build/$name/libnbci.so"
