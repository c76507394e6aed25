#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]: a MEASUREMENT build of the library into build/<name>/libnbci.so
# (e.g. tools/build_variant.sh stamps -DNBCI_STAMPS); load it with NBCI_LIB=build/<name>/libnbci.so. build/ is git-ignored
# but travels to the GPU box with gpurun.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/$name
mkdir -p "$out"
cd "$root/llm_bci_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -mllvm -pragma-unroll-threshold=200000 "$@" -c "$f" -o "$out/${f%.hip}.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libnbci.so" "$out"/*.o
echo "built $out/libnbci.so"
