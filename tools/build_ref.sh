#!/bin/bash
# usage: tools/build_ref.sh <git-ref> <name>: builds the library of ANOTHER commit into build/<name>/libnbci.so (same flags as the Makefile), for
# in-box A/B runs of two commits (NBCI_LIB=build/<name>/libnbci.so python bench.py ...). build/ is git-ignored but travels with gpurun.
set -e
ref=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$ref" llm_bci_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/llm_bci_amd/csrc" -j8 > "$tmp/build.log" 2>&1 || { tail -20 "$tmp/build.log"; exit 1; }
mkdir -p "$root/build/$name"
cp "$tmp/llm_bci_amd/csrc/libnbci.so" "$root/build/$name/libnbci.so"
rm -rf "$tmp"
echo "built build/$name/libnbci.so from $ref"
