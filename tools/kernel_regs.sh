#!/bin/bash
# usage: tools/kernel_regs.sh <object.o> [name-filter]: VGPR / AGPR / spill / LDS of every gfx950 kernel in a hipcc object
set -e
obj=$1; pat=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fb.bin "$obj"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | grep -E "^\s+\.(name|vgpr_count|agpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):" \
  | awk '/\.name:/{if(name!="")print name, info; name=$2; info=""} !/\.name:/{info=info" "$1$2} END{print name, info}' | grep -E "$pat" | c++filt
rm -rf $tmp
