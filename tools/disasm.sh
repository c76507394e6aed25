#!/bin/bash
# usage: tools/disasm.sh <object.o> <build/out.s>: disassemble the gfx950 code object of a hipcc object
set -e
mkdir -p "$(dirname "$2")"
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fb.bin "$1"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $tmp/dev.co > "$2"
rm -rf $tmp
