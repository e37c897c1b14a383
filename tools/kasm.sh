#!/bin/bash
# Disassembly of one kernel of an object file of csrc/:  tools/kasm.sh extract.o 'k_extract_bandItLi4ELi64ELi4ELb0E' > k.s
set -e
obj=$1; pat=$2
L=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$tmp/fb.bin "$obj"
$L/clang-offload-bundler --unbundle --type=o --input=$tmp/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.o
$L/llvm-objdump -d $tmp/dev.o | awk -v pat="$pat" '/^[0-9a-f]+ <.*>:$/ {if (f) exit; if ($0 ~ pat) f=1} f {print}' | sed 's://.*$::'
rm -rf $tmp
