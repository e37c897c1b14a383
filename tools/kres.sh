#!/bin/bash
# Registers / LDS / scratch of the kernels in one object file of csrc/ (from the code object's metadata):
#   tools/kres.sh extract.o [name filter]
set -e
obj=$1; pat=${2:-.}
L=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$tmp/fb.bin "$obj"
$L/clang-offload-bundler --unbundle --type=o --input=$tmp/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.o
$L/llvm-readelf --notes $tmp/dev.o | awk '
  /\.name:/ {name=$2}
  /\.vgpr_count:/ {v=$2} /\.agpr_count:/ {a=$2} /\.sgpr_count:/ {s=$2}
  /\.vgpr_spill_count:/ {sp=$2} /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2}
  /\.wavefront_size:/ {printf "%-90s vgpr %3d agpr %3d sgpr %3d spill %3d lds %6d scratch %5d\n", name, v, a, s, sp, l, p}' | c++filt | grep -E "$pat" || true
rm -rf $tmp
