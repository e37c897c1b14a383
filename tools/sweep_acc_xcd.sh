#!/bin/bash
# Pass A with the plain (column block, split) grid vs the XCD-aware mapping (every XCD reads one frame split).
for shape in "2 4" "2 8" "4 2" "4 4" "8 2" "8 4" "1 8" "1 16"; do set -- $shape
  for x in 0 1; do
    echo -n "nsplit $1 unroll $2 xcd $x: "; SHG_ACC_XCD=$x SHG_ACC_NSPLIT=$1 SHG_ACC_UNROLL=$2 python tools/bench_kernels.py 2>&1 | grep "pass A" | cut -c1-100
  done
done
