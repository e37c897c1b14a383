#!/bin/bash
# Launch-shape sweep of pass A for 8-bit scans (tuning aid).
for sp in 2 3 4 5 6 8 12; do for un in 2 4 8; do
  echo -n "nsplit $sp unroll $un: "; SHG_ACC_NSPLIT=$sp SHG_ACC_UNROLL=$un python tools/bench_kernels.py --bits 8 2>&1 | grep "pass A" | cut -c1-110
done; done
echo -n "heuristic: "; python tools/bench_kernels.py --bits 8 2>&1 | grep "pass A" | cut -c1-110
