#!/bin/bash
# pass A launch-shape sweep for the shapes off the C2 sweet spot: a 500-frame shard (C3 / 8 ranks) and 8-bit files
for cfg in "--n 500" "--n 2000 --bits 8" "--n 500 --bits 8"; do
  echo "== $cfg: default plan"; python tools/bench_kernels.py $cfg 2>/dev/null | grep "pass A"
  for sp in 1 2 3 4 6 8; do for un in 2 4 8; do
    SHG_ACC_NSPLIT=$sp SHG_ACC_UNROLL=$un python tools/bench_kernels.py $cfg 2>/dev/null | grep "pass A" | sed "s/^/sp=$sp un=$un /" | cut -c1-110
  done; done
done
