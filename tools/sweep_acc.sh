#!/bin/bash
# sweep of the pass-A launch shape (tuning aid): prints one line per configuration
for nt in 1 0; do for u in 4 8 16; do for tb in 256 512 768 1024 1536 2048 3072; do
  SHG_ACC_NT=$nt SHG_ACC_UNROLL=$u SHG_ACC_TARGET_BLOCKS=$tb python tools/bench_kernels.py "$@" | head -1
done; done; done
