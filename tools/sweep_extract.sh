#!/bin/bash
# Sweep the load-batch depth of k_extract (tuning aid): rotated / un-rotated layout, S = 2 and S = 21.
for b in 1 2 4 8 16; do
  for s in 2 21; do
    echo "== batch $b shifts $s rotated"; SHG_EXT_BATCH=$b python tools/bench_kernels.py --shifts $s 2>&1 | grep "pass B"
  done
  echo "== batch $b shifts 2 un-rotated"; SHG_EXT_BATCH=$b python tools/bench_kernels.py --w 200 --h 2000 --shifts 2 2>&1 | grep "pass B"
done
