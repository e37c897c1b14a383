#!/bin/bash
for ns in 1 2 3 4; do for u in 2 4 8; do
  SHG_ACC_NT=1 SHG_ACC_UNROLL=$u SHG_ACC_NSPLIT=$ns python tools/bench_kernels.py "$@" | head -1
done; done
