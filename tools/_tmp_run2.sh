#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r04m
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "limb" > gpurun_out/r04m/limb_tests.txt 2>&1; tail -n 3 gpurun_out/r04m/limb_tests.txt
bash tools/collect_profiles.sh r04 2>&1 | tail -5
