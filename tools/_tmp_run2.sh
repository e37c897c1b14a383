#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 bench.py --workers 1 --no-e2e --no-extra > gpurun_out/r04_bench_w1.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r04_bench_w1.json')); print('w1', d['ms_per_step'], d['repeats']['ms_per_step'], d['kernel_ms_per_step'])"
