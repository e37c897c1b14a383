#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04p
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "rc=$?" >> $O/gpu_tests.txt; tail -n 3 $O/gpu_tests.txt
bash tools/step_table.sh 2>&1 | tail -3
for i in 1 2; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err; python3 -c "
import json; d=json.load(open('$O/b.json')); print(d['value'], d['ms_per_step'], d['repeats']['ms_per_step'], d['roofline']['frac'], d['whole_step']['frac'], 'c4', d['c4']['ms_per_step'], d['c4']['parity_vs_stage_route']['images_that_differ'], 'c5', d['c5_file']['ms_per_step'], d['c5_file']['parity_vs_stage_route']['images_that_differ'])"
done
