#!/bin/bash
# frames/s of bench.py with the driver's flags (--steps 20 --warmup 5) against the number of scan workers
mkdir -p gpurun_out/r2n
for w in 2 3 4 5 6 8; do for rep in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 --workers $w --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('w=$w rep=$rep', d['value'], d['ms_per_step'], 'accA', r['avg_launch_ms'])"
done; done
for w in 4 6; do for rep in 1 2; do
python bench.py --gpus 1 --steps 200 --warmup 10 --workers $w --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('steps=200 w=$w rep=$rep', d['value'], d['ms_per_step'], 'accA', r['avg_launch_ms'])"
done; done
