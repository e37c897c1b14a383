mkdir -p gpurun_out/r2g
for q in 4 8 16; do for w in 2 4 6 8; do for rep in 1 2; do
GPU_MAX_HW_QUEUES=$q python bench.py --steps 80 --warmup 24 --workers $w --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('q=$q w=$w rep=$rep', d['value'], d['ms_per_step'], 'accA', r['avg_launch_ms'], 'ext', r['secondary']['avg_launch_ms'])"
done; done; done > gpurun_out/r2g/sweep.txt 2>&1
cat gpurun_out/r2g/sweep.txt
