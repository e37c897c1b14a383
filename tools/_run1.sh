python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('C2: ms/step %.3f' % d['ms_per_step'], d['repeats']['ms_per_step'], 'passA in-flight %.3f frac %.3f kernels %.3f' % (r['avg_launch_ms'], r['frac'], d['kernel_ms_per_step']))"; done
python3 tools/host_budget.py 80 4 | head -8
