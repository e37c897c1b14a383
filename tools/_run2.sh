for w in 4 6; do python3 bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-extra --workers $w 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('workers $w: ms/step %.3f' % d['ms_per_step'], d['repeats']['ms_per_step'], 'passA in-flight %.3f frac %.3f alone %.3f kernels %.3f' % (r['avg_launch_ms'], r['frac'], r['frac_uncontended'], d['kernel_ms_per_step']))"; done
python3 tools/timeline.py 60 4 | head -40
