for w in 8 8 4; do python3 bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-extra --workers $w --repeats 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('native pool, workers $w: ms/step %.3f' % d['ms_per_step'], d['repeats']['ms_per_step'])"; done
python3 bench.py --steps 20 --warmup 30 --no-e2e --no-cpu-baseline --no-extra --workers 8 --repeats 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('warmup 30, workers 8: ms/step %.3f' % d['ms_per_step'], d['repeats']['ms_per_step'])"
