python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C2: ms/step %.3f' % d['ms_per_step'], d['repeats']['ms_per_step']); print('e2e', d['e2e']['value'], d['e2e']['ms_per_file'], d['e2e']['host_to_device_GBps_per_gpu']); print('c3', d['sharded_c3']['value'], d['sharded_c3']['ms_per_scan'])"
