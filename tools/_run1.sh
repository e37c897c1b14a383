python -m pytest tests -x -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k/step_trace_c4 -- python3 $R/tools/step_loop.py 10 -10,-9,-8,-7,-6,-5,-4,-3,-2,-1,0,1,2,3,4,5,6,7,8,9,10 > /dev/null 2>&1
cd $R
python3 tools/kernel_table.py gpurun_out/k/step_trace_c4 10
python3 bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('C2: ms/step %.3f' % d['ms_per_step'], d['repeats']['ms_per_step'], 'passA in-flight %.3f frac %.3f' % (r['avg_launch_ms'], r['frac']))
print('c4', json.dumps(d['c4']))
print('c5', json.dumps(d['c5_file']))
print('whole', json.dumps(d['whole_step']))"
