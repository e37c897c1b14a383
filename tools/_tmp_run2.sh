#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
python3 tools/lane_gaps.py 60 4 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-400
python3 tools/lane_gaps.py 60 8 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-300
for cfg in "0 4" "0 6" "0 8" "1 8"; do
  set -- $cfg
  SHG_COMBINE=$1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --workers $2 --no-cpu-baseline --no-e2e --no-extra > $O/b.json 2> $O/b.err
  python3 - $O/b.json "$cfg" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    r=d['roofline']
    print('   combine workers =',sys.argv[2],'ms/step',d['ms_per_step'],d['repeats']['ms_per_step'],'passA',r['avg_launch_ms'])
except Exception as e:
    print('failed',sys.argv[2],e)
PY
done
