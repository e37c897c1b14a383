#!/bin/bash
# Round-3 sweep: one-call route / frame-pass lane / CU-masked worker streams / worker count, driver flags.
# usage: tools/sweep_lane.sh [out_dir]
out=${1:-gpurun_out/sweep_lane}
mkdir -p "$out"
run() {   # name, env..., -- bench args
    name=$1; shift
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" python3 bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-extra "$@" > "$out/$name.json" 2> "$out/$name.log"
    python3 - "$out/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d['roofline']
    print('%-28s ms/step %.3f (min %.3f max %.3f)  passA in-flight %.3f ms frac %.3f  alone %.3f  kernels %.3f ms' % (
        sys.argv[2], d['ms_per_step'], d['repeats']['min'], d['repeats']['max'], r['avg_launch_ms'], r['frac'], r['frac_uncontended'],
        d['kernel_ms_per_step']))
except Exception as e:
    print('%-28s FAILED %r' % (sys.argv[2], e))
PY
}
run default --
run no_lane SHG_FRAME_LANE=0 --
run stage_route SHG_SCAN_CALL=0 --
run stage_route_no_lane SHG_SCAN_CALL=0 SHG_FRAME_LANE=0 --
run cus64 SHG_CHAIN_CUS=64 --
run cus128 SHG_CHAIN_CUS=128 --
run cus192 SHG_CHAIN_CUS=192 --
run workers2 -- --workers 2
run workers3 -- --workers 3
run workers6 -- --workers 6
run workers8 -- --workers 8
run workers8_cus128 SHG_CHAIN_CUS=128 -- --workers 8
run one_stack -- --stacks 1
