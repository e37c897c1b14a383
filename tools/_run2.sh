one() {   # label, env..., -- args
    label=$1; shift
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" python3 bench.py --steps 20 --warmup 8 --no-e2e --no-cpu-baseline --no-extra --repeats 7 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-26s ms/step %.3f (min %.3f max %.3f) passA in-flight %.3f frac %.3f alone %.3f' % ('$label', d['ms_per_step'], d['repeats']['min'], d['repeats']['max'], r['avg_launch_ms'], r['frac'], r['frac_uncontended']))"
}
one "w4" -- --workers 4
one "w5" -- --workers 5
one "w6" -- --workers 6
one "w8" -- --workers 8
one "w4 inflight 5MiB" SHG_ACC_INFLIGHT_KIB=5120 -- --workers 4
one "w4 inflight 10MiB" SHG_ACC_INFLIGHT_KIB=10240 -- --workers 4
one "w4 inflight 14MiB" SHG_ACC_INFLIGHT_KIB=14336 -- --workers 4
one "w6 inflight 10MiB" SHG_ACC_INFLIGHT_KIB=10240 -- --workers 6
one "w4 nsplit4" SHG_ACC_NSPLIT=4 -- --workers 4
one "w4 nsplit3 unroll4" SHG_ACC_NSPLIT=3 SHG_ACC_UNROLL=4 -- --workers 4
