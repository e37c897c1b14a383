#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench lines, rocprofv3 kernel trace + stats (default bench and --workers 1), separate PMC passes (FETCH_SIZE, WRITE_SIZE) for
#   C2 / C4 / C5 scans, per-kernel roofline tables made from them (tools/roofline_table.py -> profiles/<tag>_roofline_table_*.txt),
#   SQ / LDS counters of the kernels DESIGN.md section 3 talks about (profiles/<tag>_sq_<kernel>.json), the lane's timeline
#   (tools/lane_gaps.py) and what each part of a chain costs pass A (tools/interference.py).  Results land in gpurun_out/ and, for
#   the tables, directly in profiles/ of the box's copy -- which gpurun does not bring back -- so they are copied to
#   gpurun_out/<tag>_profiles/ too; tools/summarise_profiles.py turns gpurun_out/ into profiles/*.
# Under rocprofv3 the program itself follows `--` (no env / bash -c hop).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
O=$R/gpurun_out
P=$O/${TAG}_profiles
mkdir -p $O $P
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_flags.json 2> /dev/null
python3 $R/bench.py --workers 1 --no-e2e --no-extra > $O/${TAG}_bench_w1.json 2> /dev/null
for cfg in "c1 --frames 200 --width 800 --height 120 --bits 8" "c3_one_gpu --frames 4000" "c4 --shifts=-10:10:1" "c5_per_gpu --frames 4000 --width 2560 --height 256" "u8 --bits 8" "n500 --frames 500"; do
  set -- $cfg; t=$1; shift
  python3 $R/bench.py --no-cpu-baseline --no-e2e --no-extra "$@" > $O/${TAG}_bench_$t.json 2> /dev/null
done
python3 $R/tools/lane_gaps.py 60 4 2> /dev/null > $P/${TAG}_lane_gaps.txt
python3 $R/tools/interference.py 1.0 2> /dev/null > $P/${TAG}_interference.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 $R/bench.py --no-cpu-baseline --no-e2e --no-extra --steps 20 --warmup 5 > $O/${TAG}_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_w1 -- python3 $R/bench.py --workers 1 --no-cpu-baseline --no-e2e --no-extra --steps 20 --warmup 5 > $O/${TAG}_bench_w1_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $R/bench.py --workers 1 --no-cpu-baseline --no-e2e --no-extra --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/bench.py --workers 1 --no-cpu-baseline --no-e2e --no-extra --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
C4=-10,-9,-8,-7,-6,-5,-4,-3,-2,-1,0,1,2,3,4,5,6,7,8,9,10
# per-scan kernel tables of the per-file flow alone (no probes, no generator), with the PMC passes of the same loop
for cfg in "c2 20 0" "c4 10 $C4" "c5 10 0 4000 2560 256 16"; do
  set -- $cfg; name=$1; steps=$2; shift; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_step_trace_$name -- python3 $R/tools/step_loop.py $steps "$@" > $O/${TAG}_step_dims_$name.txt 2> /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch_$name -- python3 $R/tools/step_loop.py 3 "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write_$name -- python3 $R/tools/step_loop.py 3 "$@" > /dev/null 2>&1
done
# the same tables for a scan whose circularised disk CLAHE's 2 x 2 grid does NOT divide (2097 px wide: four of the bench's five
# synthetic scans, tools/step_loop.py), and how the scans in flight share the device (tools/overlap.py over tools/pool_loop.py)
export SHG_STEP_SEED=1
for cfg in "c2 20 0" "c4 10 $C4"; do
  set -- $cfg; name=$1; steps=$2; shift; shift
  rm -rf /tmp/odd_trace
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/odd_trace -- python3 $R/tools/step_loop.py $steps "$@" > /dev/null 2>&1
  python3 $R/tools/kernel_table.py /tmp/odd_trace $steps > $P/${TAG}_step_kernel_table_${name}_odd_width.txt
done
unset SHG_STEP_SEED
for cfg in "c2 60 0" "c4 40 $C4"; do
  set -- $cfg; name=$1; steps=$2; shift; shift
  rm -rf /tmp/ov_trace
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ov_trace -- python3 $R/tools/pool_loop.py $steps 4 "$@" > $P/${TAG}_overlap_$name.txt 2> /dev/null
  python3 $R/tools/overlap.py /tmp/ov_trace $steps >> $P/${TAG}_overlap_$name.txt
done
rm -rf /tmp/odd_trace /tmp/ov_trace
# SQ / LDS counters (one run) and L2 counters (another) of every kernel of a C4 scan -> profiles/<tag>_sq_<kernel>.json
rm -rf $O/${TAG}_sq_tmp $O/${TAG}_tcc_tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/${TAG}_sq_tmp -- python3 $R/tools/step_loop.py 3 $C4 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/${TAG}_tcc_tmp -- python3 $R/tools/step_loop.py 3 $C4 > /dev/null 2>&1
python3 - "$O/${TAG}_sq_tmp" "$O/${TAG}_tcc_tmp" "$P" "$TAG" <<'PY'
import collections, csv, glob, json, re, sys
dirs, out, tag = sys.argv[1:3], sys.argv[3], sys.argv[4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r'\bk_[a-z0-9_]+', r.get('Kernel_Name', ''))
            if m and 'anonymous' in r['Kernel_Name']:
                acc[m.group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
for k, counters in acc.items():
    json.dump({'kernel': k, 'workload': 'C4 scan (21 requested disks), tools/step_loop.py, one scan at a time; values per launch',
               'launches': {c: len(v) for c, v in counters.items()}, 'per_launch': {c: sum(v) / len(v) for c, v in counters.items()}},
              open('%s/%s_sq_%s.json' % (out, tag, k), 'w'), indent=1, sort_keys=True)
PY
rm -rf $O/${TAG}_sq_tmp $O/${TAG}_tcc_tmp
cd $R
for cfg in "c2 20" "c4 10" "c5 10"; do
  set -- $cfg
  dims=$(tail -n 1 $O/${TAG}_step_dims_$1.txt)
  python3 tools/kernel_table.py gpurun_out/${TAG}_step_trace_$1 $2 > gpurun_out/${TAG}_step_kernel_table_$1.txt
  python3 tools/roofline_table.py $TAG $1 $2 gpurun_out/${TAG}_step_trace_$1 gpurun_out/${TAG}_pmc_fetch_$1 gpurun_out/${TAG}_pmc_write_$1 -- $dims > /dev/null
  cp profiles/${TAG}_roofline_table_$1.txt $P/
  tail -n 1 gpurun_out/${TAG}_step_kernel_table_$1.txt
done
