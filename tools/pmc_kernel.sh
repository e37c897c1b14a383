# SQ counters of one kernel of a C4 scan (21 disks): tools/pmc_kernel.sh <kernel name substring> [counter ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-k_rowpair_stats}
shift
C=${@:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS}
O=$R/gpurun_out/pmc_kernel
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sq -- python3 $R/tools/step_loop.py 3 -10,-9,-8,-7,-6,-5,-4,-3,-2,-1,0,1,2,3,4,5,6,7,8,9,10 > /dev/null 2>&1
cd $R
python3 - "$K" <<'PY'
import csv, glob, sys, collections
K = sys.argv[1]
files = glob.glob('gpurun_out/pmc_kernel/sq/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(files[0])):
    if K in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('%-28s %14.0f per launch (%d launches)' % (k, sum(v) / len(v), len(v)))
PY
rm -rf $O
