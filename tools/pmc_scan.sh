# SQ counters of every kernel of a scan (C2 by default; any step_loop.py arguments after the counters' "--"):
#   tools/pmc_scan.sh [counter ...] [-- steps shifts frames width height bits]
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do C="$C $1"; shift; done; [ "$1" = "--" ] && shift
C=${C:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES}
A=${@:-5}
O=$R/gpurun_out/pmc_scan
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sq -- python3 $R/tools/step_loop.py $A > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
files = glob.glob('gpurun_out/pmc_scan/sq/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(files[0])):
    k = r.get('Kernel_Name', '')
    if 'k_' not in k or 'at::native' in k:
        continue
    name = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:34]
    acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted(acc, key=lambda n: -sum(acc[n].get('SQ_WAVE_CYCLES', [0])) / max(1, len(acc[n].get('SQ_WAVE_CYCLES', [1]))))
cols = sorted({c for n in acc for c in acc[n]})
print('%-36s' % 'kernel' + ''.join('%16s' % c[-14:] for c in cols))
for n in names:
    print('%-36s' % n + ''.join('%16.0f' % (sum(acc[n][c]) / max(1, len(acc[n][c]))) for c in cols))
PY
rm -rf $O
