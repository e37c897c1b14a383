# FETCH_SIZE / WRITE_SIZE / L2 counters of pass B's kernels on a rotated C4-shaped stack:  tools/pmc_band.sh <shape> [<shape> ...]
# (shape = G,DK,NW[,dbg] of the band kernel; the general kernel is always measured too)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_band
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  d=$O/$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/sweep_band.py "$@" > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
order = []
for f in glob.glob('gpurun_out/pmc_band/**/*counter_collection.csv', recursive=True):
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'k_extract' not in k:
            continue
        acc[(k.split('(')[0][-60:], r['Counter_Name'])].append(float(r['Counter_Value']))
# the launches of one shape follow each other (46 each: 1 + 5 + 40); report the per-launch mean of every run of equal kernel names
for (k, c), v in sorted(acc.items()):
    runs = [v[i:i + 46] for i in range(0, len(v), 46)]
    print('%-62s %-24s %s' % (k, c, '  '.join('%.4g' % (sum(r) / len(r)) for r in runs)))
PY
