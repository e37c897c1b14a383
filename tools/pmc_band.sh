# FETCH_SIZE / WRITE_SIZE / L2 counters of pass B's two kernels for rotated files (k_extract, k_extract_band) on the shapes of
# tools/sweep_band.py:  tools/pmc_band.sh [case ...]   (case = c4 | c2 | c5 | u8 | u8s21 | list5; default c4)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_band
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  d=$O/$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/sweep_band.py ${@:-c4} > /dev/null 2>&1
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
# the launches of one kernel on one case follow each other (46 each: 1 + 5 + 40); report the per-launch mean of every such run
for (k, c), v in sorted(acc.items()):
    runs = [v[i:i + 46] for i in range(0, len(v), 46)]
    print('%-62s %-24s %s' % (k, c, '  '.join('%.4g' % (sum(r) / len(r)) for r in runs)))
PY
