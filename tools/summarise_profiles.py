"""gpurun_out/<tag>_* (tools/collect_profiles.sh) -> profiles/<tag>_* summaries and profiles/traffic.json."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, 'gpurun_out'), os.path.join(root, 'profiles')


def newest(pattern):
    """gpurun merges every collection into the same directories: take the latest run's file."""
    return max(glob.glob(pattern), key=os.path.getmtime)


shutil.copy(newest(os.path.join(go, tag + '_trace', '*', '*_kernel_stats.csv')), os.path.join(pr, tag + '_bench_kernel_stats.csv'))
shutil.copy(os.path.join(go, tag + '_bench_under_rocprof.json'), os.path.join(pr, tag + '_bench_under_rocprof.json'))
shutil.copy(os.path.join(go, tag + '_bench.json'), os.path.join(pr, tag + '_bench.json'))
if os.path.exists(os.path.join(go, tag + '_step_kernel_table.txt')):
    shutil.copy(os.path.join(go, tag + '_step_kernel_table.txt'), os.path.join(pr, tag + '_step_kernel_table.txt'))
out = {}
for which in ('fetch', 'write'):
    f = newest(os.path.join(go, '%s_pmc_%s' % (tag, which), '*', '*_counter_collection.csv'))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_' in n and 'anonymous' in n:
            acc[n.split('::')[-1].split('(')[0]].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        out.setdefault(k, {})[which.upper() + '_SIZE_KB_mean'] = round(sum(v) / len(v), 1)
        out[k][which.upper() + '_SIZE_launches'] = len(v)
json.dump(out, open(os.path.join(pr, tag + '_pmc_fetch_write_per_kernel.json'), 'w'), indent=1, sort_keys=True)
a = [v for k, v in out.items() if k.startswith('k_accumulate_vec')][0]
# gfx950: FETCH_SIZE counts a wide coalesced 16 B/lane stream at exactly half (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact
traffic = int(2 * a['FETCH_SIZE_KB_mean'] * 1024 + a['WRITE_SIZE_KB_mean'] * 1024)
json.dump({'2000x2000x200x16': {'accumulate_bytes_per_launch': traffic,
                                'source': 'profiles/%s_pmc_fetch_write_per_kernel.json: 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 '
                                          '(gfx950 FETCH_SIZE half-count correction), separate --pmc passes' % tag}},
          open(os.path.join(pr, 'traffic.json'), 'w'), indent=1)
print('traffic per launch', traffic, '= %.4f x algorithmic' % (traffic / 1.6e9))
for r in csv.DictReader(open(os.path.join(pr, tag + '_bench_kernel_stats.csv'))):
    if 'k_accumulate' in r['Name'] or 'k_extract' in r['Name']:
        print(r['Name'][:60], 'calls', r['Calls'], 'avg_ns', r['AverageNs'])
print(open(os.path.join(pr, tag + '_bench.json')).read()[:600])
