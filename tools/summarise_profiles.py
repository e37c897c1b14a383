"""gpurun_out/<tag>_* (tools/collect_profiles.sh) -> profiles/<tag>_* summaries, profiles/traffic.json and
profiles/<tag>_numbers.txt (what this script prints: the figures DESIGN.md / profiles/README.md quote)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
_lines = []


def say(*a):
    line = ' '.join(str(x) for x in a)
    _lines.append(line)
    print(line)

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, 'gpurun_out'), os.path.join(root, 'profiles')


def newest(pattern):
    """gpurun merges every collection into the same directories: take the latest run's file."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def copy(src, dst):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(pr, dst))


shutil.copy(newest(os.path.join(go, tag + '_trace', '*', '*_kernel_stats.csv')), os.path.join(pr, tag + '_bench_kernel_stats.csv'))
shutil.copy(newest(os.path.join(go, tag + '_trace_w1', '*', '*_kernel_stats.csv')), os.path.join(pr, tag + '_bench_w1_kernel_stats.csv'))
for name in ('bench', 'bench_driver_flags', 'bench_w1', 'bench_under_rocprof', 'bench_w1_under_rocprof', 'bench_c1', 'bench_c3_one_gpu', 'bench_c4',
             'bench_c5_per_gpu', 'bench_u8', 'bench_n500'):
    copy(os.path.join(go, '%s_%s.json' % (tag, name)), '%s_%s.json' % (tag, name))
for name in ('step_kernel_table_c2', 'step_kernel_table_c4', 'step_kernel_table_c5'):
    copy(os.path.join(go, '%s_%s.txt' % (tag, name)), '%s_%s.txt' % (tag, name))


def pmc(suffix):
    out = {}
    for which in ('fetch', 'write'):
        f = newest(os.path.join(go, '%s_pmc_%s%s' % (tag, which, suffix), '*', '*_counter_collection.csv'))
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            m = re.search(r'\bk_[a-z0-9_]+', n)             # the kernel's own name: an argument type such as shg::PtrBatch also holds '::'
            if m and 'anonymous' in n:
                acc[m.group(0)].append(float(r['Counter_Value']))
        for k, v in sorted(acc.items()):
            out.setdefault(k, {})[which.upper() + '_SIZE_KB_mean'] = round(sum(v) / len(v), 1)
            out[k][which.upper() + '_SIZE_launches'] = len(v)
    return out


def traffic_of(table, prefix):
    # gfx950: FETCH_SIZE counts a wide coalesced 16 B/lane stream at exactly half (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact
    a = [v for k, v in table.items() if k.startswith(prefix)][0]
    return int(2 * a['FETCH_SIZE_KB_mean'] * 1024 + a['WRITE_SIZE_KB_mean'] * 1024)


for f in glob.glob(os.path.join(go, tag + '_profiles', '*')):
    shutil.copy(f, os.path.join(pr, os.path.basename(f)))

traffic = {}
for suffix, key, what in (('', '2000x2000x200x16', 'bench.py --workers 1'), ('_c2', None, 'step_loop C2'), ('_c4', None, 'step_loop C4'), ('_c5', '4000x2560x256x16', 'step_loop C5')):
    try:
        table = pmc(suffix)
    except ValueError:
        continue
    json.dump(table, open(os.path.join(pr, '%s_pmc_fetch_write_per_kernel%s.json' % (tag, suffix)), 'w'), indent=1, sort_keys=True)
    if key:
        t = traffic_of(table, 'k_accumulate_vec')
        traffic[key] = {'accumulate_bytes_per_launch': t,
                        'source': 'profiles/%s_pmc_fetch_write_per_kernel%s.json (%s): 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE '
                                  'half-count correction), separate rocprofv3 --pmc passes; PMC counters cannot be read from inside the '
                                  'process, so bench.py quotes this file' % (tag, suffix, what)}
        say(key, 'pass A HBM traffic per launch (PMC)', t, 'bytes')
json.dump(traffic, open(os.path.join(pr, 'traffic.json'), 'w'), indent=1)
for which in ('', '_w1'):
    for r in csv.DictReader(open(os.path.join(pr, tag + '_bench%s_kernel_stats.csv' % which))):
        if 'k_accumulate' in r['Name'] or 'k_extract' in r['Name']:
            say('rocprofv3 --stats, bench %s:' % ('--workers 1' if which else '(4 workers)'), r['Name'].split('::')[-1][:40], 'calls', r['Calls'], 'avg_ns', r['AverageNs'])
for name in ('bench', 'bench_driver_flags', 'bench_w1', 'bench_c1', 'bench_c3_one_gpu', 'bench_c4', 'bench_c5_per_gpu', 'bench_u8', 'bench_n500'):
    p = os.path.join(pr, '%s_%s.json' % (tag, name))
    if os.path.exists(p) and os.path.getsize(p):
        d = json.load(open(p))
        r = d['roofline']
        say('%-20s %10.0f f/s  %.3f ms/step  kernels %.3f ms  busy %.2f  passA in-flight %.0f GB/s (%.3f ms)  uncontended %.0f GB/s (%.3f ms)  extract %.1f us' % (
            name, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['gpu_busy_frac'], r['achieved'], r['avg_launch_ms'],
            r['uncontended']['achieved'], r['uncontended']['avg_launch_ms'], r['secondary']['uncontended_avg_launch_ms'] * 1e3))
        if d.get('e2e'):
            say('   e2e', d['e2e']['value'], 'f/s', d['e2e']['host_to_device_GBps_per_gpu'], 'GB/s;  c3', d['sharded_c3']['value'], 'f/s', d['sharded_c3']['ms_per_scan'], 'ms/scan')
        if d.get('cpu_baseline'):
            say('   cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('parity_vs_gpu'))
for name in ('step_kernel_table_c2', 'step_kernel_table_c4', 'step_kernel_table_c5'):
    f = os.path.join(pr, '%s_%s.txt' % (tag, name))
    if os.path.exists(f):
        say(name, open(f).read().strip().splitlines()[-1])
open(os.path.join(pr, tag + '_numbers.txt'), 'w').write('\n'.join(_lines) + '\n')
