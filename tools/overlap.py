"""How well do the scans in flight share the device?  From a rocprofv3 --kernel-trace output directory: over the last `scans` scans of
the trace (from the start of the pass A that is `scans` from the end), the fraction of the wall clock with 0 / 1 / 2 / ... kernels
running, and every kernel's average duration there (to put beside its duration alone, tools/ab_tables.sh).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ov -- python3 tools/pool_loop.py 40 4 -10,...,10
    python3 tools/overlap.py gpurun_out/ov 40"""
import collections
import csv
import glob
import os
import re
import sys

path = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
scans = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for r in csv.DictReader(open(path)):
    m = re.search(r'\bk_[a-z0-9_]+', r['Kernel_Name'])
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(0) if m else r['Kernel_Name'][:40]))
rows.sort()
t0 = [r[0] for r in rows if r[2].startswith('k_accumulate')][-scans]
rows = [r for r in rows if r[0] >= t0]
events = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
level, last, at = 0, events[0][0], collections.Counter()
for t, d in events:
    at[level] += t - last
    last = t
    level += d
wall = events[-1][0] - events[0][0]
n_a = sum(1 for r in rows if r[2].startswith('k_accumulate'))
print('%d kernels in %.1f ms, %d scans: %.3f ms per scan; sum of kernel durations %.3f ms per scan' %
      (len(rows), wall / 1e6, n_a, wall / 1e6 / max(n_a, 1), sum(r[1] - r[0] for r in rows) / 1e6 / max(n_a, 1)))
for k in sorted(at):
    print('  %d kernels running: %5.1f %% of the time' % (k, 100.0 * at[k] / wall))
per = collections.defaultdict(list)
for a, b, name in rows:
    per[name].append(b - a)
for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print('  %-34s %6.2f calls/scan  %8.1f us avg  %8.1f us/scan' % (name, len(d) / max(n_a, 1), sum(d) / len(d) / 1e3, sum(d) / 1e3 / max(n_a, 1)))
