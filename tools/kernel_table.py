"""Per-step kernel table from a rocprofv3 --kernel-trace --stats run of tools/step_loop.py."""
import csv
import glob
import sys

import os

path = max(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
steps = int(sys.argv[2])
rows = [r for r in csv.DictReader(open(path)) if '(anonymous namespace)::k_' in r['Name'] and 'at::native' not in r['Name'] and int(r['Calls']) >= steps]
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    per_step = float(r['TotalDurationNs']) / steps / 1e3
    tot += per_step
    name = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print('%-46s %5.1f calls/step  %8.1f us avg  %8.1f us/step' % (name[:46], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, per_step))
print('library kernels per step: %.1f us' % tot)
