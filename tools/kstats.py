"""Print (calls, average us) of the kernels whose name contains argv[2] from a rocprofv3 --kernel-trace --stats output directory."""
import csv
import glob
import os
import sys

path = max(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(path)):
    if sys.argv[2] in r['Name']:
        print('  %-70s %5d calls  %8.1f us avg' % (r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70], int(r['Calls']), float(r['AverageNs']) / 1e3))
