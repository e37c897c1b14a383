"""How much do the kernels of concurrent scans overlap on the GPU?  overlap.py <rocprofv3 output dir>
Reads *_kernel_trace.csv: sum of kernel durations, union of busy intervals, and what runs while pass A runs."""
import csv
import glob
import os
import sys

path = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(path)))
ev = []
for r in rows:
    name = r.get('Kernel_Name') or r.get('Name')
    if '(anonymous namespace)::k_' not in name:
        continue
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name.split('::')[-1].split('(')[0], r.get('Queue_Id', '?')))
ev.sort()
# the densest part: the last 60 % of pass A launches
acc = [e for e in ev if e[2].startswith('k_accumulate')]
t0, t1 = acc[len(acc) * 4 // 10][0], acc[-1][1]
sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
total = sum(e[1] - e[0] for e in sel)
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
n_acc = sum(1 for e in sel if e[2].startswith('k_accumulate'))
print('window %.2f ms, %d scans: kernel time %.2f ms, GPU busy (union) %.2f ms, idle %.2f ms' % ((t1 - t0) / 1e6, n_acc, total / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
print('per scan: wall %.1f us, kernel sum %.1f us, union %.1f us, overlap factor %.2f' % ((t1 - t0) / 1e3 / n_acc, total / 1e3 / n_acc, busy / 1e3 / n_acc, total / busy))
queues = sorted(set(e[3] for e in sel))
print('queues used:', queues)
# while pass A runs: time covered by other kernels, and pass A's duration alone vs overlapped
alone, shared = [], []
others = [e for e in sel if not e[2].startswith('k_accumulate')]
for s, e, n, q in sel:
    if not n.startswith('k_accumulate'):
        continue
    ov = sum(max(0, min(e, oe) - max(s, os_)) for os_, oe, _, _ in others if oe > s and os_ < e)
    (shared if ov > 0.2 * (e - s) else alone).append(((e - s) / 1e3, ov / 1e3))
for tag, v in (('pass A mostly alone', alone), ('pass A with company', shared)):
    if v:
        print('%-22s n=%3d  mean duration %.1f us, other kernels overlapping it %.1f us' % (tag, len(v), sum(a for a, _ in v) / len(v), sum(b for _, b in v) / len(v)))

# what sharing the device with a pass A does to the small kernels: mean duration of every kernel of the per-file chain when it
# runs entirely inside a pass A of another scan, and when no pass A is running at all
acc_iv = [(s, e) for s, e, n, _ in sel if n.startswith('k_accumulate')]


def cover(s, e):
    return sum(max(0, min(e, ae) - max(s, as_)) for as_, ae in acc_iv if ae > s and as_ < e)


per = {}
for s, e, n, q in others:
    c = cover(s, e) / max(e - s, 1)
    slot = per.setdefault(n, [[], [], []])
    slot[0 if c > 0.9 else (1 if c < 0.1 else 2)].append((e - s) / 1e3)
print('%-34s %6s %9s %6s %9s %7s' % ('kernel', 'n', 'beside A', 'n', 'no A', 'ratio'))
tot_in = tot_out = 0.0
for n, (a, b, _) in sorted(per.items(), key=lambda kv: -sum(kv[1][0] + kv[1][1] + kv[1][2])):
    if a and b:
        ma, mb = sum(a) / len(a), sum(b) / len(b)
        print('%-34s %6d %9.1f %6d %9.1f %7.2f' % (n[:34], len(a), ma, len(b), mb, ma / mb))
        per_scan = (len(a) + len(b)) / max(n_acc, 1)
        tot_in += ma * per_scan
        tot_out += mb * per_scan
print('chain per scan: %.1f us if all of it ran beside a pass A, %.1f us if none did' % (tot_in, tot_out))
