"""The device's timeline of a pipelined batch WITHOUT a profiler attached: every library entry point of `steps` scans
(4 scan workers by default) bracketed by HIP events on its own stream (shg_profile_*), dumped and analysed --
busy / idle time of the device, how many entry points run at once, what runs beside a pass A and how long it takes there.
    python3 tools/timeline.py [steps] [workers]
The events cost about a microsecond each; unlike rocprofv3 they do not slow the launches down (rocprofv3: 0.8 ms per scan
where the plain run takes 0.5)."""
import contextlib
import csv
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, _lib, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
stacks = [synth.synth_frames_torch(2000, 2000, 200, 16, seed=j, padded=True) for j in range(workers + 1)]
torch.cuda.synchronize()


def batch(n):
    tasks = []
    for i in range(n):
        opts = SHG_MAIN.default_options()
        opts.update(_nolog=True)
        tasks.append((array_reader(stacks[i % len(stacks)]), opts))
    with contextlib.redirect_stdout(io.StringIO()):
        Solex_recon.solex_do_work(tasks, True, distribute='none', workers=workers)
    torch.cuda.synchronize()


batch(8)
import gc  # noqa: E402
gc.collect()
gc.freeze()
t0 = time.perf_counter()
batch(steps)
plain = (time.perf_counter() - t0) / steps * 1e3
_lib.profile_reset()
_lib.profile_enable(True)
t0 = time.perf_counter()
batch(steps)
timed = (time.perf_counter() - t0) / steps * 1e3
_lib.profile_enable(False)
out = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'timeline.csv')
os.makedirs(os.path.dirname(out), exist_ok=True)
_lib.check(_lib.lib.shg_profile_dump(out.encode()), 'shg_profile_dump')
rows = [(r['tag'], r['stream'], float(r['start_ms']) * 1e3, float(r['stop_ms']) * 1e3) for r in csv.DictReader(open(out))]
rows.sort(key=lambda r: r[2])
acc = [r for r in rows if r[0] == 'accumulate']
lo, hi = acc[len(acc) // 5][2], acc[-1][3]                 # steady state: skip the fill
sel = [r for r in rows if r[2] >= lo and r[3] <= hi]
n_scans = sum(1 for r in sel if r[0] == 'accumulate')
print('%d scans, %d workers: %.3f ms per scan plain, %.3f with every entry point bracketed by events' % (steps, workers, plain, timed))
edges = sorted([(r[2], 1) for r in sel] + [(r[3], -1) for r in sel])
depth, last, hist = 0, lo, {}
for t, d in edges:
    hist[depth] = hist.get(depth, 0.0) + (t - last)
    last = t
    depth += d
span = hi - lo
print('window %.1f ms, %d scans: %.1f us per scan; entry points in flight: ' % (span / 1e3, n_scans, span / n_scans) +
      ', '.join('%d: %.0f%%' % (k, 100 * v / span) for k, v in sorted(hist.items())))
acc_iv = [(r[2], r[3]) for r in sel if r[0] == 'accumulate']
lane_busy = sum(e - s for s, e in acc_iv)
print('pass A: %.1f us average in flight, lane busy %.0f%% of the window' % (lane_busy / len(acc_iv), 100 * lane_busy / span))


def cover(s, e):
    return sum(max(0.0, min(e, ae) - max(s, as_)) for as_, ae in acc_iv if ae > s and as_ < e)


per = {}
for tag, _, s, e in sel:
    if tag == 'accumulate':
        continue
    c = cover(s, e) / max(e - s, 1e-9)
    per.setdefault(tag, [[], [], []])[0 if c > 0.9 else (1 if c < 0.1 else 2)].append(e - s)
print('%-22s %5s %9s %5s %9s %6s   per scan' % ('entry point', 'n', 'beside A', 'n', 'no A', 'ratio'))
tot = [0.0, 0.0, 0.0]
for tag, (a, b, m) in sorted(per.items(), key=lambda kv: -sum(sum(x) for x in kv[1])):
    ma = sum(a) / len(a) if a else float('nan')
    mb = sum(b) / len(b) if b else float('nan')
    total = (sum(a) + sum(b) + sum(m)) / n_scans
    print('%-22s %5d %9.1f %5d %9.1f %6.2f   %7.1f us' % (tag[:22], len(a), ma, len(b), mb, ma / mb if a and b else float('nan'), total))
    calls = (len(a) + len(b) + len(m)) / n_scans
    if a and b:
        tot[0] += ma * calls
        tot[1] += mb * calls
    tot[2] += total
print('chain per scan: %.1f us as it ran, %.1f us if all of it ran beside a pass A, %.1f us if none did' % (tot[2], tot[0], tot[1]))
# per stream: how long a worker's stream is empty between two of its entry points (host time: control plane, launches, waking up)
by_stream = {}
for tag, st, s, e in sel:
    by_stream.setdefault(st, []).append((s, e, tag))
for st, ev in by_stream.items():
    ev.sort()
    busy = sum(e - s for s, e, _ in ev)
    gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
    big = sorted(((g, ev[i][2], ev[i + 1][2]) for i, g in enumerate(gaps) if g > 20), reverse=True)
    print('stream %s: %d entry points, busy %.0f%%, gaps > 20 us: %d, their sum %.0f us per scan of this stream' % (
        st, len(ev), 100 * busy / span, len(big), sum(g for g, _, _ in big) / max(1, sum(1 for x in ev if x[2] == 'extract'))))
