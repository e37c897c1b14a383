"""Where a scan's wall time goes in ONE thread: inside the C calls (no interpreter lock held) vs in Python
(lock held: what serialises the scan workers).  Prints per-call totals per scan."""
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, _lib, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
shifts = [int(s) for s in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=0, padded=True)
torch.cuda.synchronize()

acc = {}
gaps = None
last = [0.0]


class Timed:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.fn(*a)
        t1 = time.perf_counter()
        d = t1 - t
        e = acc.setdefault(self.name, [0.0, 0])
        e[0] += d
        e[1] += 1
        if gaps is not None and self.name.startswith(('shg_stage', 'shg_warp')):
            g = gaps.setdefault('before ' + self.name, [0.0, 0])
            g[0] += t - last[0]
            g[1] += 1
            last[0] = t1
        return r


class Proxy:
    def __init__(self, lib):
        self._lib = lib
        self._cache = {}

    def __getattr__(self, name):
        if name not in self._cache:
            self._cache[name] = Timed(name, getattr(self._lib, name))
        return self._cache[name]


proxy = Proxy(_lib.lib)
import solex_ser_recon_en_amd.ops as ops_mod  # noqa: E402
import solex_ser_recon_en_amd.stages as stages_mod  # noqa: E402
ops_mod.lib = proxy
stages_mod.lib = proxy


def run(n):
    tasks = []
    for _ in range(n):
        o = SHG_MAIN.default_options()
        o.update(_nolog=True, shift=list(shifts))
        tasks.append((array_reader(stack), o))
    with contextlib.redirect_stdout(io.StringIO()):
        Solex_recon.solex_do_work(tasks, True, workers=1)
    torch.cuda.synchronize()


run(5)
acc.clear()
gaps = {}
t0 = time.perf_counter()
last[0] = t0
run(steps)
wall = time.perf_counter() - t0
for k, v in gaps.items():
    print('  Python %-42s %7.1f us/scan' % (k, v[0] / steps * 1e6))
gaps = None
in_c = sum(v[0] for v in acc.values())
print('wall %.3f ms per scan; inside C calls %.3f ms; Python (interpreter lock held) %.3f ms' % (
    wall / steps * 1e3, in_c / steps * 1e3, (wall - in_c) / steps * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print('  %-40s %6.1f calls/scan %9.1f us/scan' % (k, v[1] / steps, v[0] / steps * 1e6))

for workers in (2, 4, 8):
    def run_w(n):
        tasks = []
        for _ in range(n):
            o = SHG_MAIN.default_options()
            o.update(_nolog=True, shift=list(shifts))
            tasks.append((array_reader(stack), o))
        with contextlib.redirect_stdout(io.StringIO()):
            Solex_recon.solex_do_work(tasks, True, workers=workers)
        torch.cuda.synchronize()
    run_w(4 * workers)
    acc.clear()
    t0 = time.perf_counter()
    run_w(steps * 2)
    wall = time.perf_counter() - t0
    n = steps * 2
    in_c = sum(v[0] for v in acc.values())
    print('\n%d workers: wall %.3f ms per scan (%.0f scans/s); summed over threads: inside C calls %.3f ms per scan, thread time %.3f ms per scan '
          '-> Python + lock waits %.3f ms per scan' % (workers, wall / n * 1e3, n / wall, in_c / n * 1e3, wall * workers / n * 1e3,
                                                      (wall * workers - in_c) / n * 1e3))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:5]:
        print('  %-40s %6.1f calls/scan %9.1f us/scan' % (k, v[1] / n, v[0] / n * 1e6))
