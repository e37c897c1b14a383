"""Where does a scan worker's wall clock go?  `steps` scans through `workers` scan workers with the library's host-side
section timer on (shg_host_timing_*): per scan, the time inside shg_scan_file, inside each stage composite, waiting in each
stream synchronise and inside each control-plane routine -- and what is left for the interpreter between two calls.
    python3 tools/host_budget.py [steps] [workers] [shifts a,b,c]"""
import contextlib
import ctypes
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, _lib, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
shifts = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0]
stacks = [synth.synth_frames_torch(2000, 2000, 200, 16, seed=j, padded=True) for j in range(5)]
torch.cuda.synchronize()


def batch(n, workers):
    tasks = []
    for i in range(n):
        opts = SHG_MAIN.default_options()
        opts.update(_nolog=True, shift=list(shifts))
        tasks.append((array_reader(stacks[i % len(stacks)]), opts))
    with contextlib.redirect_stdout(io.StringIO()):
        Solex_recon.solex_do_work(tasks, True, distribute='none', workers=workers)
    torch.cuda.synchronize()


# the ONE interpreter thread that feeds the native pool: what it spends per scan preparing, submitting and finishing (waiting apart)
main_thread = {}


def clocked(cls, name, tag):
    inner = getattr(cls, name)

    def outer(*a, **kw):
        t = time.perf_counter()
        try:
            return inner(*a, **kw)
        finally:
            main_thread[tag] = main_thread.get(tag, 0.0) + time.perf_counter() - t
    setattr(cls, name, outer)


from solex_ser_recon_en_amd import stages as _stages  # noqa: E402
clocked(Solex_recon._OneCall, '__init__', 'prepare (_OneCall.__init__)')
clocked(Solex_recon._OneCall, 'submit', 'submit')
clocked(Solex_recon._OneCall, 'finish', 'finish, waiting included')
clocked(_stages.ScanCall, 'wait', 'waiting for the pool (ScanCall.wait)')

import gc  # noqa: E402
for workers in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 4]):
    batch(8, workers)
    gc.collect()
    gc.freeze()
    _lib.lib.shg_host_timing_enable(1)
    main_thread.clear()
    t0 = time.perf_counter()
    batch(steps, workers)
    wall = time.perf_counter() - t0
    _lib.lib.shg_host_timing_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.lib.shg_host_timing_report(buf, len(buf))
    print('== %d workers: %.3f ms per scan; a worker\'s cycle %.3f ms' % (workers, wall / steps * 1e3, wall / steps * 1e3 * workers))
    rows = []
    for line in buf.value.decode().splitlines():
        tag, sec, calls = line.rsplit(' ', 2)
        rows.append((tag, float(sec), int(calls)))
    for tag, sec, calls in sorted(rows, key=lambda r: -r[1]):
        print('  %-58s %8.1f us per scan  (%5.2f calls, %7.1f us each)' % (tag, sec / steps * 1e6, calls / steps, sec / calls * 1e6))
    if main_thread:
        print('  -- the feeding thread, per scan:')
        for tag, sec in sorted(main_thread.items(), key=lambda kv: -kv[1]):
            print('     %-55s %8.1f us' % (tag, sec / steps * 1e6))
        busy = sum(v for k, v in main_thread.items() if 'wait' not in k) + main_thread.get('finish, waiting included', 0.0) \
            - main_thread.get('waiting for the pool (ScanCall.wait)', 0.0)
        print('     %-55s %8.1f us of %.1f us per scan' % ('busy (everything but waiting)', busy / steps * 1e6, wall / steps * 1e6))
    scan = dict((t, s) for t, s, _ in rows).get('scan_file', 0.0)
    print('  outside shg_scan_file (interpreter, allocation, waiting for a task): %.1f us per scan' % ((wall * workers - scan) / steps * 1e6))
