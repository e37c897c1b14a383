"""Where the time of a batch goes: K scans of a resident stack with W scan workers, every stage call timed on the host.
scan_timeline.py [steps] [workers] [warmup]   (prints wall per scan, per-stage host time, per-worker busy fraction)"""
import collections
import contextlib
import io
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, stages, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 5
stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=0, padded=True)
torch.cuda.synchronize()
log = []
lock = threading.Lock()


def timed(name, fn):
    def inner(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            t1 = time.perf_counter()
            with lock:
                log.append((threading.current_thread().name, name, t0, t1))
    return inner


for name in ('mean_fit', 'extract', 'limb_fit', 'process_frames'):
    setattr(stages, name, timed(name, getattr(stages, name)))


class _TimedLib:
    """The ctypes library with the stage composites timed: the time inside the C call (no interpreter lock held)."""
    def __init__(self, inner):
        self._inner, self._cache = inner, {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            fn = getattr(self._inner, name)
            if name.startswith('shg_stage_') and not name.endswith('_bytes'):
                fn = timed('C:' + name[10:], fn)
            self._cache[name] = fn
        return fn


stages.lib = _TimedLib(stages.lib)
threading.Thread.start = timed('P:Thread.start', threading.Thread.start)
torch.cuda.set_device = timed('P:set_device', torch.cuda.set_device)
Solex_recon.default_device = timed('P:default_device', Solex_recon.default_device)
_orig_run = Solex_recon._Decoder._run
Solex_recon._Decoder._run = timed('P:decoder._run', _orig_run)
for name in ('_Decoder', '_scan_pool', 'cpu_plan', '_worker_context', 'bind_thread'):
    setattr(Solex_recon, name, timed('P:' + name, getattr(Solex_recon, name)))
Solex_recon.solex_read = timed('solex_read', Solex_recon.solex_read)
Solex_recon.solex_process = timed('solex_process', Solex_recon.solex_process)


def batch(n):
    tasks = []
    for _ in range(n):
        opts = SHG_MAIN.default_options()
        opts.update(_nolog=True)
        tasks.append((array_reader(stack), opts))
    with contextlib.redirect_stdout(io.StringIO()):
        Solex_recon.solex_do_work(tasks, True, distribute='none', workers=workers)


batch(warmup)
if os.environ.get('GC') == 'off':
    import gc
    gc.collect()
    gc.disable()
elif os.environ.get('GC') == 'freeze':
    import gc
    gc.collect()
    gc.freeze()
if os.environ.get('GC') == 'cb':
    import gc
    gc.callbacks.append(lambda phase, info: phase == 'stop' and log.append(('gc', 'gc gen%d' % info['generation'], time.perf_counter(), time.perf_counter())))
for rep in range(int(os.environ.get('REPS', 3))):
    torch.cuda.synchronize()
    del log[:]
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    batch(steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = collections.defaultdict(list)
    by_thread = collections.defaultdict(list)
    for th, name, a, b in log:
        per[name].append(b - a)
        if name in ('solex_read', 'solex_process'):
            by_thread[th].append((a - t0, b - t0))
    ms1 = torch.cuda.memory_stats()
    print('rep %d: %d scans, %d workers: %.3f ms/scan wall   device mallocs %d frees %d, allocator requests %d, reserved %.0f MB' % (
        rep, steps, workers, wall / steps * 1e3, ms1['num_device_alloc'] - ms0['num_device_alloc'], ms1['num_device_free'] - ms0['num_device_free'],
        ms1['allocation.all.allocated'] - ms0['allocation.all.allocated'], ms1['reserved_bytes.all.current'] / 1e6))
    slow = sorted((b - a, th, name, a - t0) for th, name, a, b in log if (name in ('mean_fit', 'extract', 'limb_fit', 'process_frames') or name.startswith('C:')) and b - a > 2e-3)
    if os.environ.get('PRE'):
        for th, name, a, b in sorted(log, key=lambda r: r[2]):
            if name.startswith('P:') or (name == 'solex_read' and a - t0 < 8e-3):
                print('   %-12s %-18s %7.2f -> %7.2f ms' % (th, name, (a - t0) * 1e3, (b - t0) * 1e3))
    for th, name, a, b in log:
        if th == 'gc':
            print('   GC %s finished at %.2f ms' % (name, (a - t0) * 1e3))
    for d, th, name, at in slow[-10:]:
        print('   SLOW %-14s %7.2f ms on %s at %.2f ms' % (name, d * 1e3, th, at * 1e3))
    for name in ('solex_read', 'mean_fit', 'extract', 'solex_process', 'limb_fit', 'process_frames'):
        v = per[name]
        if v:
            print('   %-15s n=%3d  mean %7.1f us  min %7.1f  max %7.1f' % (name, len(v), sum(v) / len(v) * 1e6, min(v) * 1e6, max(v) * 1e6))
    for th in sorted(by_thread):
        iv = sorted(by_thread[th])
        busy = sum(b - a for a, b in iv)
        print('   %-12s first start %6.2f ms  last end %6.2f ms  in stage calls %5.2f ms (%d calls)' % (th, iv[0][0] * 1e3, iv[-1][1] * 1e3, busy * 1e3, len(iv)))
