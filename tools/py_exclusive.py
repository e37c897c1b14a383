"""Exclusive Python time (interpreter lock held) per function of a scan, one worker: wrappers with a call stack, the
ctypes stage calls counted as children.  py_exclusive.py [scans]"""
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, ellipse_to_circle as e2c, ops, solex_util, stages, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=0, padded=True)
torch.cuda.synchronize()
excl, calls, stack_ = {}, {}, []


def wrap(mod, name, label=None):
    fn = getattr(mod, name)
    label = label or name

    def inner(*a, **k):
        t0 = time.perf_counter()
        stack_.append(0.0)
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            child = stack_.pop()
            excl[label] = excl.get(label, 0.0) + dt - child
            calls[label] = calls.get(label, 0) + 1
            if stack_:
                stack_[-1] += dt
    setattr(mod, name, inner)


class Lib:
    def __init__(self, inner):
        self._inner, self._c = inner, {}

    def __getattr__(self, name):
        f = self._c.get(name)
        if f is None:
            raw = getattr(self._inner, name)

            def f(*a, _raw=raw, _n='C:' + name):
                t0 = time.perf_counter()
                try:
                    return _raw(*a)
                finally:
                    dt = time.perf_counter() - t0
                    excl[_n] = excl.get(_n, 0.0) + dt
                    calls[_n] = calls.get(_n, 0) + 1
                    if stack_:
                        stack_[-1] += dt
            self._c[name] = f
        return f


stages.lib = Lib(stages.lib)
ops.lib = Lib(ops.lib)
for mod, names in ((stages, ('mean_fit', 'extract', 'limb_fit', '_limb_call', 'process_frames', '_scratch')),
                   (ops, ('warp_rows_u16',)),
                   (solex_util, ('logme', 'clearlog', 'make_header')),
                   (Solex_recon, ('solex_read', 'solex_process', 'process_images', 'compute_mean_return_fit', 'extract_disks', 'ellipse_to_circle',
                                  'correct_image', 'logme', 'clearlog', 'make_header', 'crop_plan', 'write_complete')),
                   (e2c, ('_log_geometry', '_warp_geometry'))):
    for n in names:
        if hasattr(mod, n):
            wrap(mod, n, mod.__name__.split('.')[-1] + '.' + n)


def run(n):
    tasks = []
    for _ in range(n):
        o = SHG_MAIN.default_options()
        o.update(_nolog=True)
        tasks.append((array_reader(stack), o))
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        Solex_recon.solex_do_work(tasks, True, workers=1)
        return time.perf_counter() - t0


run(10)
excl.clear()
calls.clear()
wall = run(steps)
tot_py = sum(v for k, v in excl.items() if not k.startswith('C:'))
tot_c = sum(v for k, v in excl.items() if k.startswith('C:'))
print('wall %.1f us/scan; C calls %.1f; instrumented Python %.1f; the rest (scan loop, decoder hand-over, wrappers) %.1f' % (
    wall / steps * 1e6, tot_c / steps * 1e6, tot_py / steps * 1e6, (wall - tot_c - tot_py) / steps * 1e6))
for k, v in sorted(excl.items(), key=lambda kv: -kv[1]):
    if not k.startswith('C:'):
        print('  %-40s %5.1f calls/scan %7.1f us/scan' % (k, calls[k] / steps, v / steps * 1e6))
