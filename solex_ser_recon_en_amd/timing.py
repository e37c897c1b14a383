"""Optional host wall-clock accounting per pipeline stage (bench.py prints it).  Disabled
by default; when enabled every stage ends with a synchronisation of the calling thread's
stream so that GPU work is charged to the stage that launched it.  Safe to use from several
scan workers at once (each waits for its own stream only; the totals are guarded), though the
figures then include whatever the other workers' kernels cost this one."""
import threading
import time
from contextlib import contextmanager

enabled = False
totals = {}
_lock = threading.Lock()


def reset():
    with _lock:
        totals.clear()


@contextmanager
def stage(name):
    if not enabled:
        yield
        return
    import torch
    torch.cuda.current_stream().synchronize()
    t0 = time.perf_counter()
    try:
        yield
    finally:
        torch.cuda.current_stream().synchronize()
        dt = time.perf_counter() - t0
        with _lock:
            totals[name] = totals.get(name, 0.0) + dt
