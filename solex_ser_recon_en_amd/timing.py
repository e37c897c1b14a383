"""Optional host wall-clock accounting per pipeline stage (bench.py prints it).  Disabled
by default; when enabled every stage ends with a device synchronisation so that GPU work
is charged to the stage that launched it."""
import time
from contextlib import contextmanager

enabled = False
totals = {}


def reset():
    totals.clear()


@contextmanager
def stage(name):
    if not enabled:
        yield
        return
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        yield
    finally:
        torch.cuda.synchronize()
        totals[name] = totals.get(name, 0.0) + (time.perf_counter() - t0)
