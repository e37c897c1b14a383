"""`steps` scans through `workers` scan workers of the native pool, nothing else after the warm-up batch: what tools/overlap.py looks at
under rocprofv3 --kernel-trace.    python3 tools/pool_loop.py [steps] [workers] [shifts a,b,c]"""
import contextlib
import gc
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
shifts = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0]
stacks = [synth.synth_frames_torch(2000, 2000, 200, 16, seed=j, padded=True) for j in range(5)]
torch.cuda.synchronize()


def batch(n):
    tasks = []
    for i in range(n):
        opts = SHG_MAIN.default_options()
        opts.update(_nolog=True, shift=list(shifts))
        tasks.append((array_reader(stacks[i % len(stacks)]), opts))
    with contextlib.redirect_stdout(io.StringIO()):
        Solex_recon.solex_do_work(tasks, True, distribute='none', workers=workers)
    torch.cuda.synchronize()


batch(max(8, 2 * (workers + 2)))
gc.collect()
gc.freeze()
t0 = time.perf_counter()
batch(steps)
print('%d scans, %d workers, %d requested shifts: %.3f ms per scan' % (steps, workers, len(shifts), (time.perf_counter() - t0) / steps * 1e3))
