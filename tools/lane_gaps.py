"""How busy is the frame-pass lane?  A batch of scans through the native pool with HIP events around every pass A (the library's own
profile scopes: no profiler attached); prints the passes' durations and the idle gaps between consecutive passes on the lane.
    python3 tools/lane_gaps.py [steps] [workers]"""
import contextlib
import io
import os
import sys
import tempfile
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


batch(3 * workers)
import gc  # noqa: E402
gc.collect()
gc.freeze()
_lib.profile_reset()
_lib.profile_enable(True, only=('accumulate',))
t0 = time.perf_counter()
batch(steps)
wall = time.perf_counter() - t0
_lib.profile_enable(False)
path = os.path.join(tempfile.gettempdir(), 'lane_%d.csv' % os.getpid())
_lib.check(_lib.lib.shg_profile_dump(path.encode()), 'shg_profile_dump')
rows = [ln.split(',') for ln in open(path).read().splitlines()[1:]]
os.remove(path)
spans = sorted((float(r[2]), float(r[3])) for r in rows if r[0] == 'accumulate')
dur = [b - a for a, b in spans]
gaps = [spans[i + 1][0] - spans[i][1] for i in range(len(spans) - 1)]
print('%d scans, %d workers: %.3f ms per scan wall; pass A %.3f ms mean (min %.3f max %.3f); lane gaps mean %.3f ms (min %.3f max %.3f); lane busy %.0f%% of first start -> last end'
      % (steps, workers, wall / steps * 1e3, sum(dur) / len(dur), min(dur), max(dur), sum(gaps) / len(gaps), min(gaps), max(gaps),
         100 * sum(dur) / (spans[-1][1] - spans[0][0])))
print('gaps (ms):', ' '.join('%.3f' % g for g in gaps))
print('pass A (ms):', ' '.join('%.3f' % d for d in dur))
