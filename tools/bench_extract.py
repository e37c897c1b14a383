"""Time the extraction stage (shg_stage_extract) on a resident stack with HIP events: S = 2 and S = 21, rotated and
un-rotated files (run it under rocprofv3 --kernel-trace for the kernel alone: the events bracket the whole stage call).
Tried here in round 2 and dropped: a wide-load kernel for rotated files in which a lane owns four slit rows and fetches
them with one 8-byte request (three file rows per shift, left / right picked per row): a quarter of the load
instructions but 256-row x 64-frame tiles (68 KB of LDS for two shifts), i.e. one workgroup per CU -- 16.5 us against
13.3 us at C2 S=2 and 146 us against 100 us at S=21.  The extraction is bound by requests in flight per CU, not by the
number of load instructions."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import stages, synth  # noqa: E402


def timeit(fn, iters=30, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2], t[0]


# (the fifth: C2 / C4 with file rows of 4096 bytes -- what padding the rows of a 2000-px file to 128 bytes would give pass B: every 128-byte run on a line)
SHAPES = ((2000, 2000, 200, 16), (4000, 2560, 256, 16), (2000, 200, 2000, 16), (2000, 2000, 200, 8), (2000, 2048, 200, 16))
for (n, w, h, bits) in (SHAPES if len(sys.argv) < 2 else [SHAPES[int(sys.argv[1])]]):
    stack = synth.synth_frames_torch(n, w, h, bits, seed=0, padded=True)
    ih, iw = max(w, h), min(w, h)
    curve = synth.curve_of_row(np.arange(ih, dtype=np.float64), ih, iw)
    fit = np.stack([np.floor(curve), curve - np.floor(curve), np.arange(ih, dtype=float), curve], axis=1)
    for shifts in ([10, 0], [10, 0] + [s for s in range(-10, 11) if s not in (10, 0)]):
        out, mm = stages.extract(stack, fit, shifts, want_minmax=True)
        med, best = timeit(lambda: stages.extract(stack, fit, shifts, out=out, want_minmax=True))
        s = len(shifts)
        u = 4 if s == 2 else s + 1
        alg = n * ih * (u * stack.element_size() + 2 * s) / 1e9
        print('%dx%dx%d %d-bit S=%-2d  %.1f us median  %.1f us best (events, stage incl. plan upload + fold)  algorithmic %.3f GB -> %.2f TB/s' % (
            n, w, h, bits, s, med * 1e3, best * 1e3, alg, alg / med))
    del stack
