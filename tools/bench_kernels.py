"""Micro-benchmark of the two frame passes on a device-resident stack (tuning aid)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import ops, synth  # noqa: E402


def timeit(fn, iters=20, warmup=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=2000)
    ap.add_argument('--w', type=int, default=2000)
    ap.add_argument('--h', type=int, default=200)
    ap.add_argument('--bits', type=int, default=16)
    ap.add_argument('--shifts', type=int, default=2)
    ap.add_argument('--dense', action='store_true', help='dense frames instead of the 8 KiB-rounded frame pitch')
    a = ap.parse_args()
    stack = synth.synth_frames_torch(a.n, a.w, a.h, a.bits, seed=0, padded=not a.dense)
    n, h, w = stack.shape
    bpp = stack.element_size()
    ih, iw = max(h, w), min(h, w)
    gb = n * h * w * bpp / 1e9
    ws = torch.empty(ops.lib.shg_accumulate_workspace_bytes(n, h, w, bpp), dtype=torch.uint8, device='cuda')
    med, best = timeit(lambda: ops.accumulate_sum_max(stack, ws))
    print('pass A  (sum/max)   %.3f ms median  %.3f ms best  -> %.2f TB/s median (%.1f%% of 8 TB/s)  env: %s' % (
        med, best, gb / med, 100 * gb / med / 8, {k: v for k, v in os.environ.items() if k.startswith('SHG_')}))
    curve = synth.curve_of_row(np.arange(ih, dtype=np.float64), ih, iw)
    shifts = [10, 0] + [s for s in range(-10, 11) if s not in (10, 0)]
    shifts = shifts[:a.shifts]
    ind_l = np.clip(np.floor(curve)[None, :] + np.array(shifts)[:, None], 0, iw - 2).astype(np.int32)
    frac = curve - np.floor(curve)
    lw, rw = 1 - frac, 1 - (1 - frac)
    ind_d, lw_d, rw_d = torch.from_numpy(ind_l).cuda(), torch.from_numpy(lw).cuda(), torch.from_numpy(rw).cuda()
    out = ops.extract_columns(stack, ind_d, lw_d, rw_d)
    med, best = timeit(lambda: ops.extract_columns(stack, ind_d, lw_d, rw_d, out=out))
    u = 4 if a.shifts == 2 else 2 * a.shifts
    alg = n * ih * (u * bpp + 2 * len(shifts)) / 1e9
    print('pass B  (extract S=%d) %.3f ms median  %.3f ms best  -> algorithmic %.3f GB -> %.2f TB/s' % (
        len(shifts), med, best, alg, alg / med))


if __name__ == '__main__':
    main()
