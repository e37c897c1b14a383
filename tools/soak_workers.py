"""Soak for the scan workers: batches of different scans (shapes, depths, options) through 4 workers, every product hashed
and compared with the one-at-a-time order, round after round.  soak_workers.py [rounds] [workers]"""
import contextlib
import io
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
specs = [(300, 400, 32, 16, 3, {}), (260, 520, 40, 16, 4, {'shift': [-2, 0, 3]}), (300, 400, 32, 8, 5, {'flip_x': True}),
         (280, 32, 400, 16, 6, {'crop_width_square': True}), (300, 400, 32, 16, 7, {'transversalium': False}),
         (320, 480, 36, 16, 8, {'de-vignette': True}), (300, 400, 32, 16, 9, {'fixed_width': 300, 'img_rotate': 90}),
         (1000, 1200, 100, 16, 10, {}), (300, 400, 32, 16, 11, {'stubborn_transversalium': True, 'trans_strength': 41}),
         (700, 900, 64, 16, 12, {}), (500, 640, 48, 8, 13, {})]
stacks = [torch.from_numpy(synth.synth_frames_numpy(n, w, h, bits, seed=seed, tilt=0.01, curv=5e-5)).cuda()
          for n, w, h, bits, seed, _ in specs]


def run(order, n_workers):
    tasks = []
    for i in order:
        opts = SHG_MAIN.default_options()
        opts.update(specs[i][5], _nolog=True)
        tasks.append((array_reader(stacks[i]), opts))
    with contextlib.redirect_stdout(io.StringIO()):
        res = Solex_recon.solex_do_work(tasks, True, return_results=True, workers=n_workers)
    out = {}
    for i, per_file, (_, opts) in zip(order, res, tasks):
        out[i] = ([np.asarray(x) for pair in per_file for x in pair], (opts['ratio_fixe'], opts['slant_fix']))
    return out


base = run(list(range(len(specs))), 1)
bad = 0
rng = random.Random(0)
for r in range(rounds):
    order = list(range(len(specs))) * 2
    rng.shuffle(order)
    got = run(order, workers)
    for i in got:
        imgs, geo = got[i]
        ref_imgs, ref_geo = base[i]
        if geo != ref_geo:
            bad += 1
            print('round %d task %d: geometry %r != %r' % (r, i, geo, ref_geo), flush=True)
        for j, (a, b) in enumerate(zip(imgs, ref_imgs)):
            if a.shape != b.shape or not np.array_equal(a, b):
                bad += 1
                d = np.flatnonzero(a.ravel() != b.ravel()) if a.shape == b.shape else []
                print('round %d task %d product %d: %d pixels differ (first at %s, rows %s)' % (
                    r, i, j, len(d), np.unravel_index(d[0], a.shape) if len(d) else '-',
                    sorted(set((d // a.shape[1]).tolist()))[:8] if len(d) else '-'), flush=True)
print('%d rounds x %d scans through %d workers: %d mismatches' % (rounds, 2 * len(specs), workers, bad))
sys.exit(1 if bad else 0)
