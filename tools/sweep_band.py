"""Pass B on rotated files: the band kernel against the general kernel (SHG_EXT_GENERAL=1 / SHG_EXT_DENSE=0) on C4's, C2's and C5's
shapes and an 8-bit file, the disks and extrema compared bit for bit; kernel time from the library's own events around the launch
(SHG_PROF tag "extract").  Usage: python3 tools/sweep_band.py [case ...]   case = c4 | c2 | c5 | u8 | u8s21 | list5"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import _lib, stages, synth  # noqa: E402

S21 = [10, 0] + [s for s in range(-10, 11) if s not in (10, 0)]
CASES = {
    'c4': (2000, 2000, 200, 16, S21),
    'c2': (2000, 2000, 200, 16, [10, 0]),
    'c5': (4000, 2560, 256, 16, [10, 0]),
    'u8': (2000, 2000, 200, 8, [10, 0]),
    'u8s21': (2000, 2000, 200, 8, S21),
    'list5': (2000, 2000, 200, 16, [10, 0, -7, 3, 25]),
}


def run(stack, fit, shifts, label, alg, iters=40):
    out, mm = stages.extract(stack, fit, shifts, want_minmax=True)
    for _ in range(5):
        stages.extract(stack, fit, shifts, out=out, want_minmax=True)
    torch.cuda.synchronize()
    _lib.profile_enable(True, only=['extract'])
    _lib.profile_reset()
    for _ in range(iters):
        stages.extract(stack, fit, shifts, out=out, want_minmax=True)
    torch.cuda.synchronize()
    ms, k = _lib.profile_get('extract')
    _lib.profile_enable(False)
    us = ms / max(k, 1) * 1e3
    print('%-34s %7.1f us  %.2f TB/s of algorithmic bytes (%.3f of 8)' % (label, us, alg / us * 1e3, alg / us * 1e3 / 8), flush=True)
    return out.clone(), mm.clone()


for case in (sys.argv[1:] or ['c4', 'c2', 'c5', 'u8', 'u8s21', 'list5']):
    n, w, h, bits, shifts = CASES[case]
    stack = synth.synth_frames_torch(n, w, h, bits, seed=0, padded=True)
    ih, iw = max(w, h), min(w, h)
    curve = synth.curve_of_row(np.arange(ih, dtype=np.float64), ih, iw)
    fit = np.stack([np.floor(curve), curve - np.floor(curve), np.arange(ih, dtype=float), curve], axis=1)
    s = len(shifts)
    u = len({c for sh in shifts for c in (sh, sh + 1)})
    alg = n * ih * (u * stack.element_size() + 2 * s) / 1e9
    os.environ['SHG_EXT_DENSE'] = '0'
    os.environ['SHG_EXT_GENERAL'] = '1'
    ref, ref_mm = run(stack, fit, shifts, case + ' general kernel', alg)
    os.environ.pop('SHG_EXT_DENSE')
    os.environ.pop('SHG_EXT_GENERAL')
    out, mm = run(stack, fit, shifts, case + ' band kernel', alg)
    print('    == general kernel: %s' % (torch.equal(out.view(torch.int16), ref.view(torch.int16)) and torch.equal(mm, ref_mm)), flush=True)
    del stack, ref, out
