"""Pass B on a rotated Doppler stack (C4's shape): the band kernel's launch shapes against the general kernel, every one
checked against the general kernel's disks bit for bit; kernel time from the library's own events around the launch
(SHG_PROF tag "extract").  Usage: python3 tools/sweep_band.py [shape ...]   shape = G,DK,NW[,dbg]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import _lib, stages, synth  # noqa: E402

n, w, h, bits = 2000, 2000, 200, 16
if os.environ.get('SWEEP_SHAPE'):
    n, w, h, bits = (int(x) for x in os.environ['SWEEP_SHAPE'].split(','))
stack = synth.synth_frames_torch(n, w, h, bits, seed=0, padded=True)
ih, iw = max(w, h), min(w, h)
curve = synth.curve_of_row(np.arange(ih, dtype=np.float64), ih, iw)
fit = np.stack([np.floor(curve), curve - np.floor(curve), np.arange(ih, dtype=float), curve], axis=1)
shifts = [10, 0] + [s for s in range(-10, 11) if s not in (10, 0)]
s = len(shifts)
alg = n * ih * ((s + 1) * stack.element_size() + 2 * s) / 1e9


def run(label, iters=40):
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
    print('%-28s %7.1f us  %.2f TB/s of algorithmic bytes (%.3f of 8)' % (label, us, alg / us * 1e3 / 1e3, alg / us / 8 * 1e3 / 1e3), flush=True)
    return out, mm


os.environ['SHG_EXT_DENSE'] = '0'
ref, ref_mm = run('general kernel')
ref = ref.clone()
ref_mm = ref_mm.clone()
shapes = sys.argv[1:] or ['7,64,4', '7,64,8', '11,64,8', '5,64,4', '4,64,4', '3,64,4', '3,128,8', '5,128,8', '21,32,8', '7,32,4', '11,32,4']
os.environ['SHG_EXT_DENSE'] = '2'
for shp in shapes:
    os.environ['SHG_EXT_BAND'] = shp
    try:
        out, mm = run('band ' + shp)
    except Exception as e:  # noqa: BLE001
        print('band %s: %s' % (shp, e))
        continue
    if len(shp.split(',')) < 4 or shp.split(',')[3] == '0':
        same = torch.equal(out.view(torch.int16), ref.view(torch.int16)) and torch.equal(mm, ref_mm)
        print('    == general kernel: %s' % same, flush=True)
