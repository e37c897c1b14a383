"""apply_clahe: the numeric part of the reference's stand-alone CLAHE tool
(clahe_apply.py:243-256).  Reads an 8/16-bit grayscale PNG, applies
createCLAHE(0.8, (tile, tile)) on the GPU, optionally stretches between two percentiles,
writes <name>_clahe.png.  The GUI around it (clahe_apply.py:17-241) is out of scope.
"""
import os

import numpy as np
import torch

from . import ops, png_io
from .device import default_device
from .solex_util import percentile_from_hist, rescale_brightness

options = {'workDir': '', 'language': 'English', 'lo': 0, 'hi': 100, 'do_stretch': False, 'sat': 80, 'tile_size': 2}


def apply_clahe(file, options, write_file=True):
    frame = png_io.read_png_gray(file) if isinstance(file, str) else np.asarray(file)
    if frame.ndim != 2 or frame.dtype not in (np.uint8, np.uint16):
        raise TypeError('apply_clahe expects an 8- or 16-bit grayscale image')
    dev = torch.from_numpy(np.ascontiguousarray(frame)).to(default_device())
    cl1_t = ops.clahe(dev, 0.8, options['tile_size'])
    hist = ops.histogram(dev).cpu().numpy()
    dark = percentile_from_hist(hist, options['lo'])
    bright = percentile_from_hist(hist, options['hi'])
    if options['do_stretch']:
        out = rescale_brightness(cl1_t, dark, bright, alpha=options['sat'] / 100)
        cl1 = out.cpu().numpy() if isinstance(out, torch.Tensor) else np.asarray(out)
    else:
        cl1 = cl1_t.cpu().numpy()
    if write_file and isinstance(file, str):
        print('save:', os.path.splitext(file)[0] + '_clahe.png')
        png_io.write_png(os.path.splitext(file)[0] + '_clahe.png', cl1, 1)
    return cl1
