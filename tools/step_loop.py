"""N scans of a resident stack through the production entry point, one at a time, and nothing else (rocprofv3 target: per-kernel
time of one scan).  step_loop.py [steps] [shifts a,b,c] [frames width height bits]
SHG_STEP_SEED picks the synthetic scan (0: a circularised disk 2096 px wide, which CLAHE's 2 x 2 grid divides; 1: 2097 px, which it
does not -- the tiles then count OpenCV's reflected border)."""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shifts = [int(s) for s in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
n, w, h, bits = (int(v) for v in sys.argv[3:7]) if len(sys.argv) > 6 else (2000, 2000, 200, 16)
stack = synth.synth_frames_torch(n, w, h, bits, seed=int(os.environ.get('SHG_STEP_SEED', '0')), padded=True)
torch.cuda.synchronize()
for _ in range(steps):
    opts = SHG_MAIN.default_options()
    opts.update(_nolog=True, shift=list(shifts))
    with contextlib.redirect_stdout(io.StringIO()):
        res = Solex_recon.solex_do_work([(array_reader(stack), opts)], True, return_results=True)
torch.cuda.synchronize()
# what tools/roofline_table.py needs: N W H bits S k out_w
print(n, w, h, bits, len(dict.fromkeys([10, 0] + list(shifts))), len(shifts), res[0][0][0].shape[1])
