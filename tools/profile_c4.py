"""cProfile of the host side of one C4 step (-w -10:10:1, 21 requested disks) -- tuning aid."""
import cProfile
import contextlib
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=0, padded=True)


def step():
    opts = SHG_MAIN.default_options()
    opts.update(_nolog=True, shift=list(range(-10, 11)))
    with contextlib.redirect_stdout(io.StringIO()):
        return Solex_recon.solex_do_work([(array_reader(stack), opts)], True, return_results=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(40)
st.sort_stats('tottime').print_stats(25)
