"""cProfile of the host side of one bench step (tuning aid)."""
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

stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=0)


def step():
    opts = SHG_MAIN.default_options()
    opts['_nolog'] = True
    rdr = array_reader(stack)
    with contextlib.redirect_stdout(io.StringIO()):
        disk_list, bounds, hdr = Solex_recon.solex_read(rdr, opts)
        return Solex_recon.solex_process(opts, disk_list, bounds, hdr)


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(40)
