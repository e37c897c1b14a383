"""cProfile of the Python side of a scan (one worker): which functions hold the interpreter lock.  py_profile.py [scans]"""
import contextlib
import cProfile
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=0, padded=True)
torch.cuda.synchronize()


def run(n):
    tasks = []
    for _ in range(n):
        o = SHG_MAIN.default_options()
        o.update(_nolog=True)
        tasks.append((array_reader(stack), o))
    with contextlib.redirect_stdout(io.StringIO()):
        Solex_recon.solex_do_work(tasks, True, workers=1)
    torch.cuda.synchronize()


run(10)
pr = cProfile.Profile()
pr.enable()
run(steps)
pr.disable()
out = io.StringIO()
st = pstats.Stats(pr, stream=out)
st.sort_stats('tottime').print_stats(45)
text = out.getvalue()
print(text.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/', ''))
print('(times are totals over %d scans: divide by %d for per-scan; ctypes stage calls include the GPU work)' % (steps, steps))
