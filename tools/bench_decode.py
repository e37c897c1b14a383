"""File -> pinned host -> HBM upload rate of video_reader.device_stack() (PCIe-inclusive decode)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import video_reader  # noqa: E402

n, w, h = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 2000, 200
path = '/dev/shm/bench_decode.ser' if os.path.isdir('/dev/shm') else '/tmp/bench_decode.ser'
stack = synth.synth_frames_torch(n, w, h, 16, seed=0)
synth.write_ser(path, stack.cpu().numpy())
size = os.path.getsize(path)
try:
    for rep in range(3):
        rdr = video_reader(path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev = rdr.device_stack()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('device_stack: %.1f MB in %.1f ms -> %.1f GB/s, %.0f frames/s (file in %s)' % (
            size / 1e6, dt * 1e3, size / dt / 1e9, n / dt, os.path.dirname(path)))
    assert torch.equal(dev.view(torch.int16), stack.view(torch.int16))
finally:
    os.remove(path)

# ---- PCIe-inclusive throughput of a folder of files on one GPU (decode of file k+1 overlaps file k) ----
import contextlib
import io

from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon  # noqa: E402

files = []
for i in range(5):
    f = path.replace('.ser', '_%d.ser' % i)
    synth.write_ser(f, stack.cpu().numpy())
    files.append(f)
try:
    for rep in range(2):
        tasks = []
        for f in files:
            o = SHG_MAIN.default_options()
            o['_nolog'] = True
            tasks.append((f, o))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            Solex_recon.solex_do_work(tasks, True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('solex_do_work on %d files from %s: %.1f ms per file -> %.0f frames/s PCIe-inclusive' % (
            len(files), os.path.dirname(path), dt / len(files) * 1e3, n * len(files) / dt))
finally:
    for f in files:
        os.remove(f)
