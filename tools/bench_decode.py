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
