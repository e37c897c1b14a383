"""device_stack() upload rate against the number of reader threads and the chunk size (file in /dev/shm).
sweep_decode.py [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import device, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import video_reader  # noqa: E402

n, w, h = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 2000, 200
path = '/dev/shm/sweep_decode.ser'
stack = synth.synth_frames_torch(n, w, h, 16, seed=0)
synth.write_ser(path, stack.cpu().numpy())
size = os.path.getsize(path)
device.bind_thread('io')                                  # where solex_do_work's decoder thread runs
try:
    for readers in (4, 8, 12, 16):
        for chunk_mb in (8, 16, 32, 64):
            best = 0.0
            for rep in range(4):
                rdr = video_reader(path)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rdr.device_stack(chunk_bytes=chunk_mb << 20, readers=readers)
                torch.cuda.synchronize()
                best = max(best, size / (time.perf_counter() - t0) / 1e9)
                rdr._stack = None
            print('readers %2d  chunk %2d MB  %.1f GB/s' % (readers, chunk_mb, best), flush=True)
finally:
    os.remove(path)
