"""A file the page cache does not hold (written to a disk-backed directory, fsync'ed, dropped with POSIX_FADV_DONTNEED) -> HBM:
buffered reads (SHG_READ_DIRECT=0: storage -> page cache -> pinned buffer -> GPU) against O_DIRECT (auto: storage -> pinned buffer
-> GPU).  What bounds either is the storage device; what differs is the host memory traffic and the page cache left behind.
    python3 tools/bench_cold_read.py [directory] [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import video_reader  # noqa: E402

where = sys.argv[1] if len(sys.argv) > 1 else '/tmp'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
path = os.path.join(where, 'shg_cold_%d.ser' % os.getpid())
stack = synth.synth_frames_torch(n, 2000, 200, 16, seed=0)
synth.write_ser(path, stack.cpu().numpy())
size = os.path.getsize(path)


def drop():
    fd = os.open(path, os.O_RDONLY)
    os.fsync(fd)
    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
    os.close(fd)


try:
    for mode in ('0', 'auto', '0', 'auto'):
        os.environ['SHG_READ_DIRECT'] = mode
        drop()
        rdr = video_reader(path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev = rdr.device_stack()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = torch.equal(dev.view(torch.int16), stack.view(torch.int16))
        print('SHG_READ_DIRECT=%-4s %.1f MB cold from %s in %.1f ms -> %.2f GB/s  (frames identical: %s)' % (mode, size / 1e6, where, dt * 1e3, size / dt / 1e9, ok))
    os.environ['SHG_READ_DIRECT'] = '0'
    video_reader(path).device_stack()                          # one buffered read: the page cache holds the file now
    os.environ['SHG_READ_DIRECT'] = 'auto'
    t0 = time.perf_counter()
    video_reader(path).device_stack()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('SHG_READ_DIRECT=auto on the now cached file: %.2f GB/s (mincore says cached: the buffered path)' % (size / dt / 1e9))
finally:
    os.remove(path)
