"""The host-to-device copy rate the decode-inclusive figure (bench.py's e2e) is up against: one pinned 1.6 GB buffer copied to the device
asynchronously, in one piece and in 64 MB pieces (the size the upload readers hand over), HIP events around 5 copies each."""
import torch

dev = torch.device('cuda', 0)
n = 1_600_000_000
host = torch.empty(n, dtype=torch.uint8).pin_memory()
host.fill_(7)
dst = torch.empty(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ms = timed(lambda: dst.copy_(host, non_blocking=True))
print('one 1.6 GB copy: %.2f ms = %.1f GB/s' % (ms, n / ms / 1e6))
piece = 64 << 20


def pieces():
    for o in range(0, n, piece):
        dst[o:o + piece].copy_(host[o:o + piece], non_blocking=True)


ms = timed(pieces)
print('64 MB pieces:    %.2f ms = %.1f GB/s' % (ms, n / ms / 1e6))
s2 = torch.cuda.Stream(device=dev)


def two_streams():
    half = n // 2
    cur = torch.cuda.current_stream(dev)
    s2.wait_stream(cur)
    dst[:half].copy_(host[:half], non_blocking=True)
    with torch.cuda.stream(s2):
        dst[half:].copy_(host[half:], non_blocking=True)
    cur.wait_stream(s2)


ms = timed(two_streams)
print('two streams:     %.2f ms = %.1f GB/s' % (ms, n / ms / 1e6))
