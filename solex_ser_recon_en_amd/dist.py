"""Multi-GPU sharding of one scan: one process per GPU, frames split into contiguous
blocks, RCCL (torch.distributed backend "nccl") over xGMI for the two exchange steps:

  * after pass A  -- all-reduce SUM of the integer sum frame and all-reduce MAX of the max
    frame (ih*iw*8 + ih*iw*4 bytes: 1.6 MB + 0.8 MB at 2000x200).  Integer reductions are
    order independent, so every rank count gives bit-identical mean/max images;
  * after pass B  -- all-reduce SUM of the zero-initialised disk mosaic [S, ih, n_total] into
    which every rank has extracted the columns of its own frames (16 MB at S=2, C3); in a series of
    sharded scans a reduce to the scan's owner (scan k belongs to rank k mod G, scan_owner), who alone
    post-processes it and writes its files.

Both messages are small: the collectives are latency-bound, the per-link xGMI bandwidth
does not bind.  Everything after the gather (fit of the limb, warp, transversalium, CLAHE)
works on the mosaic.  Folder mode (one file per GPU) uses no collective at all.
The helpers also run on CPU tensors with the gloo backend (tests/test_dist_cpu.py).
"""
import torch
import torch.distributed as td


def active():
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def rank():
    return td.get_rank() if active() else 0


def world_size():
    return td.get_world_size() if active() else 1


def frame_block(n_frames, r=None, w=None):
    """Contiguous block [k0, k1) of rank r: sizes differ by at most one frame."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, extra = divmod(int(n_frames), w)
    k0 = r * base + min(r, extra)
    return k0, k0 + base + (1 if r < extra else 0)


def is_sharded(rdr):
    rng = getattr(rdr, 'frame_range', None)
    return active() and rng is not None and tuple(rng) != (0, int(rdr.FrameCount))


def _staged(t):
    """gloo moves host memory only: stage GPU tensors through the host (functional tests with several
    ranks on one GPU).  RCCL (backend "nccl") works on the device buffers directly."""
    return t.is_cuda and td.get_backend() == 'gloo'


def allreduce_sum_max(total, mx):
    """total: int64 [P] partial sums; mx: uint16 [P] partial maxima (raw sample units)."""
    wide = mx.view(torch.int16).to(torch.int32) & 0xffff          # RCCL has no uint16 MAX; widen losslessly
    if _staged(total):
        dev = total.device
        total, wide = total.cpu(), wide.cpu()
        td.all_reduce(total, op=td.ReduceOp.SUM)
        td.all_reduce(wide, op=td.ReduceOp.MAX)
        total, wide = total.to(dev), wide.to(dev)
    else:
        td.all_reduce(total, op=td.ReduceOp.SUM)
        td.all_reduce(wide, op=td.ReduceOp.MAX)
    return total, wide.to(torch.int16).view(torch.uint16)


def mosaic_columns(frame_range, n_total, flip_x=False):
    """Column span [c0, c1) of the mosaic that this rank's frames [k0, k1) fill (reversed under flip_x)."""
    k0, k1 = int(frame_range[0]), int(frame_range[1])
    return (n_total - k1, n_total - k0) if flip_x else (k0, k1)


def scan_owner(index):
    """The rank that post-processes scan `index` of a series of frame-sharded scans and writes its files: index mod world, so that
    the tails of consecutive scans (limb fit, warp, transversalium, CLAHE, encoders) run on different GPUs at the same time."""
    return int(index) % world_size()


def gather_columns(fill, n_shifts, ih, frame_range, n_total, flip_x, device, dst=None):
    """The full disks [S, ih, n_total] on every rank (dst None) or on rank `dst` only (the other ranks get their own partial
    mosaic back and must not use it) from per-rank column blocks, with ONE collective and no
    re-layout pass: every rank extracts the columns of its own frames straight into a zeroed mosaic
    (`fill(mosaic, k0)` -- shg_extract_columns takes the column offset and the flip), then the mosaic is
    all-reduced with SUM.  Column blocks are disjoint, so every 16-bit cell receives one non-zero contribution:
    summing the buffer as 32-bit words carries nothing from one cell into the next and the result is the exact
    mosaic (RCCL has no 16-bit integer type).  BASELINE.json: "all-reduce ... for the final column mosaic"."""
    w = world_size()
    blocks = [frame_block(n_total, r, w) for r in range(w)]
    if blocks[rank()] != (int(frame_range[0]), int(frame_range[1])):
        raise RuntimeError('gather_columns: this rank holds frames %s, expected %s' % (tuple(frame_range), blocks[rank()]))
    pitch = (n_total + 63) // 64 * 64                        # even: whole 32-bit words
    mosaic = torch.zeros((n_shifts, ih, pitch), dtype=torch.uint16, device=device)
    fill(mosaic[:, :, :n_total], int(frame_range[0]))
    words = mosaic.view(torch.int32)

    def exchange(t):
        if dst is None:
            td.all_reduce(t, op=td.ReduceOp.SUM)
        else:
            td.reduce(t, dst=int(dst), op=td.ReduceOp.SUM)         # half an all-reduce's traffic: only the owner needs the mosaic
    if _staged(words):
        host = words.cpu()
        exchange(host)
        if dst is None or rank() == int(dst):
            words.copy_(host)
    else:
        exchange(words)
    return mosaic[:, :, :n_total]


def broadcast_object(obj, src=0):
    """A small picklable object (the limb geometry: a dozen floats) from rank src to every rank."""
    box = [obj]
    td.broadcast_object_list(box, src=src)
    return box[0]


def my_share(index):
    """Round-robin dealing of the requested disks of a frame-sharded Doppler stack: disk `index` (in request order) is
    post-processed by rank index mod world."""
    return index % world_size() == rank()


def agree(fn, src=0):
    """Run fn() on rank `src` and give every rank its outcome: the (picklable) result everywhere, or an exception everywhere --
    rank `src` re-raises its own, the others a RuntimeError carrying its text.  What keeps a failure on one rank (the limb
    fit of a sharded scan runs on rank 0 only) from leaving the others waiting in the next collective."""
    failure, message = None, None
    if rank() == src:
        try:
            message = (True, fn())
        except Exception as e:      # noqa: BLE001 -- handed to every rank below
            failure = e
            message = (False, repr(e))
    ok, payload = broadcast_object(message, src=src)
    if not ok:
        raise failure if failure is not None else RuntimeError('rank %d failed: %s' % (src, payload))
    return payload


def refuse_unshardable(n_frames, what=''):
    """A scan with fewer frames than ranks cannot be sharded: every rank sees the same header, so every rank raises (none is
    left waiting in the first all-reduce)."""
    if int(n_frames) < world_size():
        raise Exception('error input file %s: %d frames cannot be sharded over %d ranks' % (what, int(n_frames), world_size()))


def any_failed(failed, device=None):
    """Whether ANY rank reports a failure (one tiny all-reduce): ranks that shard a series of scans ask before every scan, so that
    a rank that failed outside a collective (rank 0 post-processing the previous scan) stops all of them together instead of
    leaving the others waiting in the next scan's all-reduce."""
    if not active():
        return bool(failed)
    on_gpu = td.get_backend() != 'gloo' and device is not None
    t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=device if on_gpu else 'cpu')
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return bool(int(t.item()))
