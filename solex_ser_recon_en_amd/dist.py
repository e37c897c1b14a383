"""Multi-GPU sharding of one scan: one process per GPU, frames split into contiguous
blocks, RCCL (torch.distributed backend "nccl") over xGMI for the two exchange steps:

  * after pass A  -- all-reduce SUM of the integer sum frame and all-reduce MAX of the max
    frame (ih*iw*8 + ih*iw*4 bytes: 1.6 MB + 0.8 MB at 2000x200).  Integer reductions are
    order independent, so every rank count gives bit-identical mean/max images;
  * after pass B  -- all-gather of the per-rank column blocks [S, ih, n_local] into the
    full disks (4 MB per rank at S=2).

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


def gather_columns(local, frame_range, n_total, flip_x=False):
    """local: uint16 [S, ih, n_local] (this rank's frames, in frame order).
    Returns uint16 [S, ih, n_total] on every rank, columns in frame order (reversed if flip_x)."""
    w = world_size()
    s, ih, n_local = local.shape
    blocks = [frame_block(n_total, r, w) for r in range(w)]
    if (blocks[rank()][0], blocks[rank()][1]) != tuple(frame_range) or n_local != frame_range[1] - frame_range[0]:
        raise RuntimeError('gather_columns: this rank holds frames %s, expected %s' % (tuple(frame_range), blocks[rank()]))
    n_max = max(b - a for a, b in blocks)
    send = torch.zeros((s, ih, n_max), dtype=torch.int16, device=local.device)
    send[:, :, :n_local] = local.view(torch.int16)
    # neither RCCL nor gloo moves 16-bit integers: gather the bytes
    send8 = send.view(torch.uint8)
    if _staged(send8):
        host = send8.cpu()
        recv_h = [torch.empty_like(host) for _ in range(w)]
        td.all_gather(recv_h, host)
        recv8 = [r.to(local.device) for r in recv_h]
    else:
        recv8 = [torch.empty_like(send8) for _ in range(w)]
        td.all_gather(recv8, send8)
    recv = [r.view(torch.int16) for r in recv8]
    full = torch.cat([recv[r][:, :, :blocks[r][1] - blocks[r][0]] for r in range(w)], dim=2)
    if flip_x:
        full = torch.flip(full, dims=(2,))
    pitch = (n_total + 63) // 64 * 64
    out = torch.zeros((s, ih, pitch), dtype=torch.int16, device=local.device)
    out[:, :, :n_total] = full
    return out.view(torch.uint16)[:, :, :n_total]


def broadcast_object(obj, src=0):
    """A small picklable object (the limb geometry: a dozen floats) from rank src to every rank."""
    box = [obj]
    td.broadcast_object_list(box, src=src)
    return box[0]
