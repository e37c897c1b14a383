"""Multi-GPU sharding of one scan: one process per GPU, frames split into contiguous
blocks, RCCL (torch.distributed backend "nccl") over xGMI for the TWO exchange steps of a scan:

  * after pass A  -- ONE all-gather of every rank's packed frame statistics: the partial sums as 32-bit words, the partial
    maxima as 16-bit words, and one word that says whether this rank has failed anywhere in the series so far
    (ih*iw*6 + 4 bytes per rank: 2.4 MB at 2000x200); every rank then adds / maximises the G pieces itself.  Integer
    reductions are order independent, so every rank count gives bit-identical mean / max images.  (Rounds 1-5 used an
    all-reduce SUM of the int64 sums, an all-reduce MAX of the widened maxima and, in a series, a one-word all-reduce before
    every scan to ask whether anyone had failed: three latency-bound collectives where one does.)
  * after pass B  -- all-reduce SUM of the zero-initialised disk mosaic [S, ih, n_total] into
    which every rank has extracted the columns of its own frames (16 MB at S=2, C3); in a series of
    sharded scans a reduce to the scan's owner (scan k belongs to rank k mod G, scan_owner), who alone
    post-processes it and writes its files.

In a series two scans are being read at a time (Solex_recon._sharded_series: scan k + 1 is decoded, summed and exchanged
while scan k is fitted and extracted); the collectives of the two reading threads are issued in ONE order on every rank
(Sequencer: exchange 1 of scan k + 1 before exchange 2 of scan k).  `counters` counts what was issued: the tests hold a
series to two collectives per scan.

Both messages are small: the collectives are latency-bound, the per-link xGMI bandwidth
does not bind.  Everything after the gather (fit of the limb, warp, transversalium, CLAHE)
works on the mosaic.  Folder mode (one file per GPU) uses no collective at all.
The helpers also run on CPU tensors with the gloo backend (tests/test_dist_cpu.py).
"""
import threading

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


counters = {'collectives': 0}          # collectives issued by this process (the tests hold a series to two per scan)
_tls = threading.local()               # .scan: index of the scan this thread is reading, inside a series
_series = None                         # the running series of sharded scans (Solex_recon._sharded_series), or None


class SeriesAborted(RuntimeError):
    """Some rank reported a failure in the exchange after pass A: every rank leaves the series at the same scan."""


class Sequencer:
    """The collectives of a series of sharded scans, issued in ONE order on every rank although two threads read scans:
    exchange 1 (frame statistics) of scan k has key 2 k, exchange 2 (mosaic) of scan k has key 2 k + 3 -- so scan k + 1's
    statistics travel before scan k's mosaic, which is still being extracted.  A thread runs its collective when every smaller
    key has run or been skipped (its scan failed on every rank alike before it got there)."""

    def __init__(self, n_scans, interleaved=True):
        self.second = 3 if interleaved else 1           # one reading thread: exchange 2 of scan k right behind its exchange 1
        self.keys = sorted([2 * k for k in range(n_scans)] + [2 * k + self.second for k in range(n_scans)])
        self.done = set()
        self.pos = 0
        self.cancelled_after = None
        self.cv = threading.Condition()

    def _advance(self):
        while self.pos < len(self.keys) and self.keys[self.pos] in self.done:
            self.pos += 1
        self.cv.notify_all()

    def run(self, key, fn):
        with self.cv:
            while True:
                if self.cancelled_after is not None and key > self.cancelled_after:
                    raise SeriesAborted('series of sharded scans stopped before collective %d' % key)
                if self.pos < len(self.keys) and self.keys[self.pos] == key:
                    break
                self.cv.wait()
        try:
            return fn()
        finally:
            with self.cv:
                self.done.add(key)
                self._advance()

    def is_done(self, key):
        with self.cv:
            return key in self.done

    def skip(self, *keys):
        with self.cv:
            self.done.update(keys)
            self._advance()

    def cancel_after(self, key):
        with self.cv:
            if self.cancelled_after is None or key < self.cancelled_after:
                self.cancelled_after = key
            self.cv.notify_all()


class Series:
    """What the exchanges need to know about the running series: the order of its collectives and whether this rank has failed."""

    def __init__(self, n_scans, failed, interleaved=True):
        self.seq = Sequencer(n_scans, interleaved)
        self.failed = failed                 # () -> bool: has this rank recorded an error in the series so far?
        self.aborted = False
        self.flagged = False                 # this rank has set the failure word in an exchange


def begin_series(n_scans, failed, interleaved=True):
    global _series
    _series = Series(n_scans, failed, interleaved)
    return _series


def end_series():
    global _series
    _series = None


def reading(scan_index):
    """This thread reads scan `scan_index` of the running series (None: no longer)."""
    _tls.scan = scan_index


def run_series(n_scans, read_scan, two_readers=True, second_thread=None, before_verdict=None, device=None, also_failed=None,
               placeholder_second=None):
    """Drive a series of frame-sharded scans on this rank: read_scan(i) does scan i -- decode, pass A, exchange_frame_stats, fit, pass
    B, gather_columns, hand-over of the mosaic -- on the calling thread and raises what goes wrong.  two_readers: this thread takes
    the even scans, a second one (run inside the context manager second_thread(), which gives it its device and stream) the odd ones;
    their collectives go out in one order on every rank (Sequencer).  A failure of this rank travels in the next exchange after
    pass A and stops every rank at that scan (SeriesAborted); a scan that fails on every rank alike gives up its places in the
    order.  placeholder_second(i): what a rank does when ITS scan i fails between the two exchanges (pass B ran out of memory, say):
    it still takes part in exchange 2, with an empty mosaic -- the other ranks are on their way into it -- and reports in the next
    exchange after pass A; without the callback the place is given up, which is only right for a failure every rank runs into alike.
    before_verdict(): waited for after the last scan (the owner's post-processing threads), also_failed(): their failures.
    At the end one word goes round
    (any_failed): -> the list of (scan index, exception) of this rank, with a RuntimeError appended when only another rank failed."""
    import contextlib
    errors = []
    lock = threading.Lock()

    def failed():                                           # (also_failed: failures outside the reading threads -- the owner's post-processing)
        with lock:
            return bool(errors) or (also_failed is not None and bool(also_failed()))
    series = begin_series(n_scans, failed, interleaved=two_readers)
    second = series.seq.second

    def note(i, e):
        with lock:
            errors.append((i, e))

    def reader(first, step):
        for i in range(first, n_scans, step):
            go_on = not series.aborted
            if go_on:
                reading(i)
                try:
                    read_scan(i)
                except SeriesAborted:
                    go_on = False
                    if series.flagged and not failed():      # (the caller usually has the real exception on record: also_failed)
                        note(i, RuntimeError('this rank set the failure word in the exchange of scan %d' % i))
                except BaseException as e:      # noqa: BLE001 -- reported to the caller; the series goes on to the exchange that tells the others
                    note(i, e)
                    if placeholder_second is not None and series.seq.is_done(2 * i) and not series.seq.is_done(2 * i + second):
                        try:
                            placeholder_second(i)
                        except SeriesAborted:
                            go_on = False
                        except BaseException as e2:      # noqa: BLE001
                            note(i, e2)
                finally:
                    reading(None)
                # (the places of collectives this scan did not get to -- it failed on every rank alike, or the series stopped -- are
                # given up; those it did run are in the set already)
                series.seq.skip(2 * i, 2 * i + second)
            if not go_on:
                series.seq.skip(*[k for j in range(i, n_scans, step) for k in (2 * j, 2 * j + second)])
                break

    try:
        if two_readers and n_scans > 1:
            def other():
                try:
                    with (second_thread() if second_thread is not None else contextlib.nullcontext()):
                        reader(1, 2)
                except BaseException as e:      # noqa: BLE001
                    note(n_scans, e)
                    series.seq.skip(*[k for j in range(1, n_scans, 2) for k in (2 * j, 2 * j + second)])
            t = threading.Thread(target=other, name='shg-read1', daemon=True)
            t.start()
            try:
                reader(0, 2)
            finally:
                t.join()
        else:
            reader(0, 1)
        if before_verdict is not None:
            before_verdict()
    finally:
        end_series()
    # the owners' post-processing of the last scans has no later exchange to report into: one word at the end of the series
    if any_failed(failed(), device) and not errors:
        errors.append((n_scans, RuntimeError('another rank failed in this series of sharded scans')))
    return errors


def _ordered(phase, fn):
    k = getattr(_tls, 'scan', None)
    if _series is None or k is None:
        return fn()
    return _series.seq.run(2 * k + (_series.seq.second if phase else 0), fn)


def _count(n=1):
    counters['collectives'] += n


def exchange_frame_stats(total, mx, failed=False, n_frames=None):
    """The exchange after pass A.  total: int64 [P] partial sums; mx: uint16 [P] partial maxima (raw sample units) of this rank's
    frames -> (int64 [P] sums, uint16 [P] maxima) over all ranks' frames.  ONE all-gather of [P sums as 32-bit words | P maxima as
    16-bit words | failure word] per rank, reduced here by every rank for itself (32-bit partial sums: a rank's share must stay
    below 65 537 frames of 16-bit samples -- a longer share takes the two all-reduces of the earlier rounds).  Inside a series
    (begin_series) the failure word carries `failed` or the series' own record, and a word set by ANY rank raises SeriesAborted on
    EVERY rank: all of them leave the series at this scan instead of one leaving the others waiting in the next collective."""
    series = _series if getattr(_tls, 'scan', None) is not None else None
    flag = bool(failed) or (series is not None and series.failed())
    p = int(total.numel())
    # (n_frames: this rank's share -- bounds its sums without a look at them, which would cost a device synchronisation)
    fits = int(n_frames) * 65535 < (1 << 32) if n_frames is not None else bool((total < (1 << 32)).all())
    if not fits:
        return _allreduce_sum_max(total, mx)
    g = world_size()
    dev = total.device
    words = p + (p + 1) // 2 + 1
    mine = torch.zeros(words, dtype=torch.int32, device=dev)
    mine[:p] = total.to(torch.int32)                                  # (wraps: the bit pattern of the unsigned 32-bit sum)
    mine[p:p + (p + 1) // 2].view(torch.int16)[:p] = mx.view(torch.int16)
    mine[words - 1] = 1 if flag else 0
    if flag and series is not None:
        series.flagged = True
    staged = _staged(mine)
    send = mine.cpu() if staged else mine
    every = torch.empty(g * words, dtype=torch.int32, device=send.device)

    def gather():
        _count()
        td.all_gather_into_tensor(every, send)
        # (the verdict before this collective's turn is given up: the other reading thread must not start the next one on some ranks
        # and be cancelled on others)
        stop = bool((every.view(g, words)[:, words - 1] != 0).any())
        if stop and series is not None:
            series.aborted = True
            series.seq.cancel_after(2 * _tls.scan)
        return stop
    if _ordered(0, gather):
        raise SeriesAborted('a rank reported a failure: every rank stops at this scan')
    every = every.view(g, words)
    if staged:
        every = every.to(dev)
    if every.is_cuda:
        # the G pieces folded in one launch (shg_reduce_frame_stats)
        from . import ops
        return ops.reduce_frame_stats(every, p)
    sums = (every[:, :p].to(torch.int64) & 0xffffffff).sum(dim=0)
    maxima = (every[:, p:p + (p + 1) // 2].contiguous().view(torch.int16)[:, :p].to(torch.int32) & 0xffff).amax(dim=0)
    return sums, maxima.to(torch.int16).view(torch.uint16)


def _allreduce_sum_max(total, mx):
    """Two all-reduces (sums too large for 32-bit pieces): SUM of the int64 sums, MAX of the widened maxima."""
    wide = mx.view(torch.int16).to(torch.int32) & 0xffff          # RCCL has no uint16 MAX; widen losslessly
    staged = _staged(total)
    dev = total.device
    if staged:
        total, wide = total.cpu(), wide.cpu()

    def both():
        _count(2)
        td.all_reduce(total, op=td.ReduceOp.SUM)
        td.all_reduce(wide, op=td.ReduceOp.MAX)
    _ordered(0, both)
    if staged:
        total, wide = total.to(dev), wide.to(dev)
    return total, wide.to(torch.int16).view(torch.uint16)


def allreduce_sum_max(total, mx):
    """(the name of rounds 1-5) -> exchange_frame_stats."""
    return exchange_frame_stats(total, mx)


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
        _count()
        if dst is None:
            td.all_reduce(t, op=td.ReduceOp.SUM)
        else:
            td.reduce(t, dst=int(dst), op=td.ReduceOp.SUM)         # half an all-reduce's traffic: only the owner needs the mosaic
    if _staged(words):
        host = words.cpu()
        _ordered(1, lambda: exchange(host))
        if dst is None or rank() == int(dst):
            words.copy_(host)
    else:
        _ordered(1, lambda: exchange(words))
    return mosaic[:, :, :n_total]


def broadcast_object(obj, src=0):
    """A small picklable object (the limb geometry: a dozen floats) from rank src to every rank."""
    box = [obj]
    _count()
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
    """Whether ANY rank reports a failure (one tiny all-reduce).  A series of sharded scans asks ONCE, at its end (the owners'
    post-processing of the last scans has no later exchange to report into); during the series the word travels with the
    exchange after pass A (exchange_frame_stats)."""
    if not active():
        return bool(failed)
    on_gpu = td.get_backend() != 'gloo' and device is not None
    t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=device if on_gpu else 'cpu')
    _count()
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return bool(int(t.item()))
