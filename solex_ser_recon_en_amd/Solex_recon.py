"""Orchestration of the SHG reconstruction on MI355X.

Keeps the reference's call surface (Solex_recon.py:26-174): solex_do_work(tasks,
flag_command_line), solex_read(file, options), solex_process(options, disk_list,
backup_bounds, hdr), single_image_process(...), including the `options` keys they
mutate (basefich0, shift_requested, shift, ratio_fixe, slant_fix) and the output files.

What changed underneath:
  * the SER file is decoded ONCE into HBM (the reference reads it twice, :61-63) and
    both frame passes are HIP kernels on the resident stack;
  * disks stay in HBM from extraction to the final contrast products (DeviceImage);
    the reference pickles them into a multiprocessing.Pool worker (:38);
  * a batch of files is processed by up to four scan workers at once (threads with their own HIP
    stream, the reference's Pool(4), :30-42), fed by one decoder thread; every stage function is one C
    call (stages.py), so the workers do not queue for the interpreter;
  * file encoders and the matplotlib diagnostics run on background threads (outputs.py);
  * with torch.distributed initialised (one process per GPU) the frames of ONE scan are
    sharded across ranks: all-reduce after the mean/max pass, all-reduce of the zero-filled disk
    mosaic after extraction, post-processing on the mosaic (rank 0 writes the files).
    Several files (folder mode) are dealt round-robin to ranks instead: no collective.
"""
import contextlib
import functools
import math
import os
import queue
import sys
import threading

import numpy as np

from . import dist, ops, outputs, timing
from .device import DeviceImage, bind_thread, cpu_plan, default_device, to_device_u16
from .ellipse_to_circle import correct_image, ellipse_to_circle
from .fits_io import write_fits
from . import stages
from .solex_util import (as_uint16_image, clearlog, compute_mean_return_fit, correct_transversalium2_batch, extract_disks,
                         image_process_batch, logme, make_header, output_path, plots_enabled, removeVignette, savgol_taps,
                         savgol_window, write_complete, write_products)
from .video_reader import video_reader


WORKERS = 4          # the reference post-processes up to four files at once (Pool(4), Solex_recon.py:30)


def _worker_count(workers, n_tasks):
    if workers is None:
        try:
            workers = int(os.environ.get('SHG_WORKERS', WORKERS))
        except ValueError:
            workers = WORKERS
    return max(1, min(int(workers), n_tasks))


class _Service:
    """A long-lived daemon thread that runs the jobs posted to it, one after the other.  The decoder and the scan workers
    of a batch run on such threads instead of fresh ones: starting five threads costs a batch of twenty scans a tenth
    of its time (a new thread's first steps wait for the interpreter lock, 0.25 ms each; the HIP runtime adds
    milliseconds to one of the first thread starts after another generation of threads has exited)."""

    _all = {}
    _guard = threading.Lock()

    def __init__(self, name):
        self.jobs = queue.SimpleQueue()
        self.thread = threading.Thread(target=self._loop, name=name, daemon=True)
        self.thread.start()

    def _loop(self):
        while True:
            job = self.jobs.get()
            try:
                job()
            except BaseException:       # noqa: BLE001 -- a job reports its own failure to whoever waits for it
                pass

    @classmethod
    def named(cls, name):
        with cls._guard:
            svc = cls._all.get(name)
            if svc is None or not svc.thread.is_alive():
                svc = cls._all[name] = cls(name)
            return svc


class _Decoder:
    """Decodes the files of a batch into HBM, in order, ahead of the scans that consume them (reader threads + copy
    streams, video_reader.device_stack): the overlap the reference gets from reading in the parent while its Pool workers
    post-process (Solex_recon.py:30-42).  At most `ahead` decoded stacks wait for a scan worker.  The thread
    binds itself to the caller's device: torch's current device is thread-local and a fresh thread starts on
    device 0, which is the wrong GPU for every rank but the first."""

    def __init__(self, tasks, frame_range, ahead):
        self.tasks, self.frame_range = tasks, frame_range
        self.device = default_device()                       # on the caller's thread
        self.slots = threading.Semaphore(max(1, ahead))
        self.ready = [threading.Event() for _ in tasks]
        self.out = [None] * len(tasks)
        self.stop = False
        self.finished = threading.Event()
        _Service.named('shg-decode-%s' % self.device).jobs.put(self._run)

    def _run(self):
        try:
            import torch
            torch.cuda.set_device(self.device)
            bind_thread('io', self.device)                   # the reader threads inherit it: off the scan workers' cores
            queued = []                                      # files whose chunks are with the upload service, oldest first

            def land(j):
                rdr = self.out[j][0]
                try:
                    if rdr is not None and not hasattr(self.tasks[j][0], 'device_stack'):
                        rdr.device_stack(device=self.device)            # waits for the last chunk; raises what a chunk raised
                except BaseException as e:      # noqa: BLE001 -- re-raised when this file's turn comes
                    self.out[j] = (None, e)
                self.ready[j].set()

            for i, (file, _) in enumerate(self.tasks):
                self.slots.acquire()
                if self.stop:
                    self.out[i] = (None, RuntimeError('batch cancelled'))
                elif hasattr(file, 'device_stack'):
                    self.out[i] = (file, None)
                else:
                    try:
                        rdr = video_reader(file)
                        if self.frame_range is not None:
                            _check_shardable(rdr)
                            rdr.frame_range = self.frame_range(int(rdr.FrameCount))
                        rdr.begin_device_stack(device=self.device)      # queue its chunks behind the previous file's: the link stays busy
                        self.out[i] = (rdr, None)
                    except BaseException as e:      # noqa: BLE001 -- re-raised when this file's turn comes
                        self.out[i] = (None, e)
                queued.append(i)
                if len(queued) > 1:
                    land(queued.pop(0))
            while queued:
                land(queued.pop(0))
        except BaseException as e:      # noqa: BLE001 -- whatever stops this thread must not leave a scan waiting for its file
            for j in range(len(self.tasks)):
                if not self.ready[j].is_set():
                    self.out[j] = (None, e)
                    self.ready[j].set()
        finally:
            self.finished.set()

    def get(self, i):
        """The reader of task i, its frames resident in HBM (raises what its decode raised)."""
        self.ready[i].wait()
        rdr, err = self.out[i]
        self.out[i] = None
        self.slots.release()                                 # the decoder may start one more file
        if err is not None:
            raise err
        return rdr

    def cancel(self):
        self.stop = True
        for _ in self.tasks:
            self.slots.release()
        self.finished.wait()
        self.out = [None] * len(self.tasks)                  # stacks decoded for scans that will not run: let go of them (GBs of HBM)


def solex_do_work(tasks, flag_command_line=False, distribute='auto', return_results=False, workers=None):
    """tasks: list of (file, options).  Raises on failure (the front door catches, SHG_MAIN.py:136-143).

    distribute (only matters under torch.distributed with more than one rank):
      'auto'   one file -> its frames are sharded over the ranks; several files -> file i goes to rank i mod G
      'frames' shard the frames of every file over the ranks (collectives per file)
      'none'   this rank processes exactly the tasks it was given (the caller already dealt the files)
    return_results: also return the solex_process results, ONE ENTRY PER TASK in task order: a list of (cc, protus) per file this
      rank post-processed, None for a file whose products another rank holds (a series of frame-sharded single-shift scans: scan k's
      products exist on rank k mod G only -- dist.scan_owner; a sharded Doppler stack's disks are dealt to all ranks: every rank gets
      the list of its own).
    workers: scans in flight at once in this process (default SHG_WORKERS or 4, the reference's Pool(4)): each scan
      worker is a thread with its own HIP stream, so the host control plane of one file (polynomial fits, limb
      geometry, Savitzky-Golay trend) overlaps the kernels of the others.  Files are independent, so every product
      is bit-identical to the serial order.  A series of frame-sharded scans is read by two threads and post-processed by a third
      (_sharded_series); `workers` does not apply to it."""
    tasks = list(tasks)
    if distribute not in ('auto', 'frames', 'none'):
        raise ValueError("distribute must be 'auto', 'frames' or 'none'")
    shard_frames = dist.active() and (distribute == 'frames' or (distribute == 'auto' and len(tasks) == 1))
    if dist.active() and distribute == 'auto' and not shard_frames:
        tasks = tasks[dist.rank()::dist.world_size()]       # folder mode: file i belongs to rank i mod G
    if not tasks:
        return [] if return_results else None
    n_workers = 1 if shard_frames else _worker_count(workers, len(tasks))
    cpu_plan()                                              # on this thread, before any worker asks (torch's device queries
    native = n_workers > 1 and os.environ.get('SHG_SCAN_POOL', 'native') != 'threads'             # are not re-entrant at first use)
    # decoded stacks waiting for a scan: the native pool holds its in-flight scans' stacks itself (workers + 2), so two ahead
    decoder = _Decoder(tasks, dist.frame_block if shard_frames else None, ahead=2 if native else n_workers + 1)
    collected = [None] * len(tasks)

    def scan(i, rdr=None):
        file, options = tasks[i]
        if rdr is None:
            print('file %s is processing' % file)
            options['_shard_frames'] = shard_frames
            rdr = decoder.get(i)
        if shard_frames:
            _check_shardable(rdr)
        if _one_call_ok(rdr, options):
            res = scan_one_call(rdr, options)               # the whole file below the interpreter (shg_scan_file)
            if return_results:
                collected[i] = res
            return
        disk_list, backup_bounds, hdr = solex_read(rdr, options)
        _release_stack(rdr)                                 # the frame stack goes before the next file's lands
        if shard_frames:
            n_requested = sum(1 for s in options['shift'] if s in options['shift_requested'])
            if n_requested > 1:
                options['_deal_disks'] = True               # Doppler stack: every rank post-processes its share of the disks
            elif dist.rank() != 0:
                return
        res = solex_process(options, disk_list, backup_bounds, hdr)
        if return_results:
            collected[i] = res

    try:
        if shard_frames and len(tasks) > 1 and os.environ.get('SHG_SHARD_OVERLAP', '1') != '0':
            _sharded_series(tasks, decoder, collected if return_results else None)
        elif n_workers == 1:
            previous = bind_thread('scan', decoder.device)  # this thread is the scan worker for the duration (device.cpu_plan)
            try:
                for i in range(len(tasks)):
                    scan(i)
            finally:
                if previous is not None:
                    os.sched_setaffinity(0, previous)
        elif native:
            _ensure_lane(decoder.device)
            if os.environ.get('SHG_CHAIN_CUS') or os.environ.get('SHG_CHAIN_CU_MASK'):
                # CU-masked streams are BLOCKING streams (there is no other kind): they synchronise with the legacy null stream, so
                # this thread -- whose current stream every scan's pass A is launched behind -- must not sit on that one
                import torch
                with torch.cuda.stream(_feeder_stream(decoder.device)):
                    _scan_batch_native(tasks, decoder, n_workers, scan, collected if return_results else None)
            else:
                _scan_batch_native(tasks, decoder, n_workers, scan, collected if return_results else None)
        else:
            _ensure_lane(decoder.device)
            _scan_pool(scan, len(tasks), n_workers, decoder.device)
    finally:
        decoder.cancel()
        outputs.flush()
    if not return_results:
        return None
    return collected


def _check_shardable(rdr):
    """Every rank sees the same header, so every rank raises (none is left waiting in the first all-reduce)."""
    dist.refuse_unshardable(rdr.FrameCount, rdr.file)


def _release_stack(rdr):
    """Drop the reader's reference to the frame stack.  The stack was allocated on the decoder's stream; tell the
    caching allocator that this thread's stream still reads it, so the block is not handed to the next decode
    while pass B is queued."""
    import torch
    stack = getattr(rdr, '_stack', None)
    if stack is not None and stack.is_cuda:
        stack.record_stream(torch.cuda.current_stream(stack.device))
    rdr._stack = None


_worker_contexts = {}          # (device, k) -> {'stream', 'buffers'}: a scan worker's stream and staging buffers outlive a batch
_contexts_lock = threading.Lock()
_lanes = {}                    # device -> the frame-pass lane's stream (csrc/streams.hip), created once per process


def _chain_cu_mask(n_cus):
    """The CU mask of the scan workers' streams, as 32-bit words, or None: SHG_CHAIN_CU_MASK=<hex> verbatim, or
    SHG_CHAIN_CUS=<count> CUs spread evenly over the device (every XCD keeps its share).  Off by default."""
    explicit = os.environ.get('SHG_CHAIN_CU_MASK', '').strip()
    words = (n_cus + 31) // 32
    if explicit:
        value = int(explicit, 16) & ((1 << n_cus) - 1)
    else:
        try:
            want = int(os.environ.get('SHG_CHAIN_CUS', '0'))
        except ValueError:
            want = 0
        if want <= 0 or want >= n_cus:
            return None
        value = 0
        for j in range(want):
            value |= 1 << (j * n_cus // want)
    if value == 0:
        return None
    return [(value >> (32 * i)) & 0xffffffff for i in range(words)]


def _native_stream(device, priority=0, cu_mask=None):
    """A stream made by the library (shg_stream_create: priority or CU mask), as a torch stream."""
    import ctypes
    import torch
    from ._lib import check, lib
    out = ctypes.c_void_p()
    with torch.cuda.device(device):
        if cu_mask:
            arr = (ctypes.c_uint32 * len(cu_mask))(*cu_mask)
            check(lib.shg_stream_create(0, arr, len(cu_mask), ctypes.byref(out)), 'shg_stream_create')
        else:
            check(lib.shg_stream_create(priority, None, 0, ctypes.byref(out)), 'shg_stream_create')
    return torch.cuda.ExternalStream(out.value, device=device)


def _ensure_lane(device):
    """The frame-pass lane of this device: ONE stream through which pass A of every scan in flight runs (HBM-bound passes
    side by side only halve each other's bandwidth).  SHG_FRAME_LANE=0 leaves every pass on its scan's own stream."""
    import torch
    from ._lib import check, lib
    if os.environ.get('SHG_FRAME_LANE', '1') == '0':
        return None
    with _contexts_lock:
        lane = _lanes.get(str(device))
        if lane is None:
            lane = _lanes[str(device)] = _native_stream(device, priority=-1)
            with torch.cuda.device(device):
                check(lib.shg_frame_pass_lane_set(lane.cuda_stream), 'shg_frame_pass_lane_set')
    return lane


_feeders = {}


def _feeder_stream(device):
    import torch
    with _contexts_lock:
        st = _feeders.get(str(device))
        if st is None:
            st = _feeders[str(device)] = torch.cuda.Stream(device=device)
    return st


def _worker_context(device, k):
    import ctypes
    import torch
    from ._lib import check, lib
    with _contexts_lock:
        ctx = _worker_contexts.get((str(device), k))
        if ctx is None:
            n_cus = ctypes.c_int(0)
            with torch.cuda.device(device):
                check(lib.shg_device_cu_count(ctypes.byref(n_cus)), 'shg_device_cu_count')
            mask = _chain_cu_mask(n_cus.value)
            stream = _native_stream(device, cu_mask=mask) if mask else torch.cuda.Stream(device=device)
            ctx = {'stream': stream, 'buffers': {}}
            _worker_contexts[(str(device), k)] = ctx
    return ctx


def _sharded_series(tasks, decoder, collected):
    """Several scans, each sharded over the ranks.  Per scan TWO collectives (dist.py): the all-gather of the frame statistics after
    pass A -- which also carries every rank's "I have failed" word, so a failure anywhere stops all ranks at the same scan -- and the
    reduction of the disk mosaic to the scan's owner after pass B.  Three things run at once on every rank:
      * two READING threads (this one takes the even scans, a service thread the odd ones), each on a stream of its own: while scan
        k is fitted and extracted, scan k + 1 is already decoded, summed and exchanged.  Their collectives are issued in one order on
        every rank (dist.Sequencer: statistics of k + 1, then mosaic of k);
      * what has no collective in it, the post-processing of a finished mosaic, on a third thread ON THE SCAN'S OWNER: scan k belongs
        to rank k mod G (dist.scan_owner), the mosaic is reduced to that rank only, and it alone writes the scan's files and log --
        G tails at once, the reference's Pool over files (Solex_recon.py:26-44) spread over the GPUs.
    A Doppler stack (several requested disks) deals its disks to all ranks and agrees on the limb fit -- collectives inside
    solex_process -- so a series with such a scan in it is read by ONE thread, that scan all-reduced and post-processed in line."""
    import torch
    device = decoder.device
    post = _Service.named('shg-post-%s' % device)
    ctx = _worker_context(device, 'post')
    slots = threading.Semaphore(2)                          # mosaics waiting for / in post-processing
    pending = []
    errors = []
    lock = threading.Lock()
    plain = all(len(set(options['shift'])) == 1 for _, options in tasks)
    two_readers = plain and len(tasks) > 1 and os.environ.get('SHG_SHARD_READERS', '2') != '1'

    def post_job(i, options, disk_list, bounds, hdr, ready, done):
        try:
            torch.cuda.set_device(device)
            bind_thread('scan', device)
            stages.use_buffers(ctx['buffers'])
            stream = ctx['stream']
            stream.wait_event(ready)                        # the mosaic was written on the reading thread's stream
            for d in disk_list:
                d.t.record_stream(stream)
                if d.minmax is not None:
                    d.minmax.record_stream(stream)
            with torch.cuda.stream(stream):
                res = solex_process(options, disk_list, bounds, hdr)
                stream.synchronize()
            if collected is not None:
                collected[i] = res
        except BaseException as e:      # noqa: BLE001 -- re-raised on the caller's thread
            with lock:
                errors.append((i, e))
        finally:
            slots.release()
            done.set()

    def read_scan(i):
        """Scan i on the calling thread: decode, pass A, exchange 1, fit, pass B, exchange 2; its mosaic to the owner's post-processing."""
        file, options = tasks[i]
        print('file %s is processing' % file)
        options['_shard_frames'] = True
        n_requested = len(set(options['shift']))
        if n_requested == 1:
            options['_owner'] = options['_mosaic_to'] = dist.scan_owner(i)
        try:
            rdr = decoder.get(i)
            _check_shardable(rdr)
        except BaseException as e:      # noqa: BLE001
            # this rank cannot read its share: the others are on their way into the exchange after pass A -- join them there with the
            # failure word set (the header says how large the message is), so that every rank leaves the series together
            with lock:
                errors.append((i, e))
            hdr_only = video_reader(file) if not hasattr(file, 'device_stack') else file
            p = int(hdr_only.Width) * int(hdr_only.Height)
            dist.exchange_frame_stats(torch.zeros(p, dtype=torch.int64, device=device), torch.zeros(p, dtype=torch.uint16, device=device),
                                      failed=True, n_frames=1)
            raise
        disk_list, bounds, hdr = solex_read(rdr, options)
        _release_stack(rdr)
        if n_requested > 1:
            options['_deal_disks'] = True                   # every rank post-processes its share: collectives inside
            res = solex_process(options, disk_list, bounds, hdr)
            if collected is not None:
                collected[i] = res
        elif dist.rank() == options['_owner']:
            slots.acquire()
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(device))
            done = threading.Event()
            with lock:
                pending.append(done)
            post.jobs.put(functools.partial(post_job, i, options, disk_list, bounds, hdr, ready, done))

    def placeholder_second(i):
        """Scan i failed on THIS rank between its two exchanges: the other ranks are on their way into the reduction of the mosaic --
        join them with an empty one (the scan is lost: this rank's failure word travels in the next exchange after pass A)."""
        file, options = tasks[i]
        rdr = file if hasattr(file, 'device_stack') else video_reader(file)
        n_total, ih = int(rdr.FrameCount), int(max(rdr.Width, rdr.Height))
        shifts = list(dict.fromkeys([options['ellipse_fit_shift'], 0] + list(options.get('shift_requested', options['shift']))))
        dist.gather_columns(lambda out, k0: None, len(shifts), ih, dist.frame_block(n_total), n_total, bool(options['flip_x']), device,
                            dst=options.get('_mosaic_to'))

    @contextlib.contextmanager
    def second_thread():
        rctx = _worker_context(device, 'read1')
        torch.cuda.set_device(device)
        bind_thread('scan', device)
        stages.use_buffers(rctx['buffers'])
        with torch.cuda.stream(rctx['stream']):
            yield
            rctx['stream'].synchronize()

    def posts_done():
        while True:
            with lock:
                waiting = [d for d in pending if not d.is_set()]
            if not waiting:
                return
            waiting[0].wait()

    def post_failed():
        with lock:
            return bool(errors)
    previous = bind_thread('scan', device)
    try:
        read_errors = dist.run_series(len(tasks), read_scan, two_readers=two_readers, second_thread=second_thread, before_verdict=posts_done,
                                      device=device, also_failed=post_failed, placeholder_second=placeholder_second if plain else None)
    finally:
        posts_done()
        if previous is not None:
            os.sched_setaffinity(0, previous)
    with lock:
        errors.extend(e for e in read_errors if not any(e[1] is x[1] for x in errors))
    if errors:
        raise min(errors, key=lambda ie: ie[0])[1]


_native_pools = {}             # (device, workers) -> shg_pool handle; the threads live as long as the process


def _native_pool(device, n_workers):
    """The native scan pool of this device (csrc/pool.hip): n_workers threads, thread k on worker k's stream, placed on the
    cpus the scan workers are meant to run on (device.cpu_plan)."""
    import ctypes
    import torch
    from ._lib import check, lib
    key = (str(device), n_workers)
    pool = _native_pools.get(key)
    if pool is None:
        streams = [_worker_context(device, k)['stream'] for k in range(n_workers)]
        arr = (ctypes.c_void_p * n_workers)(*[st.cuda_stream for st in streams])
        plan = cpu_plan(device)
        cpus = sorted(plan['scan']) if plan else []
        carr = (ctypes.c_int32 * len(cpus))(*cpus) if cpus else None
        out = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(lib.shg_pool_create(arr, n_workers, carr, len(cpus), ctypes.byref(out)), 'shg_pool_create')
        pool = _native_pools[key] = out
    return pool


def _scan_batch_native(tasks, decoder, n_workers, scan_here, collected):
    """The files of a batch through the native scan pool: this ONE thread prepares each scan (buffers, options), submits it
    and, when its turn comes, turns what the pool computed into log lines, files and results -- in file order.  Up to
    n_workers scans run at once (plus two prepared ones waiting), none of them inside the interpreter.  Files the one-call
    route does not cover (de-vignette, stubborn transversalium) run here, stage by stage, between the others.
    The first failure (lowest file index) is re-raised once the scans in flight have finished; no new file starts after it --
    a batch halts on an unsuitable file, as result.get() makes the reference's (Solex_recon.py:42)."""
    import collections
    pool = _native_pool(decoder.device, n_workers)
    # scans handed to the pool at once: the workers' own, and enough queued behind them (their pass A already on the lane) that the
    # lane never waits for a worker to come free
    try:
        depth = int(os.environ.get('SHG_POOL_DEPTH', '0'))
    except ValueError:
        depth = 0
    if depth <= 0:
        depth = n_workers + 8
    inflight = collections.deque()
    errors = []
    previous = bind_thread('scan', decoder.device)

    def finish_oldest():
        i, call = inflight.popleft()
        try:
            res = call.finish()
            if collected is not None:
                collected[i] = res
        except BaseException as e:      # noqa: BLE001 -- re-raised below, lowest index first
            errors.append((i, e))

    try:
        nxt = 0
        while nxt < len(tasks) and not errors:
            if len(inflight) >= depth or (inflight and not decoder.ready[nxt].is_set()):
                finish_oldest()                             # nothing to submit yet: collect (waits for the oldest scan in flight)
                continue
            i, nxt = nxt, nxt + 1
            file, options = tasks[i]
            try:
                print('file %s is processing' % file)
                options['_shard_frames'] = False
                rdr = decoder.get(i)
                if _one_call_ok(rdr, options):
                    inflight.append((i, _OneCall(rdr, options, pooled=True).submit(pool)))
                else:
                    scan_here(i, rdr)
            except BaseException as e:      # noqa: BLE001
                errors.append((i, e))
        while inflight:
            finish_oldest()
    finally:
        # whatever ends this loop early (an interrupt in the feeding thread): the pool's threads still read the requests and
        # write into the buffers of the scans in flight -- those stay alive until the pool has let go of them
        while inflight:
            _, call = inflight.popleft()
            try:
                call.call.wait()
            except BaseException:       # noqa: BLE001 -- the first error is the one that is raised
                pass
        if previous is not None:
            os.sched_setaffinity(0, previous)
    if errors:
        raise min(errors, key=lambda ie: ie[0])[1]


def _scan_pool(scan, n_tasks, n_workers, device):
    """Run scan(0..n_tasks-1) on n_workers threads, each with its own HIP stream (torch's current stream and device
    are thread-local; ops.py launches on the current stream).  Worker k keeps its thread (_Service), stream, device
    workspace and pinned staging buffers from batch to batch (pinning memory and growing the allocator's per-stream pools
    cost milliseconds).
    Tasks are taken in order.  The first failure (lowest task index) is re-raised after the workers have drained, and no
    new task starts once one has failed -- a batch halts on an unsuitable file, as result.get() makes the reference's
    (Solex_recon.py:42)."""
    import torch
    lock = threading.Lock()
    state = {'next': 0, 'errors': []}

    done = threading.Semaphore(0)

    def run(k):
        try:
            work(k)
        except BaseException as e:      # noqa: BLE001 -- e.g. the final stream.synchronize(): fails the batch, is not lost
            with lock:
                state['errors'].append((n_tasks, e))
        finally:
            done.release()

    def work(k):
        try:
            torch.cuda.set_device(device)
            bind_thread('scan', device)                     # one L3 group next to the GPU (device.cpu_plan)
            ctx = _worker_context(device, k)
            stages.use_buffers(ctx['buffers'])
            stream = ctx['stream']
        except BaseException as e:      # noqa: BLE001 -- a worker that cannot start fails the batch instead of vanishing
            with lock:
                state['errors'].append((state['next'], e))
            return
        with torch.cuda.stream(stream):
            while True:
                with lock:
                    i = state['next']
                    if i >= n_tasks or state['errors']:
                        break
                    state['next'] = i + 1
                try:
                    scan(i)
                except BaseException as e:      # noqa: BLE001 -- re-raised on the caller's thread
                    with lock:
                        state['errors'].append((i, e))
                    break
            stream.synchronize()                # results handed back to the caller are complete

    for k in range(n_workers):
        _Service.named('shg-scan-%s-%d' % (device, k)).jobs.put(functools.partial(run, k))
    for _ in range(n_workers):
        done.acquire()
    if state['errors']:
        raise min(state['errors'], key=lambda ie: ie[0])[1]


def _one_call_ok(rdr, options):
    """Whether shg_scan_file covers this scan: everything but frame-sharded scans (collectives between the stages), the
    de-vignette and stubborn-transversalium branches (host SciPy filters between the kernels), AVI input re-laid on the fly,
    and runs that time every stage (timing.enabled).  SHG_SCAN_CALL=0 forces the stage-by-stage route."""
    if os.environ.get('SHG_SCAN_CALL', '1') == '0' or timing.enabled:
        return False
    if options.get('_shard_frames') or dist.is_sharded(rdr):
        return False
    if options['de-vignette'] or (options['transversalium'] and options.get('stubborn_transversalium')):
        return False
    return True


def scan_one_call(rdr, options):
    """solex_read + solex_process for one file as ONE C call, on this thread's stream.  -> [(cc, frame_protus), ...]"""
    return _OneCall(rdr, options).run().finish()


class _OneCall:
    """solex_read + solex_process for one file as ONE C call (stages.ScanCall -> shg_scan_file): the same kernels and the
    same host control plane as the stage-by-stage route, and the same log lines, files and `options` side effects
    (basefich0, shift_requested, shift, ratio_fixe, slant_fix, _transversalium_cache) -- produced by finish() from what the
    call returned.  A scan that fails half-way leaves the log the reference would have left, then raises.
    run(): the call is made here, on this thread's current stream; submit(pool): a native scan pool makes it
    (several scans in flight, none of them holding the interpreter)."""

    def __init__(self, rdr, options, pooled=False):
        self.rdr, self.options = rdr, options
        self.basefich0 = basefich0 = os.path.splitext(str(rdr.file))[0]
        options['basefich0'] = basefich0
        self.log = log = basefich0 + '_log.txt'
        clearlog(log, options)
        logme(log, options, 'Pixel shift : ' + str(options['shift']))
        options['shift_requested'] = options['shift']
        options['shift'] = list(dict.fromkeys([options['ellipse_fit_shift'], 0] + options['shift']))
        self.hdr = make_header(rdr)
        logme(log, options, 'Width, Height : ' + str(rdr.Width) + ' ' + str(rdr.Height))
        logme(log, options, 'Number of frames : ' + str(rdr.FrameCount))
        self.shifts = shifts = options['shift']
        self.requested = [sh in options['shift_requested'] for sh in shifts]
        self.plots = plots = plots_enabled(options)
        self.call = stages.ScanCall(rdr.device_stack(), shifts, self.requested, options, savgol_taps, want_plot_data=plots,
                                    want_fit_image=plots, own_buffers=pooled)

    def run(self):
        self.call.run()
        return self

    def submit(self, pool):
        self.call.submit(pool)
        return self

    def done(self):
        return self.call.done()

    def finish(self):
        """-> [(cc, frame_protus), ...] like solex_process; raises what the scan raised."""
        from .ellipse_to_circle import _log_geometry, _warp_geometry
        rdr, options, hdr, log, basefich0 = self.rdr, self.options, self.hdr, self.log, self.basefich0
        shifts, requested, plots = self.shifts, self.requested, self.plots
        r, error = self.call.wait().collect()
        self.call = None
        _release_stack(rdr)
        phase = r['phase']
        # ---- compute_mean_return_fit's outputs (solex_util.py:191-274) ----
        if phase >= 1:
            mean_img = DeviceImage(r['mean'])
            y1, y2 = r['y1'], r['y2']
            if options['save_fit']:
                outputs.submit(write_fits, output_path(basefich0 + '_mean.fits', options), mean_img, hdr)
            logme(log, options, 'Vertical limits y1, y2 : ' + str(y1) + ' ' + str(y2))
            logme(log, options, lambda: 'Spectral line polynomial fit: ' + str(r['p']))
            if plots:
                rows = np.arange(y1, y2)
                good = r['mask_good']
                outputs.submit(outputs.plot_spectral_line, output_path(basefich0 + '_spectral_line_data.png', options), mean_img,
                               r['sharp'].astype(np.int64)[y1:y2][good], rows[good], r['fit'][:, 3], int(rdr.ih), (y2 - y1) // 20 + 1)
        # ---- the raw disks (Solex_recon.py:65-83) and solex_process's header lines (:95-102) ----
        disk_list = []
        if phase >= 2:
            hdr['NAXIS1'] = rdr.iw
            if options['save_fit'] or plots or (phase >= 3 and not r['limb_fitted'] and requested[0] and '_nolog' not in options):
                disk_list = [DeviceImage(r['disks'][i], minmax=r['extrema'][i]) for i in range(len(shifts))]
            if options.get('_keep_raw'):                    # tests and bench.py's parity legs: the raw disks of a one-call scan
                options['_raw_disks'] = [DeviceImage(r['disks'][i]) for i in range(len(shifts))]
            for i, disk in enumerate(disk_list):
                if options['save_fit'] and requested[i]:
                    outputs.submit(write_fits, output_path(basefich0 + '_shift=' + str(shifts[i]) + '_raw.fits', options), disk, hdr)
            if options['transversalium']:
                logme(log, options, 'Transversalium correction : ' + str(options['trans_strength']))
            else:
                logme(log, options, 'Transversalium disabled')
            logme(log, options, 'Mirror X : ' + str(options['flip_x']))
            logme(log, options, 'Post-rotation : ' + str(options['img_rotate']) + ' degrees')
            logme(log, options, f'Protus adjustment : {options["delta_radius"]}')
            logme(log, options, f'de-vignette : {options["de-vignette"]}')
        # ---- the geometry: ellipse_to_circle's / correct_image's log lines (ellipse_to_circle.py:131-143, 313) ----
        if phase >= 3:
            if r['limb_fitted']:
                options['ratio_fixe'] = r['ratio']
                _log_geometry(options, r['phi'], r['ratio'], r['theta_first'], np.array(r['circle'][:2]), r['circle'][2])
                print('sun borders found:' + str(r['borders']))
                options['slant_fix'] = math.degrees(r['phi'])
            elif requested[0] and '_nolog' not in options:
                # correct_image(..., center (-1, -1), height -1, print_log=True) of the first disk (Solex_recon.py:122)
                ih_, n_ = disk_list[0].shape
                theta, inv_mat, _, _, _, origin, det = _warp_geometry(float(r['phi']), float(r['ratio']), int(ih_), int(n_))
                _log_geometry(options, r['phi'], r['ratio'], theta, (inv_mat @ np.array([-1.0, -1.0]).T).T - origin,
                              -1.0 * np.sqrt(np.abs(r['ratio'] / det)), known=False)
        if error is not None:
            raise error
        # ---- single_image_process's files (Solex_recon.py:136-174) ----
        names = [basefich0 + '_shift=' + str(shifts[i]) for i in range(len(shifts)) if requested[i]]
        if plots and r['limb_fitted']:
            fix_img = r['frames'][0] if requested[0] else r['fit_image']
            outputs.submit(outputs.plot_ellipse_fit, output_path(basefich0 + '_shift=' + str(shifts[0]) + '_ellipse_fit.png', options),
                           disk_list[0], DeviceImage(fix_img), r['raw_X'], r['X_f'], r['outline'], r['borders'])
        if options['save_fit']:
            for j, basefich in enumerate(names):
                outputs.submit(write_fits, output_path(basefich + '_circular.fits', options), DeviceImage(r['frames'][j]), hdr)
        if options['transversalium']:
            for i, basefich in enumerate(names):
                c = r['factors'][i]
                options['_transversalium_cache'] = c
                if plots:
                    outputs.submit(outputs.plot_transversalium, output_path(basefich + '_transversalium_correction.png', options), c)
                if options['save_fit']:
                    outputs.submit(write_fits, output_path(basefich + '_detransversaliumed.fits', options), DeviceImage(r['detrans'][i]), hdr)
        if '_nolog' in options and not options['save_fit'] and options['img_rotate'] // 90 % 4 == 0:
            # nothing is written and nothing turned: only the two returned images are ever looked at
            return [(DeviceImage(r['cc'][i]), DeviceImage(r['protus'][i])) for i in range(len(names))]
        results = [write_products(r['final'][i], r['cl1'][i], r['hc'][i], r['protus'][i], r['cc'][i], options, hdr, names[i])
                   for i in range(len(names))]
        for _ in names:
            write_complete(log, options)
        return results


def _writes_files(options):
    """Of a frame-sharded scan only its owner writes (rank 0; in a series of sharded scans rank k mod G for scan k)."""
    return not (options.get('_shard_frames') and dist.rank() != options.get('_owner', 0))


def solex_read(file, options):
    """Read one scan; return (disk_list, (backup_y1, backup_y2), hdr).  disk_list[i] is the raw
    uint16 disk [ih, FrameCount] for options['shift'][i], in HBM."""
    rdr = file if hasattr(file, 'device_stack') else video_reader(file)
    basefich0 = os.path.splitext(str(rdr.file))[0]
    options['basefich0'] = basefich0
    # ranks other than 0 of a frame-sharded scan compute the same mosaic but write nothing
    wopts = options if _writes_files(options) else dict(options, _nolog=True, save_fit=False)
    clearlog(basefich0 + '_log.txt', wopts)
    logme(basefich0 + '_log.txt', wopts, 'Pixel shift : ' + str(options['shift']))
    options['shift_requested'] = options['shift']
    # ellipse_fit_shift and 0 are "fake" shifts; if requested they are not double counted (:55)
    options['shift'] = list(dict.fromkeys([options['ellipse_fit_shift'], 0] + options['shift']))
    if options.get('_shard_frames') and getattr(rdr, '_stack', None) is None:
        rdr.frame_range = dist.frame_block(int(rdr.FrameCount))
    hdr = make_header(rdr)
    ih, iw = rdr.ih, rdr.iw

    with timing.stage('mean_max+line_fit'):
        mean_img, fit, backup_y1, backup_y2 = compute_mean_return_fit(rdr, wopts, hdr, iw, ih, basefich0)
    with timing.stage('extract'):
        disks, extrema = extract_disks(rdr, fit, options['shift'], flip_x=bool(options['flip_x']), want_minmax=True,     # flip fused (:74-76)
                                       owner=options.get('_mosaic_to'))
    hdr['NAXIS1'] = iw          # as the reference (:65); the FITS writer takes NAXIS* from the data anyway

    disk_list = [DeviceImage(disks[i], minmax=None if extrema is None else extrema[i]) for i in range(disks.shape[0])]
    for i, disk in enumerate(disk_list):
        basefich = basefich0 + '_shift=' + str(options['shift'][i])
        flag_requested = options['shift'][i] in options['shift_requested']
        if wopts['save_fit'] and flag_requested:
            outputs.submit(write_fits, output_path(basefich + '_raw.fits', options), disk, hdr)
    return disk_list, (backup_y1, backup_y2), hdr


def solex_process(options, disk_list, backup_bounds, hdr):
    """Circularise, de-transversalium, crop and contrast every requested disk."""
    basefich0 = options['basefich0']
    if options.get('_deal_disks') and dist.active() and dist.rank() != 0:
        options = _no_log(options)                            # products yes, shared log file no
    if options['transversalium']:
        logme(basefich0 + '_log.txt', options, 'Transversalium correction : ' + str(options['trans_strength']))
    else:
        logme(basefich0 + '_log.txt', options, 'Transversalium disabled')
    logme(basefich0 + '_log.txt', options, 'Mirror X : ' + str(options['flip_x']))
    logme(basefich0 + '_log.txt', options, 'Post-rotation : ' + str(options['img_rotate']) + ' degrees')
    logme(basefich0 + '_log.txt', options, f'Protus adjustment : {options["delta_radius"]}')
    logme(basefich0 + '_log.txt', options, f'de-vignette : {options["de-vignette"]}')
    borders = [0, 0, 0, 0]
    cercle0 = (-1, -1, -1)
    results = []
    pending = []
    # Frame-sharded Doppler stack: all ranks hold every raw disk after the gather; rank 0 fits the limb once and
    # broadcasts the geometry, then the requested disks are dealt round-robin (rank 0 keeps the log file).
    deal = bool(options.get('_deal_disks')) and dist.active()
    turn = 0
    for i in range(len(disk_list)):
        flag_requested = options['shift'][i] in options['shift_requested']
        basefich = basefich0 + '_shift=' + str(options['shift'][i])
        mine = True
        if deal and flag_requested:
            mine = dist.my_share(turn)
            turn += 1
        # disk_list[0] is always the ellipse-fit shift (more limb contrast)
        if options['ratio_fixe'] is None and options['slant_fix'] is None:
            def fit_limb():
                # (the corrected image of the ellipse-fit shift is only computed when somebody uses it: a requested
                # disk, or the diagnostic plot)
                with timing.stage('ellipse_fit+warp'):
                    return ellipse_to_circle(disk_list[i], options, basefich, need_image=flag_requested and mine)
            if not deal:
                frame_circularized, cercle0, options['ratio_fixe'], phi, borders = fit_limb()
            else:
                # rank 0 fits, every rank learns the geometry -- or fails with rank 0 (dist.agree) instead of waiting for it
                kept = {}

                def fit_and_keep():
                    kept['frame'], c, ratio, ph, b = fit_limb()
                    return c, ratio, ph, b
                cercle0, options['ratio_fixe'], phi, borders = dist.agree(fit_and_keep)
                if dist.rank() == 0:
                    frame_circularized = kept['frame']
                elif flag_requested and mine:
                    # the same warp the fit ran on rank 0 (centre / height only feed the returned circle)
                    frame_circularized = correct_image(disk_list[i], phi, options['ratio_fixe'], np.array([-1.0, -1.0]),
                                                       -1.0, dict(options, _nolog=True))[0]
            options['slant_fix'] = math.degrees(phi)          # stored in degrees (:117)
        else:
            ratio = options['ratio_fixe'] if options['ratio_fixe'] is not None else 1.0
            phi = math.radians(options['slant_fix']) if options['slant_fix'] is not None else 0.0
            if flag_requested and mine:
                with timing.stage('warp'):
                    frame_circularized = correct_image(disk_list[i], phi, ratio, np.array([-1.0, -1.0]), -1.0, options,
                                                       print_log=i == 0)[0]
                if options['de-vignette']:
                    if cercle0 == (-1, -1, -1):
                        print("WARNING: cannot de-vignette without ellipse fit")
                    else:
                        frame_circularized = removeVignette(frame_circularized, cercle0)
        if not flag_requested or not mine:
            continue
        pending.append((frame_circularized, basefich))
    # every requested disk of the file shares the geometry (cercle0 / borders are those of the one limb fit), so
    # the per-disk stages run as a batch: all kernels of a stage are launched before its single device->host read
    popts = options if (not deal or dist.rank() == 0) else _no_log(options)
    if pending:
        results = process_images([f for f, _ in pending], hdr, popts, cercle0, borders, [b for _, b in pending], backup_bounds)
        for _ in pending:
            write_complete(basefich0 + '_log.txt', popts)
    return results


class _no_log(dict):
    """options of a rank that writes products but leaves the shared log file to rank 0:
    logme / clearlog / write_complete look for the '_log_off' key."""

    def __init__(self, options):
        super().__init__(options)
        self['_log_off'] = True


def single_image_process(frame_circularized, hdr, options, cercle0, borders, basefich, backup_bounds):
    return process_images([frame_circularized], hdr, options, cercle0, borders, [basefich], backup_bounds)[0]


def crop_to_width(images, cercle, options):
    """The crop / pad block of single_image_process (Solex_recon.py:155-171) for a list of same-shape images.
    Returns (images, cercle) with cx moved to the new centre."""
    # A de-vignetted frame that skipped the transversalium stage is still float64 here; the reference crops
    # the float image and truncates in image_process (solex_util.py:528).  Cropping is a pure copy, so
    # truncating first gives the same pixels (and the same fill value img[0, 0]).
    images = [as_uint16_image(img) for img in images]
    h, w = images[0].shape
    plan, cercle = crop_plan(h, w, cercle, options)
    if plan is not None:
        nw, lo, dx0, n = plan
        # the fill colour img[0, 0] is read on the device (no host round trip per image)
        images = [DeviceImage(ops.crop_pad_u16(to_device_u16(img), nw, lo, dx0, n, None)) for img in images]
    return images, cercle


def process_images(frames, hdr, options, cercle0, borders, basefichs, backup_bounds):
    """single_image_process (Solex_recon.py:136-174) for a list of circularised frames of one file:
    transversalium, crop, CLAHE + contrast products.  Returns [(cc, frame_protus), ...].
    The usual case is one stage call (shg_stage_process_frames); de-vignetted frames (float64 row factors) and the
    stubborn transversalium branch take the step-by-step route."""
    if options['save_fit']:
        for frame, basefich in zip(frames, basefichs):
            outputs.submit(write_fits, output_path(basefich + '_circular.fits', options), _as_image(frame), hdr)
    factored = any(isinstance(f, DeviceImage) and f.row_factor is not None for f in frames)
    if factored or (options['transversalium'] and options.get('stubborn_transversalium')):
        return _process_images_stepwise(frames, hdr, options, cercle0, borders, basefichs, backup_bounds)

    tensors = [to_device_u16(f) for f in frames]
    if any(t.shape != tensors[0].shape or t.stride() != tensors[0].stride() or t.stride(1) != 1 for t in tensors):
        tensors = [t.contiguous() for t in tensors]
    h, w = tensors[0].shape
    trans = None
    if options['transversalium']:
        if not cercle0 == (-1, -1, -1):
            circle, bds = cercle0, borders
        else:                                               # no limb fit: the sunlit rows found by the line fit (:146)
            circle, bds = (0, 0, 99999), [0, backup_bounds[0] + 20, w - 1, backup_bounds[1] - 20]
        y1 = math.ceil(max(circle[1] - circle[2], bds[1]))
        y2 = math.floor(min(circle[1] + circle[2], bds[3]))
        window = savgol_window(max(y2 - y1, 1), options['trans_strength'])
        trans = {'circle': circle, 'borders': bds, 'window': window, 'taps': savgol_taps(window)}
    crop, cercle = crop_plan(h, w, cercle0, options)
    disc = None
    if not cercle == (-1, -1, -1) and options['disk_display']:
        r = int(cercle[2]) + options['delta_radius']
        if r > 0:
            disc = (int(cercle[0]), int(cercle[1]), r)
    with timing.stage('transversalium+crop+clahe+contrast'):
        res = stages.process_frames(tensors, trans, crop, disc, keep_detrans=bool(options['save_fit'] and trans))
    if trans is not None:
        for i, basefich in enumerate(basefichs):
            c = res['factors'][i]
            options['_transversalium_cache'] = c
            if plots_enabled(options):
                outputs.submit(outputs.plot_transversalium, output_path(basefich + '_transversalium_correction.png', options), c)
            if options['save_fit']:
                outputs.submit(write_fits, output_path(basefich + '_detransversaliumed.fits', options), DeviceImage(res['detrans'][i]), hdr)
    return [write_products(res['final'][i], res['cl1'][i], res['hc'][i], res['protus'][i], res['cc'][i], options, hdr, basefichs[i])
            for i in range(len(tensors))]


def crop_plan(h, w, cercle, options):
    """The crop / pad block of single_image_process (Solex_recon.py:155-171) as numbers: centre on int(cx) (w // 2
    without a circle), crop or pad to `fixed_width` (or to the height for `crop_width_square`), fill with img[0, 0]:
    new[:, dx0:dx0+n] = img[:, lo:lo+n].  -> ((nw, lo, dx0, n) or None, cercle with cx moved to the new centre)."""
    if options['fixed_width'] is None and not options['crop_width_square']:
        return None, cercle
    nw = h if options['fixed_width'] is None else options['fixed_width']
    nw2 = nw // 2
    cx = w // 2 if cercle == (-1, -1, -1) else int(cercle[0])
    tx = nw2 - cx
    lo, hi = max(0, cx - nw2), min(cx + nw2, w)
    if hi < lo:
        raise ValueError('crop window [%d, %d) lies outside the %d px wide image' % (cx - nw2, cx + nw2, w))
    # new_img[:, :hi-lo] = img[:, lo:hi]; then np.roll by tx when tx > 0 and refill the first tx columns (:161-167).
    # The rolled-in tail is fill colour whenever the copied span fits, which it does: hi - lo <= nw - tx.
    n = hi - lo
    dx0 = tx if tx > 0 else 0
    if dx0 + n > nw:
        n = nw - dx0            # np.roll would wrap these columns round and the refill overwrite them
    if not cercle == (-1, -1, -1):
        cercle = (nw2, cercle[1], cercle[2])
    return (nw, lo, dx0, n), cercle


def _process_images_stepwise(frames, hdr, options, cercle0, borders, basefichs, backup_bounds):
    """process_images stage by stage (correct_transversalium2_batch, crop_to_width, image_process_batch): the route of
    de-vignetted frames and of the stubborn transversalium branch."""
    with timing.stage('transversalium'):
        if options['transversalium']:
            if not cercle0 == (-1, -1, -1):
                detrans = correct_transversalium2_batch(frames, cercle0, borders, options, 0, basefichs)
            else:
                detrans = correct_transversalium2_batch(
                    frames, (0, 0, 99999),
                    [0, backup_bounds[0] + 20, frames[0].shape[1] - 1, backup_bounds[1] - 20], options, 0, basefichs)
        else:
            detrans = list(frames)

    if options['save_fit'] and options['transversalium']:
        for img, basefich in zip(detrans, basefichs):
            outputs.submit(write_fits, output_path(basefich + '_detransversaliumed.fits', options), _as_image(img), hdr)

    detrans, cercle = crop_to_width(detrans, cercle0, options)

    with timing.stage('clahe+contrast'):
        return image_process_batch(detrans, cercle, options, hdr, basefichs)


def _as_image(x):
    return x if isinstance(x, DeviceImage) else DeviceImage(to_device_u16(x))
