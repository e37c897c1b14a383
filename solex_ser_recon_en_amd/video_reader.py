"""SER decode: header parse, host frame iterator, and the HBM upload.

Mirrors the reference's video_reader surface (video_reader.py:10-126): attributes
Width Height FrameCount FrameIndex count infilebytes infiledatatype flag_rotate ih iw
and methods next_frame() / has_frames().  New here: device_stack(), which streams the
file through pinned host buffers into one [N, Height, Width] tensor in HBM on a copy
stream (double-buffered, asynchronous hipMemcpy) -- the file is read ONCE and both
frame passes run on the resident stack (the reference decodes it twice,
Solex_recon.py:61-63).  The rotation of wide frames (video_reader.py:119-120) and the
8-bit x256 widening (:121-122) are never materialised for the stack: the kernels index
the file layout directly.  AVI: uncompressed streams (8-bit grey, 8-bit palettised, 24-bit DIB) are
indexed by avi_io.AviIndex, uploaded as they lie in the file and re-laid into the same uint8 stack by
shg_unpack_dib_frames; compressed AVI needs a codec and raises.
"""
import itertools
import os
import threading

import numpy as np
import torch

from .device import default_device

SER_HEADER_BYTES = 178

_pinned_free = []
_pinned_lock = threading.Lock()


def _lease_pinned_pair(nbytes):
    """Two pinned staging buffers for one reader thread, leased from a process-wide pool and
    returned afterwards (pinning 64 MB costs more than copying it; concurrent decodes must
    not share a buffer)."""
    with _pinned_lock:
        for i, pair in enumerate(_pinned_free):
            if pair[0].numel() >= nbytes:
                return _pinned_free.pop(i)
    return [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]


def _return_pinned_pair(pair):
    with _pinned_lock:
        _pinned_free.append(pair)


class _UploadJob:
    """The chunks of one file on their way to HBM: done when the last copy has landed (or a chunk failed)."""

    def __init__(self, n_chunks):
        self.left = n_chunks
        self.errors = []
        self.lock = threading.Lock()
        self.done = threading.Event()
        if n_chunks <= 0:
            self.done.set()

    def chunk_done(self, error=None):
        with self.lock:
            if error is not None:
                self.errors.append(error)
            self.left -= 1
            if self.left <= 0:
                self.done.set()


def _direct_mode():
    """SHG_READ_DIRECT = 0 (never), 1 (whenever the file system takes O_DIRECT), auto (default: O_DIRECT for files that are
    not in the page cache)."""
    v = os.environ.get('SHG_READ_DIRECT', 'auto').strip().lower()
    return v if v in ('0', '1') else 'auto'


_DIRECT_ALIGN = 4096                     # O_DIRECT wants offset, length and address on the device's logical block size


def _page_cache_share(fd, offset, nbytes):
    """Share of [offset, offset + nbytes) of the file that the page cache holds (mincore over a mapping nobody touches), or
    1.0 when that cannot be asked: a cached file is read through the cache at memcpy speed, a cold one is better read
    straight into pinned memory."""
    import ctypes
    import mmap
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        libc.mmap.restype = ctypes.c_void_p
        libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
        libc.munmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        libc.mincore.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        page = mmap.PAGESIZE
        lo = offset // page * page
        length = offset + nbytes - lo
        addr = libc.mmap(None, length, mmap.PROT_READ, mmap.MAP_SHARED, fd, lo)
        if addr is None or addr == ctypes.c_void_p(-1).value:
            return 1.0
        try:
            n_pages = (length + page - 1) // page
            vec = (ctypes.c_ubyte * n_pages)()
            if libc.mincore(addr, length, vec) != 0:
                return 1.0
            return float(np.count_nonzero(np.frombuffer(vec, dtype=np.uint8) & 1)) / float(n_pages)
        finally:
            libc.munmap(addr, length)
    except (OSError, ValueError, AttributeError):
        return 1.0


class _FileHandles:
    """The descriptors a reader thread holds for the file it is working through: the buffered one and, where the file system
    takes it and the file is cold, an O_DIRECT one (file -> pinned buffer by DMA, no copy through the page cache)."""

    def __init__(self):
        self.path, self.fd, self.dfd = None, -1, -1

    def open(self, path, offset, nbytes):
        if path == self.path:
            return
        self.close()
        self.fd = os.open(path, os.O_RDONLY)
        self.path = path
        mode = _direct_mode()
        if mode != '0' and hasattr(os, 'O_DIRECT'):
            try:
                if mode == '1' or _page_cache_share(self.fd, offset, min(nbytes, 1 << 20)) < 0.5:       # (a sample: 256 pages, a few microseconds)
                    self.dfd = os.open(path, os.O_RDONLY | os.O_DIRECT)
            except OSError:                                  # tmpfs and friends: EINVAL
                self.dfd = -1

    def close(self):
        for fd in (self.fd, self.dfd):
            if fd >= 0:
                os.close(fd)
        self.path, self.fd, self.dfd = None, -1, -1


def _read_chunk(handles, mv, offset, nbytes):
    """[offset, offset + nbytes) of the file into the pinned buffer behind `mv` (page-aligned, _DIRECT_ALIGN bytes longer than
    the largest chunk).  -> the position of the first wanted byte in the buffer.  O_DIRECT reads the enclosing aligned span (the
    SER header is 178 bytes: no frame starts on a block); a short or refused direct read falls back to the buffered one."""
    if handles.dfd >= 0:
        lo = offset // _DIRECT_ALIGN * _DIRECT_ALIGN
        span = (offset + nbytes - lo + _DIRECT_ALIGN - 1) // _DIRECT_ALIGN * _DIRECT_ALIGN
        try:
            got = 0
            while got < offset + nbytes - lo:
                r = os.preadv(handles.dfd, [mv[got:span]], lo + got)
                if r <= 0 or (r % _DIRECT_ALIGN and got + r < offset + nbytes - lo):
                    raise OSError('short direct read')
                got += r
            return offset - lo
        except OSError:
            os.close(handles.dfd)
            handles.dfd = -1
    got = 0
    while got < nbytes:
        r = os.preadv(handles.fd, [mv[got:nbytes]], offset + got)
        if r <= 0:
            raise Exception('error input file ' + str(handles.path) + ': short read')
        got += r
    return 0


class _Uploader:
    """File -> pinned host -> HBM for one device, as a standing service: `readers` long-lived threads, each with two pinned
    staging buffers and a copy stream of its own, take (file, offset, bytes, destination) chunks from one queue -- of whichever
    file is being decoded, and of the next one as soon as its chunks are queued, so the link does not idle across a file
    boundary (before: eight threads, eight streams and sixteen buffer leases per file, and a drain at every file's end).
    A chunk is one preadv() of whole frames into pinned memory (no interpreter lock; O_DIRECT for a file the page cache does
    not hold, see _read_chunk) and one asynchronous 2-D hipMemcpy into the (8 KiB-pitched) frames of the stack.  One service
    per (device, readers): a reader's buffers grow when a file with larger chunks comes along."""

    _all = {}
    _guard = threading.Lock()

    def __init__(self, device, readers):
        import queue
        self.device = device
        self.jobs = queue.SimpleQueue()
        self.threads = [threading.Thread(target=self._reader, args=(i,), name='shg-upload-%s-%d' % (device, i), daemon=True)
                        for i in range(readers)]
        for t in self.threads:
            t.start()

    @classmethod
    def of(cls, device, readers):
        key = (str(device), int(readers))
        with cls._guard:
            up = cls._all.get(key)
            if up is None or not all(t.is_alive() for t in up.threads):
                up = cls._all[key] = cls(device, readers)
            return up

    def _reader(self, tid):
        import queue
        from . import _lib
        from .device import bind_thread
        torch.cuda.set_device(self.device)
        bind_thread('io', self.device)                       # off the scan workers' cores (device.cpu_plan)
        stream = torch.cuda.Stream(device=self.device)
        bufs = [None, None]
        pending = [None, None]                               # per buffer: (event of its copy, job) still in flight
        handles = _FileHandles()
        slot = 0

        def settle(i):
            if pending[i] is not None:
                ev, job = pending[i]
                pending[i] = None
                try:
                    ev.synchronize()
                    job.chunk_done()
                except BaseException as e:      # noqa: BLE001
                    job.chunk_done(e)

        while True:
            # One atomic fetch: "is the queue empty?" followed by a blocking get() let another reader take the last chunk in
            # between, and this one then slept with a copy in flight whose chunk_done() nobody would ever call.
            try:
                item = self.jobs.get_nowait()
            except queue.Empty:                              # nothing queued: what is in flight lands before we sleep
                settle(0)
                settle(1)
                handles.close()
                item = self.jobs.get()
            path, offset, nbytes, dst, dst_pitch, frame_bytes, m, ready, job = item
            settle(slot)                                     # this buffer's previous copy has landed
            try:
                need = nbytes + 2 * _DIRECT_ALIGN
                if bufs[slot] is None or bufs[slot].numel() < need:
                    bufs[slot] = None
                    bufs[slot] = torch.empty(need, dtype=torch.uint8).pin_memory()
                view = bufs[slot]
                handles.open(path, offset, nbytes)
                skip = _read_chunk(handles, memoryview(view.numpy()), offset, nbytes)
                if ready is not None:
                    stream.wait_event(ready)                 # the stack may be a recycled block: its allocating stream drains first
                _lib.check(_lib.lib.shg_upload_frames(dst, dst_pitch, view.data_ptr() + skip, frame_bytes, m, stream.cuda_stream), 'shg_upload_frames')
                ev = torch.cuda.Event()
                ev.record(stream)
                pending[slot] = (ev, job)
                slot ^= 1
            except BaseException as e:      # noqa: BLE001 -- reported to whoever waits for the file
                job.chunk_done(e)


class video_reader:
    def __init__(self, file, buffer_size=25, frame_range=None):
        self.file = file
        self.buffer_size = buffer_size
        upper = str(file).upper()
        self._mm = None
        self._stack = None
        self._upload = None
        if upper.endswith('.AVI'):                              # video_reader.py:20-23, 68-80
            from .avi_io import AviIndex
            self.SER_flag, self.AVI_flag = False, True
            self._avi = AviIndex(file)
            self.infiledatatype, self.infilebytes = 'uint8', 1
            self.Width, self.Height = self._avi.width, self._avi.height
            self.PixelDepthPerPlane = 8
            self.FrameCount = self._avi.frame_count
            self.count = self.Width * self.Height
            self.FrameIndex = -1
            self.offset = self.fileoffset = 0
            self._set_orientation(frame_range)
            return
        if not upper.endswith('.SER'):
            raise Exception('error input file ' + file + 'neither is SER nor AVI')        # video_reader.py:26
        with open(file, 'rb') as f:
            head = f.read(SER_HEADER_BYTES)
        if len(head) < SER_HEADER_BYTES:
            raise Exception('error input file ' + file + ': truncated SER header')
        self.FileID = np.frombuffer(head, dtype='int8', count=14)
        self.LuID, self.ColorID, self.littleEndian = (np.frombuffer(head, '<u4', 1, 14 + 4 * i) for i in range(3))
        fields = np.frombuffer(head, '<u4', 4, 26)          # Width @26, Height @30, depth @34, FrameCount @38
        self.Width, self.Height = fields[0], fields[1]
        self.PixelDepthPerPlane = fields[2]
        self.FrameCount = fields[3]
        self.SER_flag, self.AVI_flag = True, False
        if self.PixelDepthPerPlane == 8:
            self.infiledatatype, self.infilebytes = 'uint8', 1
        else:
            self.infiledatatype, self.infilebytes = 'uint16', 2
        self.count = self.Width * self.Height
        self.FrameIndex = -1
        self.offset = self.fileoffset = SER_HEADER_BYTES
        self._set_orientation(frame_range)

    def _set_orientation(self, frame_range):
        if self.Width > self.Height:                        # video_reader.py:84-91
            self.flag_rotate, self.ih, self.iw = True, self.Width, self.Height
        else:
            self.flag_rotate, self.iw, self.ih = False, self.Width, self.Height
        # frame block owned by this process (multi-GPU sharding); FrameCount stays the scan length
        self.frame_range = (0, int(self.FrameCount)) if frame_range is None else (int(frame_range[0]), int(frame_range[1]))

    # ---- host iterator (compatibility; not used by the GPU path) ------------
    def _memmap(self):
        if self._mm is None:
            self._mm = np.memmap(self.file, dtype='<u2' if self.infilebytes == 2 else np.uint8, mode='r',
                                 offset=SER_HEADER_BYTES,
                                 shape=(int(self.FrameCount), int(self.Height), int(self.Width)))
        return self._mm

    def has_frames(self):
        return self.FrameIndex + 1 < self.FrameCount

    def next_frame(self):
        self.FrameIndex += 1
        self.offset = self.fileoffset + self.FrameIndex * int(self.count) * self.infilebytes
        img = self._avi.frame(self.FrameIndex) if self.AVI_flag else np.array(self._memmap()[self.FrameIndex])
        if self.flag_rotate:
            img = np.rot90(img)
        if self.infilebytes == 1:
            img = np.asarray(img, dtype='uint16') * 256
        return img

    def reset(self):
        self.FrameIndex = -1

    # ---- decode into HBM -----------------------------------------------------
    def device_stack(self, device=None, chunk_bytes=16 << 20, readers=None):
        """Frames [k0:k1) of the file as one tensor [n, Height, Width] in HBM (file layout).

        The device's upload service (_Uploader: `readers` threads, default min(8, cpus), each with two pinned staging buffers
        and a copy stream) takes the file in chunks of whole frames: preadv() into pinned memory (no interpreter lock), one
        asynchronous 2-D hipMemcpy into the (8 KiB-pitched) frames of the stack, next chunk.  One thread tops out at the
        page-cache memcpy rate (~12 GB/s measured); several saturate the PCIe link (tools/sweep_decode.py: 8 readers x 16 MB
        chunks 50 GB/s; 32 MB 47.7, 64 MB 42.6, 4 MB 45.8; 4 readers 33; more than 8 readers no faster).
        begin_device_stack() only queues the chunks: a caller that decodes a series of files queues the next file before it
        waits for this one, and the link stays busy across the boundary."""
        if self._stack is not None:
            return self._stack
        self.begin_device_stack(device, chunk_bytes, readers)
        if self._upload is None:
            return self._stack
        stack, job = self._upload
        job.done.wait()
        self._upload = None
        if job.errors:
            raise job.errors[0]
        self._stack = stack
        return stack

    def begin_device_stack(self, device=None, chunk_bytes=16 << 20, readers=None):
        if self._stack is not None or getattr(self, '_upload', None) is not None:
            return
        self._upload = None
        device = device or default_device()
        if self.AVI_flag:
            self._stack = self._avi_device_stack(device, chunk_bytes)
            return
        k0, k1 = self.frame_range
        n = k1 - k0
        h, w, b = int(self.Height), int(self.Width), self.infilebytes
        if n <= 0 or h <= 0 or w <= 0:
            raise Exception('error input file ' + str(self.file) + ': no frames')
        frame_bytes = h * w * b
        if os.path.getsize(self.file) < SER_HEADER_BYTES + int(self.FrameCount) * frame_bytes:
            raise Exception('error input file ' + str(self.file) + ': shorter than its header says')
        from . import ops
        dt = torch.uint16 if b == 2 else torch.uint8
        stack = ops.padded_stack(n, h, w, dt, device)                  # frame pitch rounded up to 8 KiB
        pitch_bytes = stack.stride(0) * b if n > 1 else frame_bytes
        base_ptr = stack.data_ptr()
        base = SER_HEADER_BYTES + k0 * frame_bytes
        chunk_frames = max(1, chunk_bytes // frame_bytes)
        n_chunks = (n + chunk_frames - 1) // chunk_frames
        readers = max(1, readers or min(8, os.cpu_count() or 1))
        up = _Uploader.of(device, readers)
        # the stack may be a block the caching allocator recycled from the previous file: kernels queued on the allocating
        # stream may still read it, so no upload starts before that stream has drained up to here
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(device))
        job = _UploadJob(n_chunks)
        for c in range(n_chunks):
            f0 = c * chunk_frames
            m = min(chunk_frames, n - f0)
            up.jobs.put((self.file, base + f0 * frame_bytes, m * frame_bytes, base_ptr + f0 * pitch_bytes, pitch_bytes, frame_bytes, m, ready, job))
        self._upload = (stack, job)

    def _avi_device_stack(self, device, chunk_bytes):
        """Frames [k0:k1) of an uncompressed AVI as the uint8 [n, Height, Width] stack: chunk payloads go to HBM as
        they lie in the file (one contiguous read per block when the chunks are evenly spaced, which is what capture
        programs write), then one re-layout kernel (row order, row padding, palette / BGR -> grey)."""
        from . import _lib, ops
        avi = self._avi
        k0, k1 = self.frame_range
        n = k1 - k0
        if n <= 0:
            raise Exception('error input file ' + str(self.file) + ': no frames')
        pitch = avi.stride if avi.stride is not None else avi.payload_bytes
        raw = torch.empty(n * pitch, dtype=torch.uint8, device=device)
        chunk_frames = max(1, chunk_bytes // pitch)
        bufs = _lease_pinned_pair(chunk_frames * pitch)
        events = [None, None]
        stream = torch.cuda.Stream(device=device)
        stream.wait_stream(torch.cuda.current_stream(device))       # `raw` may be a recycled block (see device_stack)
        fd = os.open(self.file, os.O_RDONLY)
        try:
            slot = 0
            for f0 in range(0, n, chunk_frames):
                m = min(chunk_frames, n - f0)
                if events[slot] is not None:
                    events[slot].synchronize()
                view = bufs[slot][:m * pitch]
                mv = memoryview(view.numpy())
                if avi.stride is not None:               # payload k sits at offsets[k0] + k * stride: one read, chunk headers included
                    spans = [(0, m * pitch - (pitch - avi.payload_bytes), int(avi.offsets[k0 + f0]))]
                else:
                    spans = [(i * pitch, avi.payload_bytes, int(avi.offsets[k0 + f0 + i])) for i in range(m)]
                for dst0, nbytes, pos in spans:
                    got = 0
                    while got < nbytes:
                        r = os.preadv(fd, [mv[dst0 + got:dst0 + nbytes]], pos + got)
                        if r <= 0:
                            raise Exception('error input file ' + str(self.file) + ': short read')
                        got += r
                _lib.check(_lib.lib.shg_upload_frames(raw.data_ptr() + f0 * pitch, m * pitch, view.data_ptr(), m * pitch, 1,
                                                      stream.cuda_stream), 'shg_upload_frames')
                ev = torch.cuda.Event()
                ev.record(stream)
                events[slot] = ev
                slot ^= 1
        finally:
            os.close(fd)
            for ev in events:
                if ev is not None:
                    ev.synchronize()
            _return_pinned_pair(bufs)
        h, w = int(self.Height), int(self.Width)
        stack = ops.padded_stack(n, h, w, torch.uint8, device)
        cur = torch.cuda.current_stream(device)
        cur.wait_stream(stream)
        lut = torch.from_numpy(avi.gray_lut).to(device) if avi.gray_lut is not None else None
        fstride = stack.stride(0) if n > 1 else h * w
        for f0 in range(0, n, 32768):                          # grid limit: 65535 frames per launch
            m = min(32768, n - f0)
            _lib.check(_lib.lib.shg_unpack_dib_frames(raw.data_ptr() + f0 * pitch, m, pitch, h, w, avi.bit_count, avi.row_bytes,
                                                      int(avi.bottom_up), lut.data_ptr() if lut is not None else None,
                                                      stack.data_ptr() + f0 * fstride, fstride, cur.cuda_stream),
                       'shg_unpack_dib_frames')
        return stack


class array_reader:
    """A reader over frames that are already in HBM (bench, tests, sharded generation).
    `stack` is [n, Height, Width] in file layout; frame_count is the whole scan length."""

    def __init__(self, stack, frame_count=None, frame_range=None):
        if stack.dim() != 3:
            raise ValueError('stack must be [N, Height, Width]')
        n, h, w = stack.shape
        self.file = '<device>'
        self.Width, self.Height = w, h
        self.FrameCount = n if frame_count is None else int(frame_count)
        self.frame_range = (0, n) if frame_range is None else (int(frame_range[0]), int(frame_range[1]))
        self.infilebytes = stack.element_size()
        self.infiledatatype = 'uint8' if self.infilebytes == 1 else 'uint16'
        self.count = w * h
        self.FrameIndex = -1
        if w > h:
            self.flag_rotate, self.ih, self.iw = True, w, h
        else:
            self.flag_rotate, self.iw, self.ih = False, w, h
        self._stack = stack

    def device_stack(self, device=None):
        return self._stack

    def has_frames(self):
        return self.FrameIndex + 1 < self._stack.shape[0]

    def next_frame(self):
        self.FrameIndex += 1
        img = self._stack[self.FrameIndex].cpu().numpy()
        if self.flag_rotate:
            img = np.rot90(img)
        if self.infilebytes == 1:
            img = np.asarray(img, dtype='uint16') * 256
        return img

    def reset(self):
        self.FrameIndex = -1


class all_video_reader:
    """Everything in host RAM, rotated, with per-frame means (reference video_reader.py:129-158;
    used by the spectral analyser).  device_stack() uploads the rotated frames: they are
    [N, ih, iw] with ih >= iw, which is a valid un-rotated file layout."""

    def __init__(self, file, buffer_size=25):
        vid_rdr = video_reader(file, buffer_size)
        self.file = file
        for name in ('ih', 'iw', 'Width', 'Height', 'FrameCount', 'count'):
            setattr(self, name, getattr(vid_rdr, name))
        self.FrameIndex = -1
        self.frames = np.zeros((int(self.FrameCount), int(self.ih), int(self.iw)), dtype=np.uint16)
        self.means = np.zeros(int(self.FrameCount))
        i = 0
        while vid_rdr.has_frames():
            frame = vid_rdr.next_frame()
            self.means[i] = np.mean(frame)
            self.frames[i, :, :] = frame
            i += 1
        self.frame_range = (0, int(self.FrameCount))
        self.infilebytes = 2
        self._stack = None

    def has_frames(self):
        return self.FrameIndex + 1 < self.FrameCount

    def next_frame(self):
        self.FrameIndex += 1
        return self.frames[self.FrameIndex, :, :]

    def reset(self):
        self.FrameIndex = -1

    def device_stack(self, device=None):
        if self._stack is None:
            self._stack = torch.from_numpy(self.frames).to(device or default_device())
        return self._stack
