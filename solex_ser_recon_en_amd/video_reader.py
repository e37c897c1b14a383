"""SER decode: header parse, host frame iterator, and the HBM upload.

Mirrors the reference's video_reader surface (video_reader.py:10-126): attributes
Width Height FrameCount FrameIndex count infilebytes infiledatatype flag_rotate ih iw
and methods next_frame() / has_frames().  New here: device_stack(), which streams the
file through pinned host buffers into one [N, Height, Width] tensor in HBM on a copy
stream (double-buffered, asynchronous hipMemcpy) -- the file is read ONCE and both
frame passes run on the resident stack (the reference decodes it twice,
Solex_recon.py:61-63).  The rotation of wide frames (video_reader.py:119-120) and the
8-bit x256 widening (:121-122) are never materialised for the stack: the kernels index
the file layout directly.  AVI input needs a video codec and is out of scope.
"""
import numpy as np
import torch

from .device import default_device

SER_HEADER_BYTES = 178


class video_reader:
    def __init__(self, file, buffer_size=25, frame_range=None):
        self.file = file
        self.buffer_size = buffer_size
        upper = str(file).upper()
        if upper.endswith('.AVI'):
            raise Exception('error input file ' + file + ': AVI input is not supported by the MI355X path (SER only)')
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
        if self.Width > self.Height:                        # video_reader.py:84-91
            self.flag_rotate, self.ih, self.iw = True, self.Width, self.Height
        else:
            self.flag_rotate, self.iw, self.ih = False, self.Width, self.Height
        # frame block owned by this process (multi-GPU sharding); FrameCount stays the scan length
        self.frame_range = (0, int(self.FrameCount)) if frame_range is None else (int(frame_range[0]), int(frame_range[1]))
        self._mm = None
        self._stack = None

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
        img = np.array(self._memmap()[self.FrameIndex])
        if self.flag_rotate:
            img = np.rot90(img)
        if self.infilebytes == 1:
            img = np.asarray(img, dtype='uint16') * 256
        return img

    def reset(self):
        self.FrameIndex = -1

    # ---- decode into HBM -----------------------------------------------------
    def device_stack(self, device=None, chunk_frames=None):
        """Frames [k0:k1) of the file as one tensor [n, Height, Width] in HBM (file layout)."""
        if self._stack is not None:
            return self._stack
        device = device or default_device()
        k0, k1 = self.frame_range
        n = k1 - k0
        h, w, b = int(self.Height), int(self.Width), self.infilebytes
        if n <= 0 or h <= 0 or w <= 0:
            raise Exception('error input file ' + str(self.file) + ': no frames')
        frame_bytes = h * w * b
        import os
        if os.path.getsize(self.file) < SER_HEADER_BYTES + int(self.FrameCount) * frame_bytes:
            raise Exception('error input file ' + str(self.file) + ': shorter than its header says')
        dt = torch.uint16 if b == 2 else torch.uint8
        stack = torch.empty((n, h, w), dtype=dt, device=device)
        flat = stack.view(-1).view(torch.uint8)
        if chunk_frames is None:
            chunk_frames = max(1, (64 << 20) // frame_bytes)                  # ~64 MiB per pinned buffer
        pinned = [torch.empty(chunk_frames * frame_bytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        events = [None, None]
        copy_stream = torch.cuda.Stream(device=device)
        with open(self.file, 'rb', buffering=0) as f:
            f.seek(SER_HEADER_BYTES + k0 * frame_bytes)
            done, slot = 0, 0
            while done < n:
                m = min(chunk_frames, n - done)
                if events[slot] is not None:
                    events[slot].synchronize()                                   # the buffer's previous copy has landed
                view = pinned[slot][:m * frame_bytes]
                got = f.readinto(memoryview(view.numpy()))
                if got != m * frame_bytes:
                    raise Exception('error input file ' + str(self.file) + ': short read')
                with torch.cuda.stream(copy_stream):
                    flat[done * frame_bytes:(done + m) * frame_bytes].copy_(view, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                events[slot] = ev
                done += m
                slot ^= 1
        torch.cuda.current_stream(device).wait_stream(copy_stream)
        self._stack = stack
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
