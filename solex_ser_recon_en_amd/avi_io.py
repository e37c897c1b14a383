"""Uncompressed AVI input: where the pixels of every video frame are in the file.

The reference opens AVI scans with cv2.VideoCapture and converts every decoded BGR frame with
cv2.COLOR_BGR2GRAY (video_reader.py:68-80, 111-113).  Compressed streams need a codec and stay out
of scope; what capture programs write for a monochrome camera is uncompressed, and that needs no
codec at all -- only the RIFF chunk walk below:

  * 'Y800' / 'Y8  ' / 'GREY'  : 8-bit luma, rows top-down, no row padding;
  * BI_RGB 8 bit              : palette indices, rows bottom-up (top-down when biHeight < 0),
                                rows padded to 4 bytes; grey = BGR2GRAY(palette[index]);
  * BI_RGB 24 bit             : B, G, R bytes, same row order and padding.

BGR2GRAY is OpenCV 4's 8-bit fixed-point form, (B*3735 + G*19235 + R*9798 + 2^14) >> 15; it is the
identity on grey pixels (B = G = R), which is the SHG case.  opencv-python is absent here and unpinned
upstream, so this reader is restated from the AVI RIFF layout, not pinned against cv2.
FrameCount is the number of non-empty video chunks in the file.
"""
import os
import struct

import numpy as np

GRAY_FOURCC = (b'Y800', b'Y8  ', b'GREY')
BY15, GY15, RY15, GRAY_SHIFT = 3735, 19235, 9798, 15


def bgr_to_gray_u8(b, g, r):
    """cv2.COLOR_BGR2GRAY for uint8 (OpenCV 4.x color_rgb: 15-bit coefficients, rounded)."""
    acc = b.astype(np.uint32) * BY15 + g.astype(np.uint32) * GY15 + r.astype(np.uint32) * RY15 + (1 << (GRAY_SHIFT - 1))
    return (acc >> GRAY_SHIFT).astype(np.uint8)


class AviIndex:
    """Geometry of the first video stream and the file offset of every frame's pixel data."""

    def __init__(self, path):
        self.path = path
        self.width = self.height = self.bit_count = 0
        self.bottom_up = False
        self.fourcc = b''
        self.gray_lut = None            # uint8[256] for palettised frames (None: identity)
        self.offsets = []
        self._video_stream = None
        self._streams_seen = 0
        self._strf_pending = False
        size = os.path.getsize(path)
        with open(path, 'rb') as f:
            pos = 0
            while pos + 12 <= size:
                f.seek(pos)
                tag, riff_size, form = struct.unpack('<4sI4s', f.read(12))
                if tag != b'RIFF' or form not in (b'AVI ', b'AVIX'):
                    if pos == 0:
                        raise Exception('error input file ' + str(path) + ': not a RIFF AVI file')
                    break
                end = min(pos + 8 + riff_size, size)
                self._walk(f, pos + 12, end, in_movi=False)
                pos = end + (riff_size & 1)
        if self._video_stream is None or self.width <= 0 or self.height <= 0:
            raise Exception('error input file ' + str(path) + ': no video stream in the AVI headers')
        self.row_bytes = self.width if self.fourcc in GRAY_FOURCC else (self.width * self.bit_count + 31) // 32 * 4
        self.payload_bytes = self.row_bytes * self.height
        sizes = np.array([s for _, s in self.offsets], dtype=np.int64)
        self.offsets = np.array([o for o, _ in self.offsets], dtype=np.int64)
        if len(self.offsets) == 0:
            raise Exception('error input file ' + str(path) + ': no video frames')
        if np.any(sizes < self.payload_bytes) or self.offsets[-1] + self.payload_bytes > size:
            raise Exception('error input file ' + str(path) + ': a frame chunk is smaller than %d x %d x %d bits'
                            % (self.width, self.height, self.bit_count))
        self.frame_count = len(self.offsets)
        step = np.diff(self.offsets)
        # capture programs write one chunk per frame back to back: then a block of frames is one contiguous read
        self.stride = int(step[0]) if len(step) and np.all(step == step[0]) and step[0] >= self.payload_bytes else None

    def _walk(self, f, start, end, in_movi):
        p = start
        while p + 8 <= end:
            f.seek(p)
            cid, csz = struct.unpack('<4sI', f.read(8))
            body = p + 8
            if cid == b'LIST':
                ltype = f.read(4)
                if ltype in (b'hdrl', b'strl', b'movi', b'rec '):
                    self._walk(f, body + 4, min(body + csz, end), in_movi or ltype == b'movi')
            elif cid == b'strh':
                fcc_type, handler = struct.unpack('<4s4s', f.read(8))
                if fcc_type == b'vids' and self._video_stream is None:
                    self._video_stream = self._streams_seen
                    self._strf_pending = True
                self._streams_seen += 1
            elif cid == b'strf' and self._strf_pending:
                self._strf_pending = False
                self._parse_bitmapinfo(f.read(csz))
            elif in_movi and self._video_stream is not None and cid[2:4] in (b'db', b'dc') and cid[:2].isdigit() \
                    and int(cid[:2]) == self._video_stream and csz > 0:
                self.offsets.append((body, csz))
            p = body + csz + (csz & 1)

    def _parse_bitmapinfo(self, data):
        if len(data) < 40:
            raise Exception('error input file ' + str(self.path) + ': truncated BITMAPINFOHEADER')
        bi_size, w, h, planes, bits, compression, size_image, _, _, clr_used, _ = struct.unpack('<IiiHH4sIiiII', data[:40])
        self.width, self.height, self.bit_count, self.fourcc = int(w), abs(int(h)), int(bits), compression
        raw_rgb = compression == b'\x00\x00\x00\x00'
        if not ((raw_rgb and bits in (8, 24)) or (compression in GRAY_FOURCC and bits == 8)):
            raise Exception('error input file %s: AVI stream %r / %d bit needs a video codec; only uncompressed 8-bit grey '
                            '(Y800, 8-bit DIB) and 24-bit DIB frames are supported' % (self.path, compression, bits))
        self.bottom_up = raw_rgb and h > 0
        if raw_rgb and bits == 8:
            n = clr_used or 256
            pal = np.frombuffer(data, dtype=np.uint8, count=min(n, (len(data) - bi_size) // 4) * 4, offset=bi_size).reshape(-1, 4)
            if len(pal):
                lut = np.arange(256, dtype=np.uint8)
                lut[:len(pal)] = bgr_to_gray_u8(pal[:, 0], pal[:, 1], pal[:, 2])
                self.gray_lut = None if np.array_equal(lut, np.arange(256, dtype=np.uint8)) else lut

    # ---- host decode of one frame (the compatibility iterator) ----------------------------------------
    def frame(self, k):
        raw = np.fromfile(self.path, dtype=np.uint8, count=self.payload_bytes, offset=int(self.offsets[k]))
        rows = raw.reshape(self.height, self.row_bytes)
        if self.bottom_up:
            rows = rows[::-1]
        if self.bit_count == 24:
            px = rows[:, :self.width * 3].reshape(self.height, self.width, 3)
            return bgr_to_gray_u8(px[:, :, 0], px[:, :, 1], px[:, :, 2])
        img = rows[:, :self.width]
        return self.gray_lut[img] if self.gray_lut is not None else np.ascontiguousarray(img)
