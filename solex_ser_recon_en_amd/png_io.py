"""16-bit / 8-bit grayscale PNG encoder and decoder (no third-party codec).

The reference writes its products with cv2.imwrite(..., [IMWRITE_PNG_COMPRESSION, 0])
(solex_util.py:556-566): 16-bit gray, uncompressed deflate.  Only the decoded pixels are
part of the output contract; the byte stream here is a valid PNG with stored (level 0)
deflate blocks, filter 0 on every row.
"""
import struct
import zlib

import numpy as np

_SIG = b'\x89PNG\r\n\x1a\n'


def _chunk(tag, payload):
    return struct.pack('>I', len(payload)) + tag + payload + struct.pack('>I', zlib.crc32(tag + payload) & 0xffffffff)


def _pieces(img, compression=0):
    """The PNG byte stream as a list of buffers (written one after the other: an 8 MB image is copied twice, for the
    big-endian rows and by deflate, instead of once per concatenation)."""
    img = np.asarray(img)
    if img.ndim != 2 or img.dtype not in (np.uint8, np.uint16):
        raise TypeError('png_bytes writes 2-D uint8/uint16 images, got %s %s' % (img.dtype, img.shape))
    h, w = img.shape
    depth = 8 * img.dtype.itemsize
    rows = np.empty((h, 1 + w * img.dtype.itemsize), dtype=np.uint8)          # filter byte 0 + big-endian samples
    rows[:, 0] = 0
    src = np.ascontiguousarray(img)
    rows[:, 1:] = (src.byteswap() if depth == 16 else src).view(np.uint8).reshape(h, -1)
    ihdr = struct.pack('>IIBBBBB', w, h, depth, 0, 0, 0, 0)
    idat = zlib.compress(rows, compression)
    crc = zlib.crc32(idat, zlib.crc32(b'IDAT')) & 0xffffffff
    return [_SIG, _chunk(b'IHDR', ihdr), struct.pack('>I', len(idat)), b'IDAT', idat, struct.pack('>I', crc), _chunk(b'IEND', b'')]


def png_bytes(img, compression=0):
    return b''.join(_pieces(img, compression))


def write_png(path, img, compression=0):
    with open(path, 'wb') as f:
        for piece in _pieces(img, compression):
            f.write(piece)


def read_png_gray(path):
    """Decode an 8/16-bit grayscale (or RGB -> gray is NOT done: colour types raise) non-interlaced PNG."""
    raw = open(path, 'rb').read()
    if raw[:8] != _SIG:
        raise ValueError('%s is not a PNG file' % path)
    pos, idat, ihdr = 8, [], None
    while pos < len(raw):
        n, tag = struct.unpack('>I4s', raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b'IHDR':
            ihdr = struct.unpack('>IIBBBBB', body)
        elif tag == b'IDAT':
            idat.append(body)
        elif tag == b'IEND':
            break
    w, h, depth, ctype, _, _, interlace = ihdr
    if ctype != 0 or interlace != 0 or depth not in (8, 16):
        raise ValueError('only non-interlaced 8/16-bit grayscale PNGs are supported')
    bpp = depth // 8
    data = np.frombuffer(zlib.decompress(b''.join(idat)), dtype=np.uint8).reshape(h, 1 + w * bpp)
    out = np.zeros((h, w * bpp), dtype=np.uint8)
    prev = np.zeros(w * bpp, dtype=np.int32)
    for y in range(h):
        ft, line = int(data[y, 0]), data[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:   # Sub / Average / Paeth need a serial scan along the row
            cur = np.zeros_like(line)
            for i in range(w * bpp):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) // 2
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    if depth == 16:
        return out.view('>u2').astype(np.uint16).reshape(h, w)
    return out.reshape(h, w)
