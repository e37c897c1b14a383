"""Synthetic SER scans (the reference ships no sample data).

The recipe follows SURVEY.md section 8(d): a bright, limb-darkened elliptical
solar disk scanned across a slit whose spectrum holds one curved Gaussian
absorption line.  The content satisfies the hidden preconditions of the
reference pipeline (sunlit span > 100 rows so that the blur height is non-zero,
solex_util.py:229-230; noisy sharp minima so that np.unique has >= 3 values,
solex_util.py:245-246; a closed limb for the ellipse fit, ellipse_to_circle.py:
245-263; no zero pixel inside the disk, solex_util.py:393; the line >= 13 px
from both spectral edges, solex_util.py:231).

Two generators are provided: a NumPy one (small inputs, files for tests and the
CLI) and a torch one that builds the frame stack directly in HBM for bench.py.
Both produce frames in *file layout* [N, Height, Width]; when Width > Height the
reference rotates each frame (video_reader.py:119-120), so the slit axis is the
file's column axis, reversed.
"""
import struct

import numpy as np

SER_HEADER_BYTES = 178  # video_reader.py:65


def curve_of_row(y, ih, iw, tilt=0.002, curv=6e-6):
    """Position of the absorption line (in wavelength pixels) for slit row y."""
    yc = y - ih / 2.0
    return iw / 2.0 + curv * yc * yc + tilt * yc


def scene_params(n_frames, ih, iw):
    return dict(cx=n_frames / 2.0, cy=ih / 2.0, ax=0.42 * n_frames, ay=0.44 * ih,
                y_lo=0.06 * ih, y_hi=0.94 * ih, depth=0.8, sigma=3.0,
                gain=0.8, sky=0.02, noise=0.004)


def synth_frames_numpy(n_frames, width, height, depth_bits=16, seed=0, row_gain=None,
                       k0=0, k1=None, n_total=None, tilt=0.002, curv=6e-6, scene=None):
    """Return frames [k1-k0, Height, Width] (file layout) as uint8/uint16.

    scene overrides entries of scene_params() (disk centre / semi-axes, lit slit span, line depth / width, gain,
    sky, noise) for scans that are off-centre, elongated, noisier ...; scene['spots'] is a list of
    (frame, row, radius_frames, radius_rows, depth) Gaussian features: depth > 0 darkens (sunspots), < 0 brightens
    (plages, or prominences when placed off the limb).

    n_total is the length of the whole scan (defaults to n_frames); k0:k1 selects
    a block of frames of that scan (used for sharded generation).  The noise of
    frame k only depends on (seed, k), so shards agree with the whole.
    """
    n_total = n_frames if n_total is None else n_total
    k1 = n_total if k1 is None else k1
    rotate = width > height
    ih, iw = (width, height) if rotate else (height, width)
    sp = dict(scene_params(n_total, ih, iw), **(scene or {}))
    full = 255.0 if depth_bits == 8 else 65535.0
    y = np.arange(ih, dtype=np.float64)
    x = np.arange(iw, dtype=np.float64)
    line = 1.0 - sp['depth'] * np.exp(-0.5 * ((x[None, :] - curve_of_row(y, ih, iw, tilt, curv)[:, None]) / sp['sigma']) ** 2)
    lit = ((y > sp['y_lo']) & (y < sp['y_hi'])).astype(np.float64)
    if row_gain is not None:
        lit = lit * np.asarray(row_gain, dtype=np.float64)
    out = np.empty((k1 - k0, height, width), dtype=np.uint8 if depth_bits == 8 else np.uint16)
    for k in range(k0, k1):
        r2 = ((k - sp['cx']) / sp['ax']) ** 2 + ((y - sp['cy']) / sp['ay']) ** 2
        bright = np.where(r2 < 1.0, 0.35 + 0.65 * np.sqrt(np.clip(1.0 - r2, 0.0, 1.0)), sp['sky']) * lit
        for (sk, sy, rk, ry, depth) in sp.get('spots', ()):
            bright = bright * (1.0 - depth * np.exp(-0.5 * (((k - sk) / rk) ** 2 + ((y - sy) / ry) ** 2)))
        rng = np.random.default_rng([seed, k])
        img = sp['gain'] * bright[:, None] * line + sp['noise'] * rng.standard_normal((ih, iw))
        img = np.clip(np.rint(img * full), 0, full).astype(out.dtype)
        # inverse of np.rot90 (video_reader.py:119-120): img[y, x] = raw[x, W-1-y]
        out[k - k0] = np.rot90(img, -1) if rotate else img
    return out


def ser_header(width, height, depth_bits, n_frames):
    """178-byte SER header with the fields the reference parses (video_reader.py:31-66)."""
    hdr = bytearray(SER_HEADER_BYTES)
    hdr[0:14] = b'LUCAM-RECORDER'
    struct.pack_into('<7i', hdr, 14, 0, 0, 1, width, height, depth_bits, n_frames)
    return bytes(hdr)


def write_ser(path, frames, depth_bits=None):
    """Write frames [N, Height, Width] (uint8/uint16) as a little-endian SER file."""
    frames = np.ascontiguousarray(frames)
    if depth_bits is None:
        depth_bits = 8 if frames.dtype == np.uint8 else 16
    n, h, w = frames.shape
    with open(path, 'wb') as f:
        f.write(ser_header(w, h, depth_bits, n))
        f.write(frames.astype('<u2' if depth_bits == 16 else np.uint8, copy=False).tobytes())
    return path


def write_avi(path, frames, layout='Y800', bottom_up=None, palette=None, audio_every=0, junk_bytes=0, rec_lists=False):
    """Write frames as an uncompressed AVI 1.0 file (RIFF 'AVI ', hdrl / movi / idx1) -- test input for the AVI reader.
    layout 'Y800': frames uint8 [N, H, W], 8-bit luma, top-down, no row padding;
           'pal8': frames uint8 [N, H, W] of palette indices, BI_RGB 8 bit, rows padded to 4 bytes,
                   bottom-up unless bottom_up=False (negative biHeight); palette uint8 [256, 3] B,G,R (default grey ramp);
           'bgr24': frames uint8 [N, H, W, 3] (B, G, R), BI_RGB 24 bit, same row rules.
    audio_every > 0 interleaves a dummy '01wb' chunk after every that many frames (uneven chunk spacing),
    junk_bytes > 0 puts a JUNK chunk before 'movi', rec_lists wraps every frame in a LIST 'rec '."""
    import struct
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, h, w = frames.shape[:3]
    raw_rgb = layout in ('pal8', 'bgr24')
    bits = 24 if layout == 'bgr24' else 8
    if bottom_up is None:
        bottom_up = raw_rgb
    row = w if layout == 'Y800' else (w * bits + 31) // 32 * 4
    payload = row * h

    def chunk(cid, data):
        return cid + struct.pack('<I', len(data)) + data + (b'\0' if len(data) & 1 else b'')

    def lst(ltype, body):
        return b'LIST' + struct.pack('<I', 4 + len(body)) + ltype + body

    pal = b''
    if layout == 'pal8':
        p = np.stack([np.arange(256)] * 3, axis=1).astype(np.uint8) if palette is None else np.asarray(palette, dtype=np.uint8)
        pal = np.concatenate([p, np.zeros((256, 1), np.uint8)], axis=1).tobytes()
    fourcc = b'\0\0\0\0' if raw_rgb else b'Y800'
    bih = struct.pack('<IiiHH4sIiiII', 40, w, h if (bottom_up or not raw_rgb) else -h, 1, bits, fourcc, payload, 0, 0,
                      256 if layout == 'pal8' else 0, 0) + pal
    strh = struct.pack('<4s4sIHHIIIIIIII4h', b'vids', b'DIB ' if raw_rgb else b'Y800', 0, 0, 0, 0, 1, 25, 0, n, payload, 0xffffffff, 0,
                       0, 0, w, h)
    avih = struct.pack('<14I', 40000, payload * 25, 0, 0x10, n, 0, 2 if audio_every else 1, payload, w, h, 0, 0, 0, 0)
    streams = lst(b'strl', chunk(b'strh', strh) + chunk(b'strf', bih))
    if audio_every:
        wave = struct.pack('<HHIIHH', 1, 1, 8000, 8000, 1, 8)
        strh_a = struct.pack('<4s4sIHHIIIIIIII4h', b'auds', b'\0\0\0\0', 0, 0, 0, 0, 1, 8000, 0, 0, 0, 0xffffffff, 1, 0, 0, 0, 0)
        streams += lst(b'strl', chunk(b'strh', strh_a) + chunk(b'strf', wave))
    hdrl = lst(b'hdrl', chunk(b'avih', avih) + streams)
    pre = hdrl + (chunk(b'JUNK', b'\0' * junk_bytes) if junk_bytes else b'')
    movi = bytearray()
    index = bytearray()
    for k in range(n):
        img = frames[k].reshape(h, -1)
        if bottom_up:
            img = img[::-1]
        rows = np.zeros((h, row), dtype=np.uint8)
        rows[:, :img.shape[1]] = img
        body = chunk(b'00db', rows.tobytes())
        if rec_lists:
            body = lst(b'rec ', body)
            index += struct.pack('<4sIII', b'00db', 0x10, 4 + len(movi) + 12, payload)
        else:
            index += struct.pack('<4sIII', b'00db', 0x10, 4 + len(movi), payload)
        movi += body
        if audio_every and (k + 1) % audio_every == 0:
            index += struct.pack('<4sIII', b'01wb', 0x10, 4 + len(movi), 40)
            movi += chunk(b'01wb', bytes(40))
    body = pre + lst(b'movi', bytes(movi)) + chunk(b'idx1', bytes(index))
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', 4 + len(body)) + b'AVI ' + body)
    return path


def synth_frames_torch(n_frames, width, height, depth_bits=16, seed=0, device='cuda',
                       k0=0, k1=None, n_total=None, chunk=250, padded=False):
    """Same scene built directly on `device` (bench workloads; torch RNG, so not
    bit-identical to the NumPy generator).  Returns uint8/uint16 [k1-k0, H, W]."""
    import torch
    n_total = n_frames if n_total is None else n_total
    k1 = n_total if k1 is None else k1
    rotate = width > height
    ih, iw = (width, height) if rotate else (height, width)
    sp = scene_params(n_total, ih, iw)
    full = 255.0 if depth_bits == 8 else 65535.0
    dt = torch.uint8 if depth_bits == 8 else torch.uint16
    y = torch.arange(ih, dtype=torch.float32, device=device)
    x = torch.arange(iw, dtype=torch.float32, device=device)
    yc = y - ih / 2.0
    curve = iw / 2.0 + 6e-6 * yc * yc + 0.002 * yc
    line = 1.0 - sp['depth'] * torch.exp(-0.5 * ((x[None, :] - curve[:, None]) / sp['sigma']) ** 2)
    lit = ((y > sp['y_lo']) & (y < sp['y_hi'])).float()
    if padded:          # the frame pitch video_reader.device_stack() uploads into (rounded up to 8 KiB)
        from . import ops
        out = ops.padded_stack(k1 - k0, height, width, dt, device)
    else:
        out = torch.empty((k1 - k0, height, width), dtype=dt, device=device)
    gen = torch.Generator(device=device)
    for c0 in range(k0, k1, chunk):
        c1 = min(k1, c0 + chunk)
        gen.manual_seed(seed * 1000003 + c0)
        k = torch.arange(c0, c1, dtype=torch.float32, device=device)
        r2 = ((k[:, None] - sp['cx']) / sp['ax']) ** 2 + ((y[None, :] - sp['cy']) / sp['ay']) ** 2
        bright = torch.where(r2 < 1.0, 0.35 + 0.65 * torch.sqrt(torch.clamp(1.0 - r2, min=0.0)),
                             torch.full_like(r2, sp['sky'])) * lit[None, :]
        img = sp['gain'] * bright[:, :, None] * line[None, :, :]
        img = img + sp['noise'] * torch.randn(img.shape, device=device, generator=gen)
        img = torch.clamp(torch.round(img * full), 0, full)
        if rotate:
            img = torch.rot90(img, -1, dims=(1, 2))
        out.view(torch.int16 if depth_bits == 16 else torch.uint8)[c0 - k0:c1 - k0] = img.to(torch.int32).to(
            torch.int16 if depth_bits == 16 else torch.uint8)
    return out
