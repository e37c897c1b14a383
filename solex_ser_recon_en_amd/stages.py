"""Stage composites: one C call per stage function of the reference (include/shg_hip.h, csrc/stages.hip).

A composite launches the stage's kernels, reads the few values the control plane needs through pinned memory and
runs that control plane in C++, all without the interpreter -- ctypes drops the interpreter lock for the call, so
the scan workers of Solex_recon.solex_do_work overlap for real.  This module owns what a call needs besides its
arguments: a per-thread device workspace and pinned staging areas (a scan worker is a thread with its own stream,
so buffers are never shared between streams), grown on demand and reused from file to file.
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib, ops
from ._lib import lib

_local = threading.local()


def use_buffers(store):
    """Bind this thread to a buffer store (a dict): a scan worker's buffers outlive its thread, so the next batch's worker
    with the same index finds them again instead of pinning fresh memory."""
    _local.buffers = store


def _scratch(name, nbytes, device=None, pinned=False):
    """A per-thread buffer of at least nbytes: device memory (torch's allocator) or pinned host memory."""
    store = _local.__dict__.setdefault('buffers', {})
    key = (name, str(device), pinned)
    buf = store.get(key)
    if buf is None or buf.numel() < nbytes:
        size = max(int(nbytes), 1)
        if pinned:
            buf = torch.empty(size, dtype=torch.uint8).pin_memory()
        else:
            buf = torch.empty(size, dtype=torch.uint8, device=device)
        store[key] = buf
    return buf


_size_cache = {}


def _sizes(tag, fn, *args):
    """A *_bytes query, remembered per argument tuple (every scan of a batch asks the same question)."""
    key = (tag,) + args
    v = _size_cache.get(key)
    if v is None:
        v = _size_cache[key] = fn(*args)
    return v


def _p(a):
    return a.ctypes.data          # (an int is accepted for a void* argument; data_as builds a ctypes object: twice the time)


# ---- compute_mean_return_fit ----------------------------------------------------------------------------
def mean_fit(stack, n_total, sums=None, geometry=None, want_plot_data=False):
    """-> dict(mean, max: uint16 GPU tensors [ih, iw]; y1, y2; p [4] lowest power first; fit [ih, 4];
    sharp [ih] int32 and mask_good [y2-y1] bool when want_plot_data).
    sums = (total int64 [H*W], max uint16 [H*W]) replaces pass A (the all-reduced partial sums of a sharded scan)."""
    if stack is not None:
        n, h, w, bpp = ops.stack_geometry(stack)
        dev = stack.device
        fstride = ops.frame_stride(stack)
        stack_ptr = stack.data_ptr()
    else:
        n, h, w, bpp = geometry
        dev = sums[0].device
        fstride, stack_ptr = 0, None
    ih, iw = (w, h) if w > h else (h, w)
    images = torch.empty((2, ih, iw), dtype=torch.uint16, device=dev)
    need = _sizes('mean_ws', lib.shg_stage_mean_fit_workspace_bytes, n if sums is None else 0, h, w, bpp)
    ws = _scratch('mean_fit', need, dev)
    need_pin = _sizes('mean_pin', lib.shg_stage_mean_fit_host_bytes, h, w)
    pin = _scratch('mean_fit', need_pin, pinned=True)
    y12 = np.zeros(2, dtype=np.int64)
    p4 = np.empty(4)
    fit = np.empty((ih, 4))
    sharp = np.empty(ih, dtype=np.int32) if want_plot_data else None
    mask = np.zeros(ih, dtype=np.uint8) if want_plot_data else None
    sum_ptr, max_ptr = (sums[0].data_ptr(), sums[1].data_ptr()) if sums is not None else (None, None)
    _lib.check(lib.shg_stage_mean_fit(stack_ptr, n, h, w, bpp, fstride, sum_ptr, max_ptr, int(n_total), images[0].data_ptr(),
                                      images[1].data_ptr(), _p(y12), _p(p4), _p(fit), None if sharp is None else _p(sharp),
                                      None if mask is None else _p(mask), ws.data_ptr(), ws.numel(), pin.data_ptr(), pin.numel(),
                                      ops._stream()), 'shg_stage_mean_fit')
    y1, y2 = int(y12[0]), int(y12[1])
    out = {'mean': images[0], 'max': images[1], 'y1': y1, 'y2': y2, 'p': p4, 'fit': fit}
    if want_plot_data:
        out['sharp'] = sharp
        out['mask_good'] = mask[:max(y2 - y1, 0)].astype(bool)
    return out


# ---- read_video_improved ----------------------------------------------------------------------------------
def extract(stack, fit, shifts, n_cols=None, k_offset=0, flip_x=False, out=None, want_minmax=False):
    """-> uint16 GPU tensor [S, ih, n_cols] (rows padded to 64 elements); with want_minmax also the per-plane extrema
    int32 [S, 2] = {min, max} for ops.warp_rows_u16 (only meaningful when the call covers the whole scan)."""
    n, h, w, bpp = ops.stack_geometry(stack)
    dev = stack.device
    ih = max(h, w)
    fit = np.ascontiguousarray(fit, dtype=np.float64)
    if fit.shape != (ih, 4):
        raise ValueError('fit must be [%d, 4]' % ih)
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    s = int(sh.size)
    n_cols = n if n_cols is None else int(n_cols)
    if out is None:
        pitch = (n_cols + 63) // 64 * 64
        alloc = torch.empty if (n_cols == n and int(k_offset) == 0) else torch.zeros
        out = alloc((s, ih, pitch), dtype=torch.uint16, device=dev)[:, :, :n_cols]
    if out.shape != (s, ih, n_cols) or out.stride(2) != 1:
        raise ValueError('out must be a [S, ih, n_cols] view with unit column stride')
    need = _sizes('extract', lib.shg_stage_extract_workspace_bytes, h, w, s)
    ws = _scratch('extract', need, dev)
    pin = _scratch('extract', need, pinned=True)
    mm_store = torch.empty(s * 130, dtype=torch.int32, device=dev) if want_minmax else None      # 64 slots x 2 per plane, then {min, max}
    _lib.check(lib.shg_stage_extract(stack.data_ptr(), n, h, w, bpp, ops.frame_stride(stack), _p(fit), _p(sh), s, out.data_ptr(),
                                     out.stride(1), out.stride(0), n_cols, int(k_offset), int(bool(flip_x)),
                                     None if mm_store is None else mm_store.data_ptr(), ws.data_ptr(), ws.numel(),
                                     pin.data_ptr(), pin.numel(), ops._stream()), 'shg_stage_extract')
    return (out, mm_store[s * 128:].view(s, 2)) if want_minmax else out


# ---- ellipse_to_circle: the limb fit -------------------------------------------------------------------------
def _canny_ladder_taps():
    """scipy.ndimage's Gaussian taps for canny's retry ladder sigma = 2, 1.5, 1, 0.5 (ellipse_to_circle.py:245-256)."""
    taps = getattr(_canny_ladder_taps, 'cache', None)
    if taps is None:
        taps = np.ascontiguousarray(np.concatenate([ops.gaussian_taps(s)[0] for s in (2.0, 1.5, 1.0, 0.5)]), dtype=np.float64)
        _canny_ladder_taps.cache = taps
    return taps


def _limb_call(fn, disk, extra):
    ptr, h, w, pitch = ops._img(disk, 'disk', torch.uint16)
    dev = disk.device
    ws = _scratch('limb', _sizes('limb_ws', lib.shg_stage_limb_points_workspace_bytes, h, w), dev)
    pin = _scratch('limb', _sizes('limb_pin', lib.shg_stage_limb_points_host_bytes, h, w), pinned=True)
    full = (-(-h // 4)) * (-(-w // 4))
    cap = min(full, 1 << 16)
    taps = _canny_ladder_taps()
    while True:
        points = np.empty((cap, 2), dtype=np.int32)
        flags = np.empty(cap, dtype=np.uint8)
        counts = np.zeros(3, dtype=np.int64)
        status = fn(ptr, h, w, pitch, _p(taps), _p(points), _p(flags), cap, _p(counts), *extra, ws.data_ptr(), ws.numel(),
                    pin.data_ptr(), pin.numel(), ops._stream())
        if status == -2 and cap < full:       # more edge pixels than a limb ever has: full-size arrays
            cap = full
            continue
        _lib.check(status, 'shg_stage_limb_fit')
        return points, flags, counts


def limb_points(disk):
    """disk: uint16 GPU image [h, w].  -> (X float [n, 2] limb points (row, col), raw_X int [m, 2] all canny points), both
    in disk pixels (down-scaled by 4, then upscaled back, ellipse_to_circle.py:299-302): get_edge_list."""
    points, flags, counts = _limb_call(lib.shg_stage_limb_points, disk, ())
    m = int(counts[0])
    raw = points[:m].astype(np.int64)
    return np.array(raw[flags[:m] != 0], dtype='float') * 4, raw * 4


def limb_fit(disk, want_points=False):
    """ellipse_to_circle without its warp, one call (shg_stage_limb_fit).  -> dict(center (x, y), height, phi, ratio,
    circle (cx, cy, r) of the corrected image, borders [4], h00, h01, h02, theta, out_h, out_w; with want_points also
    raw_X int [m, 2], X_f float [k, 2] (row, col) in disk pixels and outline [100, 2])."""
    geom = np.empty(16)
    dims = np.zeros(2, dtype=np.int64)
    outline = np.empty((100, 2)) if want_points else None
    points, flags, counts = _limb_call(lib.shg_stage_limb_fit, disk, (_p(geom), _p(dims), None if outline is None else _p(outline)))
    out = {'center': (float(geom[0]), float(geom[1])), 'height': float(geom[2]), 'phi': float(geom[3]), 'ratio': float(geom[4]),
           'circle': (float(geom[5]), float(geom[6]), float(geom[7])), 'borders': [float(v) for v in geom[8:12]],
           'h00': float(geom[12]), 'h01': float(geom[13]), 'h02': float(geom[14]), 'theta': float(geom[15]),
           'out_h': int(dims[0]), 'out_w': int(dims[1])}
    if want_points:
        m = int(counts[0])
        pts, fl = points[:m].astype(np.int64) * 4, flags[:m]
        out['raw_X'] = pts
        out['X_f'] = pts[(fl & 2) != 0].astype(float)
        out['outline'] = outline
    return out


# ---- single_image_process ---------------------------------------------------------------------------------------
def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def process_frames(frames, transversalium=None, crop=None, disc=None, clip_limit=0.8, tiles=2, keep_detrans=False):
    """frames: k uint16 GPU images of one shape and row pitch.
    transversalium: None or dict(circle (cx, cy, r), borders [4], taps, window) -- correct_transversalium2.
    crop: None or (crop_w, sx0, dx0, ncopy) -- the crop / pad block.  disc: None or (x0, y0, r).
    -> dict(final, cl1, hc, protus, cc: lists of k uint16 GPU images [h, out_w]; factors float64 [k, h] or None;
            detrans: list or None)."""
    k = len(frames)
    ptr, h, w, pitch = ops._img(frames[0], 'frame', torch.uint16)
    for f in frames[1:]:
        if ops._img(f, 'frame', torch.uint16)[1:] != (h, w, pitch):
            raise ValueError('process_frames: the frames must share one shape and row pitch')
    dev = frames[0].device
    crop_w, sx0, dx0, ncopy = (int(v) for v in crop) if crop is not None else (0, 0, 0, 0)
    out_w = crop_w if crop_w > 0 else w
    out_pitch = (out_w + 63) // 64 * 64
    store = torch.empty((k, 5, h, out_pitch), dtype=torch.uint16, device=dev)
    views = [[store[i, j, :, :out_w] for i in range(k)] for j in range(5)]
    detrans, detrans_pitch, detrans_arr = None, 0, None
    if keep_detrans and transversalium is not None:
        detrans_pitch = (w + 63) // 64 * 64
        dstore = torch.empty((k, h, detrans_pitch), dtype=torch.uint16, device=dev)
        detrans = [dstore[i, :, :w] for i in range(k)]
        detrans_arr = _ptr_array(detrans)
    ws = _scratch('process', _sizes('process_ws', lib.shg_stage_process_workspace_bytes, k, h, w, crop_w, int(tiles)), dev)
    pin = _scratch('process', _sizes('process_pin', lib.shg_stage_process_host_bytes, k, h), pinned=True)
    factors = None
    circle = borders = taps = None
    window = 0
    if transversalium is not None:
        circle = np.ascontiguousarray(transversalium['circle'], dtype=np.float64)
        borders = np.ascontiguousarray(transversalium['borders'], dtype=np.float64)
        taps = np.ascontiguousarray(transversalium['taps'], dtype=np.float64)
        window = int(transversalium['window'])
        factors = np.empty((k, h))
    x0, y0, r = (int(v) for v in disc) if disc is not None else (0, 0, 0)
    _lib.check(lib.shg_stage_process_frames(
        _ptr_array(frames), k, h, w, pitch, int(transversalium is not None), None if circle is None else _p(circle),
        None if borders is None else _p(borders), None if taps is None else _p(taps), window, None if factors is None else _p(factors),
        crop_w, sx0, dx0, ncopy, float(clip_limit), int(tiles), x0, y0, r, detrans_arr, detrans_pitch,
        _ptr_array(views[0]), _ptr_array(views[1]), _ptr_array(views[2]), _ptr_array(views[3]), _ptr_array(views[4]), out_pitch,
        ws.data_ptr(), ws.numel(), pin.data_ptr(), pin.numel(), ops._stream()), 'shg_stage_process_frames')
    return {'final': views[0], 'cl1': views[1], 'hc': views[2], 'protus': views[3], 'cc': views[4], 'factors': factors,
            'detrans': detrans}
