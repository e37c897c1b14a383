"""Stage composites: one C call per stage function of the reference (include/shg_hip.h, csrc/stages.hip).

A composite launches the stage's kernels, reads the few values the control plane needs through pinned memory and
runs that control plane in C++, all without the interpreter -- ctypes drops the interpreter lock for the call, so
the scan workers of Solex_recon.solex_do_work overlap for real.  This module owns what a call needs besides its
arguments: a per-thread device workspace and pinned staging areas (a scan worker is a thread with its own stream,
so buffers are never shared between streams), grown on demand and reused from file to file.
"""
import ctypes
import os
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


def _staging_slot(name, nbytes):
    """A pinned staging area the GPU may still be reading when the stage call returns (shg_stage_extract queues a copy kernel
    that reads it): one of a small per-thread ring, each slot guarded by an event recorded after its last use -- a slot is handed
    out again only once that event has completed, and a slot that has to grow is kept until then.  -> (buffer, done(stream))"""
    store = _local.__dict__.setdefault('buffers', {})
    ring = store.setdefault(('ring', name), [])
    slot = None
    for entry in ring:
        if entry[1] is None or entry[1].query():
            slot = entry
            break
    if slot is None and len(ring) >= 4:
        slot = ring[0]
        slot[1].synchronize()
    if slot is None:
        slot = [None, None]
        ring.append(slot)
    if slot[0] is None or slot[0].numel() < nbytes:
        slot[0] = torch.empty(max(int(nbytes), 1), dtype=torch.uint8).pin_memory()      # (the slot's event has completed: nobody reads the old one)
    slot[1] = None

    def done():
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        slot[1] = ev
    return slot[0], done


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
    # the stage returns with its copy kernel still to read the staging area: a slot of its own until that kernel has run
    pin, staged = _staging_slot('extract', need)
    mm_store = torch.empty(s * 130, dtype=torch.int32, device=dev) if want_minmax else None      # 64 slots x 2 per plane, then {min, max}
    try:
        _lib.check(lib.shg_stage_extract(stack.data_ptr(), n, h, w, bpp, ops.frame_stride(stack), _p(fit), _p(sh), s, out.data_ptr(),
                                         out.stride(1), out.stride(0), n_cols, int(k_offset), int(bool(flip_x)),
                                         None if mm_store is None else mm_store.data_ptr(), ws.data_ptr(), ws.numel(),
                                         pin.data_ptr(), pin.numel(), ops._stream()), 'shg_stage_extract')
    finally:
        staged()
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
    # (the answer depends on which limb kernels the library takes: SHG_LIMB_FUSED, asked per call)
    ws = _scratch('limb', _sizes('limb_ws' + os.environ.get('SHG_LIMB_FUSED', '1')[:1], lib.shg_stage_limb_points_workspace_bytes, h, w), dev)
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
    # what solex_process returns (protus, cc) apart from what only the file writers look at (final, cl1, hc): a caller that
    # keeps the results of a batch keeps two images per disk alive, not five
    store = torch.empty((k, 3, h, out_pitch), dtype=torch.uint16, device=dev)
    kept = torch.empty((k, 2, h, out_pitch), dtype=torch.uint16, device=dev)
    views = [[store[i, j, :, :out_w] for i in range(k)] for j in range(3)] + [[kept[i, j, :, :out_w] for i in range(k)] for j in range(2)]
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


# ---- one scan, one call ---------------------------------------------------------------------------------------------
_arena_hint = {}          # (S, k, ih, n, crop, detrans, fit image) -> (arena bytes, results bytes, workspace bytes) the last such scan needed
_NAN = float('nan')
FIRST_GUESS_SCALE = 1.05      # width of the corrected images a first scan of a shape is assumed to have, in raw-disk widths


class _Planes:
    """uint16 images inside a uint8 arena, made on demand: plane(off, h, pitch, w) -> [h, w] view with rows `pitch` elements
    apart at byte offset `off`.  (A view costs microseconds of interpreter time; a scan that writes no files only ever
    looks at two of its eleven images.)"""
    __slots__ = ('words',)

    def __init__(self, buf):
        self.words = buf.view(torch.uint16)

    def plane(self, off, h, pitch, w):
        return torch.as_strided(self.words, (h, w), (pitch, 1), off // 2)


class _LazyImages:
    """list-like: image j of a family of equally spaced planes, built when asked for."""
    __slots__ = ('planes', 'off', 'step', 'count', 'geom')

    def __init__(self, planes, off, step, count, h, pitch, w):
        self.planes, self.off, self.step, self.count, self.geom = planes, off, step, count, (h, pitch, w)

    def __len__(self):
        return self.count

    def __getitem__(self, j):
        if not 0 <= j < self.count:
            raise IndexError(j)
        return self.planes.plane(self.off + j * self.step, *self.geom)


def _host_buffers(name, specs):
    """Per-thread NumPy buffers that outlive a scan (a fresh 2 MB array per scan is an mmap, its page faults and a munmap)."""
    store = _local.__dict__.setdefault('host', {})
    out = []
    for tag, shape, dtype in specs:
        key = (name, tag)
        buf = store.get(key)
        if buf is None or buf.shape != tuple(shape) or buf.dtype != np.dtype(dtype):
            buf = store[key] = np.empty(shape, dtype=dtype)
        out.append(buf)
    return out


_pinned_free = []          # pinned staging areas of scans that went through a pool (a scan leases one for its lifetime)
_pinned_lock = threading.Lock()


PINNED_SLAB = 12           # staging areas cut from one pinned allocation: page-locking memory costs tens of milliseconds per call


def _lease_pinned(nbytes):
    with _pinned_lock:
        for i, buf in enumerate(_pinned_free):
            if buf.numel() >= nbytes:
                return _pinned_free.pop(i)
        each = (max(int(nbytes), 1) + 4095) // 4096 * 4096
        slab = torch.empty(each * PINNED_SLAB, dtype=torch.uint8).pin_memory()
        parts = [slab[i * each:(i + 1) * each] for i in range(PINNED_SLAB)]
        _pinned_free.extend(parts[1:])
        return parts[0]


def _return_pinned(buf):
    with _pinned_lock:
        if len(_pinned_free) < 64:
            _pinned_free.append(buf)


_host_free = {}            # spec tuple -> [lists of NumPy arrays]: host outputs of pooled scans, leased like the staging areas


def _lease_host(specs):
    with _pinned_lock:
        free = _host_free.get(specs)
        if free:
            return free.pop()
    return [np.empty(shape, dtype=dt) for _, shape, dt in specs]


def _return_host(specs, bufs):
    with _pinned_lock:
        free = _host_free.setdefault(specs, [])
        if len(free) < 32:
            free.append(bufs)


def _even256(nbytes):
    return (max(int(nbytes), 256) + 255) // 256 * 256


class ScanCall:
    """One shg_scan_file call (csrc/scan.hip): the whole per-file flow -- pass A, line fit, extraction, limb fit (or the fixed
    ratio / slant of `options`), the warp of every requested disk, transversalium, crop, CLAHE, contrast products.
    stack: the frame stack in HBM; shifts: options['shift'] after solex_read's de-duplication; requested: one flag per
    shift; taps_for(window) -> savgol_coeffs(window, 3).
    run() makes the call on this thread's current stream; submit(pool) / wait() hand it to a native scan pool.
    collect() -> (dict of everything the phases that ran produced, error or None): the caller logs what was reached, then raises."""

    def __init__(self, stack, shifts, requested, options, taps_for, want_plot_data=False, want_fit_image=False, own_buffers=False):
        n, h, w, bpp = ops.stack_geometry(stack)
        dev = stack.device
        ih, iw = (w, h) if w > h else (h, w)
        self.stack, self.dev, self.n, self.ih, self.plot = stack, dev, n, ih, want_plot_data
        self.sh = sh = np.ascontiguousarray(shifts, dtype=np.int32)
        self.rqd = rqd = np.ascontiguousarray(requested, dtype=np.uint8)
        self.s, self.k = s, k = int(sh.size), int(rqd.sum())
        ratio_fixe, slant_fix = options['ratio_fixe'], options['slant_fix']
        self.limb = limb = ratio_fixe is None and slant_fix is None
        self.trans = trans = bool(options['transversalium'])
        keep_detrans = bool(options['save_fit'] and trans)
        fit_image = bool(want_fit_image and not rqd[0])
        self.own = own_buffers

        self.pitch = pitch = (n + 63) // 64 * 64
        self.images = images = torch.empty((2, ih, iw), dtype=torch.uint16, device=dev)
        self.disks = disks = torch.empty((s, ih, pitch), dtype=torch.uint16, device=dev)
        self.mm_store = mm_store = torch.empty(s * 130, dtype=torch.int32, device=dev)
        self.cap = cap = (-(-ih // 4)) * (-(-n // 4)) if limb else 1
        self.specs = specs = (('fit', (ih, 4), 'f8'), ('sharp', (ih,), 'i4'), ('mask', (ih,), 'u1'), ('points', (cap, 2), 'i4'),
                              ('flags', (cap,), 'u1'), ('outline', (100, 2), 'f8'), ('factors', (max(k, 1), ih), 'f8'))
        # a scan in a pool keeps its host arrays to itself (several are in flight per caller thread)
        self.bufs = bufs = _lease_host(specs) if own_buffers else _host_buffers('scan', specs)
        self.fit, self.sharp, self.mask, self.points, self.flags, self.outline, self.factors = bufs
        self.gauss = gauss = _canny_ladder_taps()
        ts = int(options['trans_strength'])
        self.taps = taps = taps_for(ts) if trans and ts > 3 and ts % 2 == 1 else None   # the usual window; any other one: the callback

        self.rq = rq = _lib.ScanRequest()
        rq.struct_bytes = ctypes.sizeof(_lib.ScanRequest)
        rq.stack, rq.n_frames, rq.height, rq.width, rq.frame_stride_px, rq.bytes_per_px = stack.data_ptr(), n, h, w, ops.frame_stride(stack), bpp
        rq.flip_x = int(bool(options['flip_x']))
        rq.host_shifts, rq.host_requested, rq.n_shifts, rq.want_fit_image = _p(sh), _p(rqd), s, int(fit_image)
        rq.ratio_fixe = _NAN if ratio_fixe is None else float(ratio_fixe)
        rq.slant_fix_deg = _NAN if slant_fix is None else float(slant_fix)
        rq.transversalium, rq.keep_detrans, rq.trans_strength = int(trans), int(keep_detrans), ts
        rq.host_taps, rq.taps_window = (None, 0) if taps is None else (_p(taps), ts)
        rq.crop_square = int(bool(options['crop_width_square']))
        rq.has_fixed_width = int(options['fixed_width'] is not None)
        rq.fixed_width = 0 if options['fixed_width'] is None else int(options['fixed_width'])
        rq.disk_display, rq.tiles, rq.delta_radius, rq.clip_limit = int(bool(options['disk_display'])), 2, int(options['delta_radius']), 0.8
        rq.host_gauss_taps = _p(gauss)
        rq.mean_out, rq.max_out = images[0].data_ptr(), images[1].data_ptr()
        rq.disks, rq.disk_pitch, rq.disk_plane_stride, rq.minmax_slots = disks.data_ptr(), pitch, ih * pitch, mm_store.data_ptr()
        rq.host_fit = _p(self.fit)
        rq.host_trace_sharp, rq.host_mask_good = (_p(self.sharp), _p(self.mask)) if want_plot_data else (None, None)
        rq.host_points, rq.host_flags, rq.points_cap = (_p(self.points), _p(self.flags), cap) if limb else (None, None, 0)
        rq.host_outline200 = _p(self.outline) if (limb and want_plot_data) else None
        rq.host_factors = _p(self.factors) if (trans and k) else None

        self.key = key = (s, k, ih, n, bpp, bool(options['crop_width_square']), options['fixed_width'], keep_detrans, fit_image, limb)
        hint = _arena_hint.get(key)
        if hint is None:
            # first scan of this shape: corrected images about as wide as the raw disk (Y/X ratio near 1)
            guess_w = (int(n * FIRST_GUESS_SCALE) + 127) // 64 * 64
            prod_w = max(guess_w, ih if options['crop_width_square'] else 0, rq.fixed_width)
            hint = ((k * (1 + keep_detrans) + fit_image) * ih * guess_w * 2 + k * 3 * ih * prod_w * 2 + (1 << 16), k * 2 * ih * prod_w * 2 + 4096,
                    lib.shg_scan_workspace_bytes(ctypes.byref(rq)) + (lib.shg_stage_process_workspace_bytes(k, ih, guess_w, prod_w, 2) if k else 0) + 4096)
        self.hint = hint
        pin_bytes = _sizes('scan_pin', lambda *a: lib.shg_scan_host_bytes(ctypes.byref(rq)), s, k, n, h, w, bpp, limb)
        self.pin = _lease_pinned(pin_bytes) if own_buffers else _scratch('scan', pin_bytes, pinned=True)
        self.rs = _lib.ScanResult()
        self.status = None
        self.message = None
        self.pool = self.ticket = None
        self._buffers()

    def _buffers(self):
        rq, hint, dev = self.rq, self.hint, self.dev
        self.arena = arena = torch.empty(_even256(hint[0]), dtype=torch.uint8, device=dev)
        self.results = results = torch.empty(_even256(hint[1]), dtype=torch.uint8, device=dev)
        self.ws = ws = torch.empty(_even256(hint[2]), dtype=torch.uint8, device=dev) if self.own else _scratch('scan', hint[2], dev)
        rq.arena, rq.arena_bytes, rq.results, rq.results_bytes = arena.data_ptr(), arena.numel(), results.data_ptr(), results.numel()
        rq.workspace, rq.workspace_bytes = ws.data_ptr(), ws.numel()
        rq.host_pinned, rq.host_pinned_bytes = self.pin.data_ptr(), self.pin.numel()

    def _needs_resume(self):
        """The corrected images turned out larger than the arenas: grow them; the call resumes at the warp (the raw disks and
        the geometry stay)."""
        rs = self.rs
        if self.status == -2 and rs.phase_done == 3 and self.rq.start_phase == 0:
            self.hint = (rs.needed_arena_bytes + 4096, rs.needed_results_bytes + 4096, rs.needed_workspace_bytes + 4096)
            self.rq.start_phase = 3
            self._buffers()
            return True
        return False

    def run(self):
        stream = ops._stream()
        while True:
            self.status = lib.shg_scan_file(ctypes.byref(self.rq), ctypes.byref(self.rs), stream)
            self.message = _lib.last_error() if self.status else None
            if not self._needs_resume():
                return self

    def submit(self, pool):
        self.pool = pool
        ticket = ctypes.c_int64()
        # the stack is whatever this thread's current stream has produced by now: the scan's pass A goes onto the lane behind that
        after = torch.cuda.current_stream(self.stack.device).cuda_stream
        _lib.check(lib.shg_pool_submit_after(pool, ctypes.byref(self.rq), ctypes.byref(self.rs), after, ctypes.byref(ticket)),
                   'shg_pool_submit')
        self.ticket = ticket.value
        return self

    def done(self):
        return self.ticket is None or lib.shg_pool_poll(self.pool, self.ticket) != 0

    def wait(self):
        """Blocks (without the interpreter lock) until the pool has run the scan; resubmits once when the arenas were too small."""
        while self.ticket is not None:
            status = ctypes.c_int()
            buf = ctypes.create_string_buffer(512)
            _lib.check(lib.shg_pool_wait(self.pool, self.ticket, ctypes.byref(status), buf, len(buf)), 'shg_pool_wait')
            self.ticket = None
            self.status = status.value
            self.message = buf.value.decode('utf-8', 'replace') if self.status else None
            if self._needs_resume():
                self.submit(self.pool)
        return self

    def collect(self):
        rs, rq = self.rs, self.rq
        s, k, n = self.s, self.k, self.n
        phase = int(rs.phase_done)
        if (phase >= 3 and _arena_hint.get(self.key) is None) or rq.start_phase == 3:
            # a little headroom: the next file's ellipse differs in the third digit, its images by a few columns
            _arena_hint[self.key] = (int(rs.needed_arena_bytes * 1.03) + 4096, int(rs.needed_results_bytes * 1.03) + 4096,
                                     int(rs.needed_workspace_bytes * 1.03) + 4096)
        error = None
        if self.status != 0:
            error = _lib.take_callback_error(rs.window)
            if error is None:
                try:
                    _lib.check(self.status, 'shg_scan_file', self.message)
                except Exception as e:      # noqa: BLE001 -- handed to the caller, which logs what the scan reached first
                    error = e
        out = {'phase': phase, 'mean': self.images[0], 'max': self.images[1], 'limb_fitted': bool(rs.limb_fitted)}
        if phase >= 1:
            y1, y2 = int(rs.y1), int(rs.y2)
            out['y1'], out['y2'], out['p'] = y1, y2, np.array(rs.p4)
            if self.plot:
                out['fit'] = self.fit.copy()
                out['sharp'] = self.sharp.copy()
                out['mask_good'] = self.mask[:max(y2 - y1, 0)].astype(bool)
        if phase >= 2:
            out['disks'] = self.disks[:, :, :n]
            out['extrema'] = self.mm_store[s * 128:].view(s, 2)
        if phase >= 3:
            out['phi'], out['ratio'] = float(rs.phi), float(rs.ratio)
            out['theta_first'] = float(rs.theta_first)
            out['circle'] = (rs.circle3[0], rs.circle3[1], rs.circle3[2])
            out['borders'] = list(rs.borders4)
            if rs.limb_fitted and self.plot:
                m = int(rs.counts3[0])
                pts, fl = self.points[:m].astype(np.int64) * 4, self.flags[:m]
                out['raw_X'], out['X_f'], out['outline'] = pts, pts[(fl & 2) != 0].astype(float), self.outline.copy()
        if phase >= 4:
            oh, ow, fp, pw, pp = int(rs.out_h), int(rs.out_w), int(rs.frame_pitch), int(rs.prod_w), int(rs.prod_pitch)
            fb = (oh * fp * 2 + 255) // 256 * 256
            pb = (oh * pp * 2 + 255) // 256 * 256
            ap, rp = _Planes(self.arena), _Planes(self.results)
            out['fit_image'] = ap.plane(int(rs.fit_image_off), oh, fp, ow) if rs.fit_image_off >= 0 else None
            out['frames'] = _LazyImages(ap, int(rs.frames_off), fb, k, oh, fp, ow)
            out['detrans'] = _LazyImages(ap, int(rs.detrans_off), fb, k, oh, fp, ow) if rs.detrans_off >= 0 else None
            for idx, name in enumerate(('final', 'cl1', 'hc')):
                out[name] = _LazyImages(ap, int(rs.products_off) + idx * pb, 3 * pb, k, oh, pp, pw)
            for idx, name in enumerate(('protus', 'cc')):
                out[name] = _LazyImages(rp, idx * pb, 2 * pb, k, oh, pp, pw)
            out['factors'] = self.factors[:k].copy() if (self.trans and k) else None
        if self.own:
            _return_pinned(self.pin)                           # (the pool worker synchronised its stream: nothing reads it any more)
            _return_host(self.specs, self.bufs)                # (everything handed out above is a copy)
            self.pin = self.bufs = self.fit = self.sharp = self.mask = self.points = self.flags = self.outline = self.factors = None
        return out, error


def scan_file(stack, shifts, requested, options, taps_for, want_plot_data=False, want_fit_image=False):
    """ScanCall on this thread's current stream, start to finish."""
    return ScanCall(stack, shifts, requested, options, taps_for, want_plot_data, want_fit_image).run().collect()
