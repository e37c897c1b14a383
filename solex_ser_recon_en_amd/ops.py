"""Thin tensor-level wrappers over the C ABI (include/shg_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
computation below happens in libshg_hip.so.  All tensors must live on the GPU; a
CPU tensor raises (there is no CPU path).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import lib


def _stream():
    # raw hipStream_t of torch's current stream (the public accessor builds a Stream object: ~1 us per call)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _dev(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('%s must be a GPU tensor: the SHG hot path has no CPU fallback' % what)
    return t


def _img(t, what, dtype=None):
    """(ptr, h, w, pitch) of a 2-D image view with unit column stride."""
    _dev(t, what)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError('%s must be a 2-D image with contiguous rows' % what)
    if dtype is not None and t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (what, dtype, t.dtype))
    return t.data_ptr(), t.shape[0], t.shape[1], t.stride(0)


def pitched_u16(h, w, device, align=64):
    """uint16 image whose row pitch is a multiple of `align` elements (128 B rows)."""
    pitch = (w + align - 1) // align * align
    return torch.empty((h, pitch), dtype=torch.uint16, device=device)[:, :w]


def padded_stack(n, h, w, dtype, device):
    """An [n, h, w] frame stack whose frame pitch is shg_frame_pitch_bytes (frame size rounded up to 8 KiB):
    the layout video_reader.device_stack() uploads into and the frame-walking kernels read fastest."""
    item = torch.empty((), dtype=dtype).element_size()
    pitch = lib.shg_frame_pitch_bytes(h * w * item) // item
    store = torch.empty(n * pitch, dtype=dtype, device=device)
    return torch.as_strided(store, (n, h, w), (pitch, w, 1))


def stack_to_host(stack):
    """Dense NumPy copy [n, H, W] of a (possibly pitched) frame stack."""
    if stack.dtype == torch.uint16:
        return stack.view(torch.int16).contiguous().cpu().numpy().view(np.uint16)
    return stack.contiguous().cpu().numpy()


def frame_stride(stack):
    return stack.stride(0) if stack.shape[0] > 1 else stack.shape[1] * stack.shape[2]


def stack_geometry(stack):
    """(n, H, W, bytes_per_px) of a frame stack in file layout (frames dense, frame pitch >= H*W)."""
    _dev(stack, 'stack')
    if stack.dim() != 3 or stack.stride(2) != 1 or stack.stride(1) != stack.shape[2] or (
            stack.shape[0] > 1 and stack.stride(0) < stack.shape[1] * stack.shape[2]):
        raise ValueError('stack must be an [N, Height, Width] tensor with dense frames')
    if stack.dtype == torch.uint8:
        bpp = 1
    elif stack.dtype in (torch.uint16, torch.int16):
        bpp = 2
    else:
        raise TypeError('stack must be uint8 or uint16, got %s' % stack.dtype)
    n, h, w = stack.shape
    return n, h, w, bpp


# ---- pass A ---------------------------------------------------------------
def accumulate_sum_max(stack, workspace=None):
    """-> (sum int64 [H*W], max uint16 [H*W]) in file layout, raw sample units."""
    n, h, w, bpp = stack_geometry(stack)
    dev = stack.device
    need = lib.shg_accumulate_workspace_bytes(n, h, w, bpp)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    total = torch.empty(h * w, dtype=torch.int64, device=dev)
    mx = torch.empty(h * w, dtype=torch.uint16, device=dev)
    _lib.check(lib.shg_accumulate_sum_max(stack.data_ptr(), n, h, w, bpp, frame_stride(stack), total.data_ptr(), mx.data_ptr(),
                                          workspace.data_ptr(), workspace.numel(), _stream()),
               'shg_accumulate_sum_max')
    return total, mx


def accumulate_mean_max(stack, workspace=None):
    """compute_mean_max for a stack that is whole on this GPU -> (mean uint16 [ih, iw], max uint16 [ih, iw]):
    pass A and the division / rotation straight from its per-slab partials."""
    n, h, w, bpp = stack_geometry(stack)
    dev = stack.device
    need = lib.shg_accumulate_workspace_bytes(n, h, w, bpp)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    ih, iw = (w, h) if w > h else (h, w)
    mean = torch.empty((ih, iw), dtype=torch.uint16, device=dev)
    mout = torch.empty((ih, iw), dtype=torch.uint16, device=dev)
    _lib.check(lib.shg_accumulate_mean_max(stack.data_ptr(), n, h, w, bpp, frame_stride(stack), mean.data_ptr(), mout.data_ptr(),
                                           workspace.data_ptr(), workspace.numel(), _stream()), 'shg_accumulate_mean_max')
    return mean, mout


def pass_a_prelaunch(stack, workspace):
    """Start pass A of `stack` on the device's frame-pass lane now, behind what the current stream holds (shg_pass_a_prelaunch);
    accumulate_mean_max(stack, workspace) later finds it there.  -> False when the device has no lane (nothing was launched)."""
    n, h, w, bpp = stack_geometry(stack)
    need = lib.shg_accumulate_workspace_bytes(n, h, w, bpp)
    if workspace.numel() < need:
        raise ValueError('pass_a_prelaunch: workspace of %d bytes, %d needed' % (workspace.numel(), need))
    launched = ctypes.c_int()
    _lib.check(lib.shg_pass_a_prelaunch(stack.data_ptr(), n, h, w, bpp, frame_stride(stack), workspace.data_ptr(), workspace.numel(),
                                        _stream(), ctypes.byref(launched)), 'shg_pass_a_prelaunch')
    return bool(launched.value)


def finalize_mean_max(total, mx, n_total, height, width, bpp):
    """-> (mean uint16 [ih, iw], max uint16 [ih, iw]) in the reference's orientation."""
    _dev(total, 'sum')
    ih, iw = (width, height) if width > height else (height, width)
    mean = torch.empty((ih, iw), dtype=torch.uint16, device=total.device)
    mout = torch.empty((ih, iw), dtype=torch.uint16, device=total.device)
    _lib.check(lib.shg_finalize_mean_max(total.data_ptr(), mx.data_ptr(), int(n_total), height, width, bpp,
                                         mean.data_ptr(), mout.data_ptr(), _stream()), 'shg_finalize_mean_max')
    return mean, mout


def reduce_frame_stats(pieces, npix):
    """pieces: int32 GPU tensor [G, words], every row one rank's [npix sums as 32-bit words | npix maxima as 16-bit words | ...]
    (dist.exchange_frame_stats' all-gather) -> (int64 [npix] sums, uint16 [npix] maxima) over all ranks, one launch."""
    _dev(pieces, 'pieces')
    if pieces.dtype != torch.int32 or pieces.dim() != 2 or not pieces.is_contiguous():
        raise TypeError('pieces must be a contiguous int32 [G, words] tensor')
    total = torch.empty(int(npix), dtype=torch.int64, device=pieces.device)
    mx = torch.empty(int(npix), dtype=torch.uint16, device=pieces.device)
    _lib.check(lib.shg_reduce_frame_stats(pieces.data_ptr(), int(pieces.shape[0]), int(pieces.shape[1]), int(npix), total.data_ptr(),
                                          mx.data_ptr(), _stream()), 'shg_reduce_frame_stats')
    return total, mx


# ---- line detection helpers -------------------------------------------------
def box_blur_u16(img, kw, kh):
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    if pitch != w:
        raise ValueError('box_blur_u16 needs a dense image')
    out = torch.empty_like(img)
    tmp = torch.empty(h * w, dtype=torch.int32, device=img.device)
    _lib.check(lib.shg_box_blur_u16(ptr, h, w, int(kw), int(kh), out.data_ptr(), tmp.data_ptr(), _stream()),
               'shg_box_blur_u16')
    return out


def row_argmin_u16(img, x0, x1):
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    if pitch != w:
        raise ValueError('row_argmin_u16 needs a dense image')
    out = torch.empty(h, dtype=torch.int32, device=img.device)
    _lib.check(lib.shg_row_argmin_u16(ptr, h, w, int(x0), int(x1), out.data_ptr(), _stream()), 'shg_row_argmin_u16')
    return out


def row_mean_u16(img):
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    if pitch != w:
        raise ValueError('row_mean_u16 needs a dense image')
    out = torch.empty(h, dtype=torch.float64, device=img.device)
    _lib.check(lib.shg_row_mean_u16(ptr, h, w, out.data_ptr(), _stream()), 'shg_row_mean_u16')
    return out


def blur_row_mean_u16(img, kw, kh):
    """np.mean(cv2.blur(img, (kw, kh)), axis=1) without materialising the blurred image -> float64 [h]."""
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    if pitch != w:
        raise ValueError('blur_row_mean_u16 needs a dense image')
    out = torch.empty(h, dtype=torch.float64, device=img.device)
    _lib.check(lib.shg_blur_row_mean_u16(ptr, h, w, int(kw), int(kh), out.data_ptr(), _stream()), 'shg_blur_row_mean_u16')
    return out


def blur_argmin_u16(img, kw, kh, x0, x1):
    """(np.argmin(cv2.blur(img, (kw, kh))[:, x0:x1], axis=1), np.argmin(img, axis=1)) -> two int32 [h] tensors."""
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    if pitch != w:
        raise ValueError('blur_argmin_u16 needs a dense image')
    out = torch.empty((2, h), dtype=torch.int32, device=img.device)
    _lib.check(lib.shg_blur_argmin_u16(ptr, h, w, int(kw), int(kh), int(x0), int(x1), out[0].data_ptr(), out[1].data_ptr(), _stream()),
               'shg_blur_argmin_u16')
    return out[0], out[1]


# ---- pass B ---------------------------------------------------------------
def extract_columns(stack, ind_l, lw, rw, n_cols=None, k_offset=0, flip_x=False, out=None):
    """-> disks uint16 [S, ih, n_cols] (a view of a row-pitched buffer).

    ind_l int32 [S, ih] (clamped), lw/rw float64 [ih]: host arrays or GPU tensors."""
    n, h, w, bpp = stack_geometry(stack)
    dev = stack.device
    ih = max(h, w)
    ind_l = torch.as_tensor(np.ascontiguousarray(ind_l, dtype=np.int32) if not isinstance(ind_l, torch.Tensor) else ind_l).to(dev)
    if not isinstance(lw, torch.Tensor) and not isinstance(rw, torch.Tensor) and np.shape(lw) == np.shape(rw):
        both = torch.from_numpy(np.stack([np.asarray(lw, dtype=np.float64), np.asarray(rw, dtype=np.float64)])).to(dev)   # one upload
        lw, rw = both[0], both[1]
    else:
        lw = torch.as_tensor(np.ascontiguousarray(lw, dtype=np.float64) if not isinstance(lw, torch.Tensor) else lw).to(dev)
        rw = torch.as_tensor(np.ascontiguousarray(rw, dtype=np.float64) if not isinstance(rw, torch.Tensor) else rw).to(dev)
    if ind_l.dim() != 2 or ind_l.shape[1] != ih or lw.shape != (ih,) or rw.shape != (ih,):
        raise ValueError('ind_l must be [S, %d] and lw, rw [%d]' % (ih, ih))
    if ind_l.dtype != torch.int32 or lw.dtype != torch.float64 or rw.dtype != torch.float64:
        raise TypeError('ind_l must be int32 and lw, rw float64')
    s = ind_l.shape[0]
    n_cols = n if n_cols is None else int(n_cols)
    if out is None:
        pitch = (n_cols + 63) // 64 * 64
        # every column is written when this call covers the whole scan; a shard's call leaves the others' columns zero
        alloc = torch.empty if (n_cols == n and int(k_offset) == 0) else torch.zeros
        out = alloc((s, ih, pitch), dtype=torch.uint16, device=dev)[:, :, :n_cols]
    if out.shape != (s, ih, n_cols) or out.stride(2) != 1:
        raise ValueError('out must be a [S, ih, n_cols] view with unit column stride')
    _lib.check(lib.shg_extract_columns(stack.data_ptr(), n, h, w, bpp, frame_stride(stack), ind_l.contiguous().data_ptr(),
                                       lw.contiguous().data_ptr(), rw.contiguous().data_ptr(), s, out.data_ptr(),
                                       out.stride(1), out.stride(0), n_cols, int(k_offset), int(bool(flip_x)),
                                       _stream()), 'shg_extract_columns')
    return out


def extract_columns_dense(stack, fit, shifts, flip_x=False, want_minmax=False):
    """shg_extract_columns_dense for consecutive shifts (any order, 3..24 of them): every distinct sample of a (row, frame) loaded
    once.  fit: host [ih, 4] as compute_mean_return_fit returns it.  -> disks uint16 [S, ih, n] (and the extrema int32 [S, 2])."""
    from . import hostmath
    n, h, w, bpp = stack_geometry(stack)
    dev = stack.device
    ih, iw = (w, h) if w > h else (h, w)
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    s = int(sh.size)
    if not lib.shg_extract_dense_fits(sh.ctypes.data, s):
        raise ValueError('extract_columns_dense needs 3..24 consecutive shifts')
    fit = np.ascontiguousarray(fit, dtype=np.float64)
    ind_l, lw, rw = hostmath.column_plan(fit, sh, ih, iw)
    base = (fit[:, 0] + 1.0 * float(sh.min())).astype(np.int64).astype(np.int32)
    ind_d = torch.from_numpy(ind_l).to(dev)
    w_d = torch.from_numpy(np.stack([lw, rw])).to(dev)
    base_d = torch.from_numpy(base).to(dev)
    pitch = (n + 63) // 64 * 64
    out = torch.empty((s, ih, pitch), dtype=torch.uint16, device=dev)[:, :, :n]
    mm = torch.empty(s * 130, dtype=torch.int32, device=dev) if want_minmax else None
    _lib.check(lib.shg_extract_columns_dense(stack.data_ptr(), n, h, w, bpp, frame_stride(stack), ind_d.data_ptr(), base_d.data_ptr(),
                                             w_d[0].data_ptr(), w_d[1].data_ptr(), sh.ctypes.data, s, out.data_ptr(), out.stride(1), out.stride(0),
                                             n, 0, int(bool(flip_x)), None if mm is None else mm.data_ptr(), _stream()), 'shg_extract_columns_dense')
    return (out, mm[s * 128:].view(s, 2)) if want_minmax else out


# ---- post-processing ----------------------------------------------------------
def warp_rows_u16(img, h00, h01, h02, out_h, out_w, minmax=None):
    """minmax: the plane's extrema as shg_extract_columns_minmax left them (int32 [2] = {min, max}); None = found here."""
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    out = pitched_u16(out_h, out_w, img.device)
    if minmax is not None:
        _lib.check(lib.shg_warp_rows_minmax_u16(ptr, h, w, pitch, float(h00), float(h01), float(h02), out.data_ptr(), int(out_h),
                                                int(out_w), out.stride(0), minmax.data_ptr(), _stream()), 'shg_warp_rows_minmax_u16')
        return out
    mm = torch.empty(2, dtype=torch.int32, device=img.device)
    _lib.check(lib.shg_warp_rows_u16(ptr, h, w, pitch, float(h00), float(h01), float(h02), out.data_ptr(),
                                     int(out_h), int(out_w), out.stride(0), mm.data_ptr(), _stream()),
               'shg_warp_rows_u16')
    return out


def _row_factor_ptr(row_factor, h, device):
    if row_factor is None:
        return None, None
    rf = row_factor if isinstance(row_factor, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(row_factor, dtype=np.float64))
    rf = rf.to(device).contiguous()
    if rf.shape != (h,) or rf.dtype != torch.float64:
        raise ValueError('row_factor must be float64 [h]')
    return rf, rf.data_ptr()


def rowpair_logratio_stats(img, y1, y2, xa, xb, row_factor=None):
    """xa, xb: int32 host arrays [y2-y1] of NumPy-normalised slice bounds. -> float64 tensor [y2-y1]."""
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    if not isinstance(xa, torch.Tensor):
        xa = torch.as_tensor(np.ascontiguousarray(xa, dtype=np.int32)).to(img.device)
    if not isinstance(xb, torch.Tensor):
        xb = torch.as_tensor(np.ascontiguousarray(xb, dtype=np.int32)).to(img.device)
    if xa.numel() != y2 - y1 or xb.numel() != y2 - y1 or xa.dtype != torch.int32 or xb.dtype != torch.int32:
        raise ValueError('xa, xb must be int32 with y2 - y1 entries')
    rf, rf_ptr = _row_factor_ptr(row_factor, h, img.device)
    out = torch.empty(y2 - y1, dtype=torch.float64, device=img.device)
    _lib.check(lib.shg_rowpair_logratio_stats(ptr, h, w, pitch, int(y1), int(y2), xa.data_ptr(), xb.data_ptr(), rf_ptr,
                                              out.data_ptr(), _stream()), 'shg_rowpair_logratio_stats')
    return out


def scale_rows_u16(img, c, row_factor=None):
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    c = torch.as_tensor(np.ascontiguousarray(c, dtype=np.float64)).to(img.device) if not isinstance(c, torch.Tensor) else c
    if c.shape != (h,) or c.dtype != torch.float64:
        raise ValueError('c must be float64 [h]')
    rf, rf_ptr = _row_factor_ptr(row_factor, h, img.device)
    out = pitched_u16(h, w, img.device)
    _lib.check(lib.shg_scale_rows_u16(ptr, h, w, pitch, c.contiguous().data_ptr(), rf_ptr, out.data_ptr(), out.stride(0),
                                      _stream()), 'shg_scale_rows_u16')
    return out


_TAPS = {}


def correlate1d_rows_f64(rows, weights):
    """scipy.ndimage.correlate1d(rows, weights, axis=-1, mode='constant') for a float64 GPU tensor [k, n]; `weights`
    is a host array of odd length (as correlate1d takes them).  SciPy's symmetric-filter test is made here."""
    _dev(rows, 'rows')
    if rows.dim() != 2 or rows.dtype != torch.float64 or not rows.is_contiguous():
        raise ValueError('rows must be a contiguous float64 [k, n] tensor')
    w = np.ascontiguousarray(weights, dtype=np.float64)
    if w.ndim != 1 or not w.size & 1:
        raise ValueError('weights must be 1-D of odd length')
    key = (str(rows.device), w.tobytes())
    if key not in _TAPS:
        r = w.size // 2
        eps = np.finfo(np.float64).eps                                                          # NI_Correlate1D's tests
        sym = 1 if all(abs(w[r + i] - w[r - i]) <= eps for i in range(1, r + 1)) else (
            -1 if all(abs(w[r + i] + w[r - i]) <= eps for i in range(1, r + 1)) else 0)
        if len(_TAPS) > 64:
            _TAPS.clear()
        _TAPS[key] = (torch.from_numpy(w).to(rows.device), r, int(sym))
    wd, r, sym = _TAPS[key]
    out = torch.empty_like(rows)
    _lib.check(lib.shg_correlate1d_rows_f64(rows.data_ptr(), rows.shape[0], rows.shape[1], wd.data_ptr(), r, sym, out.data_ptr(),
                                            _stream()), 'shg_correlate1d_rows_f64')
    return out


_LOG_LUT = {}


def _log_lut(device):
    """float32 log of every uint16 value exactly as NumPy computes np.log(uint16 image) on this host."""
    key = str(device)
    if key not in _LOG_LUT:
        with np.errstate(divide='ignore'):
            lut = np.log(np.arange(65536, dtype=np.uint16))
        assert lut.dtype == np.float32
        _LOG_LUT[key] = torch.from_numpy(lut).to(device)
    return _LOG_LUT[key]


def lin_filter_u16(img, flagged, up, dn, taper, xa, xb, edge, edge_half, linlen=101, half_width=5, row_factor=None):
    """apply_lin_filter + fix_edge_effect (solex_util.py:277-375) -> uint16 image (saturated, truncated).
    flagged / up / dn / taper / xa / xb / edge: per-row host arrays (see include/shg_hip.h)."""
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    dev = img.device

    def put(a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        if a.shape != (h,):
            raise ValueError('per-row arrays must have shape (%d,)' % h)
        return torch.from_numpy(a).to(dev)
    flagged_d, up_d, dn_d = put(flagged, np.uint8), put(up, np.int32), put(dn, np.int32)
    taper_d, xa_d, xb_d, edge_d = put(taper, np.float64), put(xa, np.int32), put(xb, np.int32), put(edge, np.uint8)
    rf, rf_ptr = _row_factor_ptr(row_factor, h, dev)
    lut = _log_lut(dev)
    hl = torch.empty((h, w), dtype=torch.float64, device=dev)
    hf = torch.empty((h, w), dtype=torch.float64, device=dev)
    _lib.check(lib.shg_lin_filter_row_sums(ptr, h, w, pitch, rf_ptr, lut.data_ptr(), flagged_d.data_ptr(), up_d.data_ptr(),
                                           dn_d.data_ptr(), int(linlen), hl.data_ptr(), hf.data_ptr(), _stream()),
               'shg_lin_filter_row_sums')
    out = pitched_u16(h, w, dev)
    _lib.check(lib.shg_lin_filter_apply(ptr, h, w, pitch, rf_ptr, hl.data_ptr(), hf.data_ptr(), int(linlen), int(half_width),
                                        taper_d.data_ptr(), xa_d.data_ptr(), xb_d.data_ptr(), edge_d.data_ptr(), int(edge_half),
                                        out.data_ptr(), out.stride(0), _stream()), 'shg_lin_filter_apply')
    return out


def line_order_stats_u16(img, axis, rank_lo, rank_hi):
    """-> (lo, hi) uint16 GPU tensors: per column (axis 0) / row (axis 1) order statistics."""
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    lines = w if axis == 0 else h
    out = torch.empty((2, lines), dtype=torch.uint16, device=img.device)
    _lib.check(lib.shg_line_order_stats_u16(ptr, h, w, pitch, int(axis), int(rank_lo), int(rank_hi), out[0].data_ptr(),
                                            out[1].data_ptr(), _stream()), 'shg_line_order_stats_u16')
    return out[0], out[1]


def crop_pad_u16(img, nw, sx0, dx0, n, fill):
    """fill: 0..65535, or None for img[0, 0] (read on the device)."""
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    out = pitched_u16(h, nw, img.device)
    _lib.check(lib.shg_crop_pad_u16(ptr, h, w, pitch, out.data_ptr(), int(nw), out.stride(0), int(sx0), int(dx0),
                                    int(n), -1 if fill is None else int(fill), _stream()), 'shg_crop_pad_u16')
    return out


def clahe(img, clip_limit=0.8, tiles=2, small_workspace=False):
    """small_workspace: give shg_clahe only shg_clahe_workspace_bytes (it then builds the 16-bit tile histograms with
    global atomics); the default is the roomier shg_clahe_workspace_bytes_for."""
    _dev(img, 'img')
    if img.dtype == torch.uint16:
        bpp = 2
    elif img.dtype == torch.uint8:
        bpp = 1
    else:
        raise TypeError('clahe needs uint8 or uint16')
    ptr, h, w, pitch = _img(img, 'img')
    need = lib.shg_clahe_workspace_bytes(int(tiles), bpp) if small_workspace else lib.shg_clahe_workspace_bytes_for(h, w, int(tiles), bpp)
    if need == 0:
        raise ValueError('clahe: unsupported tile count %r' % (tiles,))
    ws = torch.empty(need, dtype=torch.uint8, device=img.device)
    out = pitched_u16(h, w, img.device) if bpp == 2 else torch.empty((h, w), dtype=torch.uint8, device=img.device)
    _lib.check(lib.shg_clahe(ptr, h, w, pitch, bpp, float(clip_limit), int(tiles), out.data_ptr(), out.stride(0),
                             ws.data_ptr(), need, _stream()), 'shg_clahe')
    return out


def contrast_stats_u16(frame, ranks_frame, ranks_cl1, out5, clip_limit=0.8, tiles=2, small_workspace=False):
    """cl1 = clahe(frame) plus the order statistics of frame (2 ranks) and cl1 (3 ranks) into out5 (float64 [5], GPU):
    one C call for the first half of image_process.  -> cl1"""
    import ctypes
    ptr, h, w, pitch = _img(frame, 'frame', torch.uint16)
    need = (lib.shg_contrast_stats_workspace_bytes(int(tiles)) if small_workspace
            else lib.shg_contrast_stats_workspace_bytes_for(h, w, int(tiles)))
    if need == 0:
        raise ValueError('contrast_stats: unsupported tile count %r' % (tiles,))
    ws = torch.empty(need, dtype=torch.uint8, device=frame.device)
    cl1 = pitched_u16(h, w, frame.device)
    rf = (ctypes.c_int64 * 2)(*[int(r) for r in ranks_frame])
    rc = (ctypes.c_int64 * 3)(*[int(r) for r in ranks_cl1])
    _lib.check(lib.shg_contrast_stats_u16(ptr, h, w, pitch, float(clip_limit), int(tiles), cl1.data_ptr(), cl1.stride(0), rf, rc,
                                          out5.data_ptr(), ws.data_ptr(), need, _stream()), 'shg_contrast_stats_u16')
    return cl1


def contrast_products_u16(frame, cl1, lo_hi6, disc=None):
    """-> (high_contrast, protus, cc): the three rescale_brightness calls of image_process and the protuberance disc
    (disc = (x0, y0, r) or None) in one C call."""
    import ctypes
    ptr, h, w, pitch = _img(frame, 'frame', torch.uint16)
    cptr, ch, cw, cpitch = _img(cl1, 'cl1', torch.uint16)
    if (ch, cw) != (h, w):
        raise ValueError('frame and cl1 must share one shape')
    store = torch.empty((3, h, (w + 63) // 64 * 64), dtype=torch.uint16, device=frame.device)
    hc, protus, cc = store[0, :, :w], store[1, :, :w], store[2, :, :w]
    bounds = (ctypes.c_double * 6)(*[float(v) for v in lo_hi6])
    x0, y0, r = (int(v) for v in disc) if disc is not None else (0, 0, 0)
    _lib.check(lib.shg_contrast_products_u16(ptr, pitch, cptr, cpitch, h, w, bounds, hc.data_ptr(), protus.data_ptr(), cc.data_ptr(),
                                             store.stride(1), x0, y0, r, _stream()), 'shg_contrast_products_u16')
    return hc, protus, cc


def histogram(img):
    """-> int32 tensor [65536] (uint16 images) or [256] (uint8)."""
    _dev(img, 'img')
    bpp = 2 if img.dtype == torch.uint16 else 1
    if img.dtype not in (torch.uint16, torch.uint8):
        raise TypeError('histogram needs uint8 or uint16')
    ptr, h, w, pitch = _img(img, 'img')
    hist = torch.empty(65536 if bpp == 2 else 256, dtype=torch.int32, device=img.device)
    _lib.check(lib.shg_hist(ptr, h, w, pitch, bpp, hist.data_ptr(), _stream()), 'shg_hist')
    return hist


def rescale_u8(img, lo, hi, alpha=1.0):
    ptr, h, w, pitch = _img(img, 'img', torch.uint8)
    out = torch.empty((h, w), dtype=torch.uint8, device=img.device)
    _lib.check(lib.shg_rescale_u8(ptr, h, w, pitch, float(lo), float(hi), float(alpha), out.data_ptr(), out.stride(0), _stream()),
               'shg_rescale_u8')
    return out


def rescale_u16(img, lo, hi, alpha=1.0):
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    out = pitched_u16(h, w, img.device)
    _lib.check(lib.shg_rescale_u16(ptr, h, w, pitch, float(lo), float(hi), float(alpha), out.data_ptr(),
                                   out.stride(0), _stream()), 'shg_rescale_u16')
    return out


def fill_disc_u16(img, x0, y0, r, value):
    """In place (cv2.circle returns its argument too)."""
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    _lib.check(lib.shg_fill_disc_u16(ptr, h, w, pitch, int(x0), int(y0), int(r), int(value), None, _stream()),
               'shg_fill_disc_u16')
    return img


def downscale_mean_u16(img, factor=4):
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    oh, ow = -(-h // factor), -(-w // factor)
    out = torch.empty((oh, ow), dtype=torch.float64, device=img.device)
    _lib.check(lib.shg_downscale_mean_u16(ptr, h, w, pitch, int(factor), out.data_ptr(), _stream()),
               'shg_downscale_mean_u16')
    return out


# ---- limb detection on the block-mean image -----------------------------------------
def box_blur_f64(img, k):
    _dev(img, 'img')
    if img.dtype != torch.float64 or img.dim() != 2 or not img.is_contiguous():
        raise TypeError('box_blur_f64 needs a dense float64 image')
    h, w = img.shape
    out = torch.empty_like(img)
    tmp = torch.empty_like(img)
    _lib.check(lib.shg_box_blur_f64(img.data_ptr(), h, w, int(k), out.data_ptr(), tmp.data_ptr(), _stream()),
               'shg_box_blur_f64')
    return out


def box_blur_key_f64(img, k):
    """cv2.blur of a block-mean image (whole numbers of 2^-20 below 1) -> (blurred float64, window sums uint32 in 2^-20 units)."""
    _dev(img, 'img')
    if img.dtype != torch.float64 or img.dim() != 2 or not img.is_contiguous():
        raise TypeError('box_blur_key_f64 needs a dense float64 image')
    h, w = img.shape
    out = torch.empty_like(img)
    tmp = torch.empty_like(img)
    keys = torch.empty((h, w), dtype=torch.int32, device=img.device)
    _lib.check(lib.shg_box_blur_key_f64(img.data_ptr(), h, w, int(k), out.data_ptr(), keys.data_ptr(), tmp.data_ptr(), _stream()),
               'shg_box_blur_key_f64')
    return out, keys


def select_keys_u32(keys, ranks, ks):
    """out[i] = the ranks[i]-th smallest window sum of keys[i], as the blurred value (key * 2^-20) / (ks[i] ** 2)."""
    import ctypes
    n = keys[0].numel()
    ptrs = (ctypes.c_void_p * len(keys))(*[k.data_ptr() for k in keys])
    rk = (ctypes.c_int64 * len(ranks))(*[int(r) for r in ranks])
    kk = (ctypes.c_int * len(ks))(*[int(v) for v in ks])
    need = lib.shg_select_keys_workspace_bytes(len(keys))
    ws = torch.empty(need, dtype=torch.uint8, device=keys[0].device)
    out = torch.empty(len(keys), dtype=torch.float64, device=keys[0].device)
    _lib.check(lib.shg_select_keys_u32(ptrs, n, rk, kk, len(keys), out.data_ptr(), ws.data_ptr(), need, _stream()), 'shg_select_keys_u32')
    return out


def gaussian_taps(sigma, truncate=4.0):
    """scipy.ndimage's _gaussian_kernel1d(sigma, 0, radius): the taps gaussian_filter correlates with."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return phi / phi.sum(), radius


def canny_masks(blurred, flood_thresh, sigma, low, high):
    """-> (low_mask, high_mask) uint8 GPU tensors [h, w]."""
    _dev(blurred, 'blurred')
    if blurred.dtype != torch.float64 or blurred.dim() != 2 or not blurred.is_contiguous():
        raise TypeError('canny_masks needs a dense float64 image')
    h, w = blurred.shape
    taps, radius = gaussian_taps(sigma)
    taps = np.ascontiguousarray(taps, dtype=np.float64)
    need = lib.shg_canny_workspace_bytes(h, w)
    ws = torch.empty(need, dtype=torch.uint8, device=blurred.device)
    masks = torch.empty((2, h, w), dtype=torch.uint8, device=blurred.device)
    import ctypes
    _lib.check(lib.shg_canny_masks_f64(blurred.data_ptr(), h, w, float(flood_thresh),
                                       taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), radius, float(low), float(high),
                                       masks[0].data_ptr(), masks[1].data_ptr(), ws.data_ptr(), need, _stream()),
               'shg_canny_masks_f64')
    return masks[0], masks[1]


def select_f64(values, ranks):
    """-> float64 GPU tensor: the ranks[i]-th smallest elements (0-based) of a dense float64 tensor."""
    import ctypes
    _dev(values, 'values')
    if values.dtype != torch.float64 or not values.is_contiguous():
        raise TypeError('select_f64 needs a dense float64 tensor')
    ranks = [int(r) for r in ranks]
    arr = (ctypes.c_int64 * len(ranks))(*ranks)
    need = lib.shg_select_workspace_bytes(len(ranks))
    if need == 0:
        raise ValueError('select_f64 takes 1..8 ranks')
    ws = torch.empty(need, dtype=torch.uint8, device=values.device)
    out = torch.empty(len(ranks), dtype=torch.float64, device=values.device)
    _lib.check(lib.shg_select_f64(values.data_ptr(), values.numel(), arr, len(ranks), out.data_ptr(), ws.data_ptr(), need,
                                  _stream()), 'shg_select_f64')
    return out


def select_multi_f64(arrays, ranks):
    """out[i] = the ranks[i]-th smallest element of arrays[i] (dense float64 GPU tensors of one size), one launch sequence."""
    import ctypes
    if len(arrays) != len(ranks) or not 1 <= len(ranks) <= 8:
        raise ValueError('select_multi_f64 takes 1..8 (array, rank) pairs')
    n = arrays[0].numel()
    for a in arrays:
        _dev(a, 'array')
        if a.dtype != torch.float64 or not a.is_contiguous() or a.numel() != n:
            raise TypeError('select_multi_f64 needs dense float64 tensors of one size')
    ptrs = (ctypes.c_void_p * len(arrays))(*[a.data_ptr() for a in arrays])
    rk = (ctypes.c_int64 * len(ranks))(*[int(r) for r in ranks])
    need = lib.shg_select_workspace_bytes(len(ranks))
    ws = torch.empty(need, dtype=torch.uint8, device=arrays[0].device)
    out = torch.empty(len(ranks), dtype=torch.float64, device=arrays[0].device)
    _lib.check(lib.shg_select_multi_f64(ptrs, n, rk, len(ranks), out.data_ptr(), ws.data_ptr(), need, _stream()),
               'shg_select_multi_f64')
    return out


def flood_stats(image, blurred, very_bright):
    """-> (stats float64 [3] = sum(image), min, max of blurred[blurred < very_bright]; counts int32 [20]) on the GPU."""
    _dev(image, 'image')
    _dev(blurred, 'blurred')
    if image.dtype != torch.float64 or blurred.dtype != torch.float64 or image.shape != blurred.shape:
        raise TypeError('flood_stats needs two dense float64 images of one shape')
    stats = torch.empty(3, dtype=torch.float64, device=image.device)
    counts = torch.empty(20, dtype=torch.int32, device=image.device)
    ws = torch.empty(32, dtype=torch.int64, device=image.device)
    _lib.check(lib.shg_flood_stats_f64(image.contiguous().data_ptr(), blurred.contiguous().data_ptr(), image.numel(),
                                       float(very_bright), stats.data_ptr(), counts.data_ptr(), ws.data_ptr(), _stream()),
               'shg_flood_stats_f64')
    return stats, counts


def flood_stats_lerp(image, blurred, order_stats, gamma):
    """flood_stats with very_bright = lerp of two device-resident order statistics (np.percentile's last step):
    the 99th percentile never travels to the host."""
    _dev(image, 'image')
    _dev(blurred, 'blurred')
    _dev(order_stats, 'order_stats')
    if image.dtype != torch.float64 or blurred.dtype != torch.float64 or image.shape != blurred.shape:
        raise TypeError('flood_stats needs two dense float64 images of one shape')
    if order_stats.dtype != torch.float64 or order_stats.numel() != 2 or not order_stats.is_contiguous():
        raise TypeError('order_stats must be two contiguous float64 values')
    stats = torch.empty(3, dtype=torch.float64, device=image.device)
    counts = torch.empty(20, dtype=torch.int32, device=image.device)
    ws = torch.empty(32, dtype=torch.int64, device=image.device)
    _lib.check(lib.shg_flood_stats_lerp_f64(image.contiguous().data_ptr(), blurred.contiguous().data_ptr(), image.numel(),
                                            order_stats.data_ptr(), float(gamma), stats.data_ptr(), counts.data_ptr(), ws.data_ptr(),
                                            _stream()), 'shg_flood_stats_lerp_f64')
    return stats, counts


def edge_components(low_mask, high_mask, prefetch=16384):
    """Hysteresis + labelling on the GPU.  -> (idx int32 [m], root int32 [m]) HOST arrays: the surviving edge
    pixels in raster order (idx = y*w + x) and the root (smallest linear index) of each pixel's component."""
    _dev(low_mask, 'low_mask')
    if low_mask.dtype != torch.uint8 or high_mask.dtype != torch.uint8 or low_mask.shape != high_mask.shape:
        raise TypeError('edge_components needs two uint8 masks of one shape')
    h, w = low_mask.shape
    n = h * w
    need = lib.shg_edge_components_workspace_bytes(h, w)
    ws = torch.empty(need, dtype=torch.uint8, device=low_mask.device)
    out = torch.empty(2 * n + 1, dtype=torch.int32, device=low_mask.device)      # [count | idx[n] | root[n]]
    _lib.check(lib.shg_edge_components(low_mask.contiguous().data_ptr(), high_mask.contiguous().data_ptr(), h, w,
                                       out[1:].data_ptr(), out[1 + n:].data_ptr(), out.data_ptr(), ws.data_ptr(), need,
                                       _stream()), 'shg_edge_components')
    k = min(prefetch, n)
    head = torch.cat([out[:1 + k], out[1 + n:1 + n + k]]).cpu().numpy()          # one read covers the usual ~1500 points
    m = int(head[0])
    if m <= k:
        return head[1:1 + m], head[1 + k:1 + k + m]
    return out[1:1 + m].cpu().numpy(), out[1 + n:1 + n + m].cpu().numpy()


def limb_prepare(disk, k, ranks4, gamma99):
    """shg_limb_prepare (csrc/limb_fused.hip): the block mean of the uint16 disk, cv2.blur with windows k and 5, their order
    statistics and the flood statistics in five launches.  -> (packed float64 [32] on the GPU: [0..3] order statistics, [4] sum,
    [5] min, [6] max, 20 uint32 counts from packed[8]; keys int32 [sh, sw]: the window sums of blur k; the workspace that
    holds them)."""
    import ctypes
    ptr, h, w, pitch = _img(disk, 'disk', torch.uint16)
    sh, sw = -(-h // 4), -(-w // 4)
    need = lib.shg_limb_prepare_workspace_bytes(sh, sw, int(k))
    if need == 0:
        raise ValueError('limb_prepare: a %d x %d window is outside the fused path' % (k, k))
    ws = torch.empty(need, dtype=torch.uint8, device=disk.device)
    packed = torch.zeros(32, dtype=torch.float64, device=disk.device)
    rk = (ctypes.c_int64 * 4)(*[int(r) for r in ranks4])
    keys_ptr = ctypes.c_void_p()
    _lib.check(lib.shg_limb_prepare(ptr, h, w, pitch, int(k), rk, float(gamma99), packed.data_ptr(), ctypes.byref(keys_ptr), ws.data_ptr(),
                                    need, _stream()), 'shg_limb_prepare')
    off = keys_ptr.value - ws.data_ptr()
    keys = ws[off:off + sh * sw * 4].view(torch.int32).view(sh, sw)
    return packed, keys, ws


def limb_edges(keys, k, flood_thresh, sigma, low, high):
    """shg_limb_edges: canny and the labelling of its low mask from the window sums of the blurred image, three launches.
    -> (idx int32 [m], root int32 [m], high bool [m]) HOST arrays: the LOW-mask pixels in raster order, the smallest linear index
    of each pixel's component, and whether the pixel is in the HIGH mask."""
    import ctypes
    _dev(keys, 'keys')
    sh, sw = keys.shape
    n = sh * sw
    taps, radius = gaussian_taps(sigma)
    taps = np.ascontiguousarray(taps, dtype=np.float64)
    need = lib.shg_limb_edges_workspace_bytes(sh, sw)
    ws = torch.empty(need, dtype=torch.uint8, device=keys.device)
    comp = torch.zeros(2 * n + 1, dtype=torch.int32, device=keys.device)
    _lib.check(lib.shg_limb_edges(keys.data_ptr(), sh, sw, int(k), float(flood_thresh), taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                  radius, float(low), float(high), comp.data_ptr(), ws.data_ptr(), need, _stream()), 'shg_limb_edges')
    host = comp.cpu().numpy()
    m = int(host[0])
    idx, root = host[1:1 + m], host[1 + n:1 + n + m]
    return idx, root & 0x3fffffff, (root >> 30 & 1).astype(bool)


def select_u16(img, ranks, out=None):
    """-> float64 GPU tensor: the ranks[i]-th smallest pixels (0-based) of a uint16 image."""
    import ctypes
    ptr, h, w, pitch = _img(img, 'img', torch.uint16)
    ranks = [int(r) for r in ranks]
    arr = (ctypes.c_int64 * len(ranks))(*ranks)
    need = lib.shg_select_u16_workspace_bytes(len(ranks))
    if need == 0:
        raise ValueError('select_u16 takes 1..8 ranks')
    ws = torch.empty(need, dtype=torch.uint8, device=img.device)
    if out is None:
        out = torch.empty(len(ranks), dtype=torch.float64, device=img.device)
    _lib.check(lib.shg_select_u16(ptr, h, w, pitch, arr, len(ranks), out.data_ptr(), ws.data_ptr(), need, _stream()),
               'shg_select_u16')
    return out


def stream_read_ceiling(buf, mode=0, vecs_per_frame=0, shapes=None, reps=5):
    """Measured HBM read ceiling in GB/s: best of a few launch shapes of a trivial read-only kernel over `buf`
    (a dense GPU tensor, ideally the frame stack itself).  mode 0: contiguous sweep (shapes = (workgroups, unroll));
    mode 2: pass A's frame-walking addresses (shapes = (frame splits, unroll)).  Returns (GB/s, best shape)."""
    _dev(buf, 'buf')
    if shapes is None:
        shapes = (((384, 4), (512, 4), (768, 4), (1024, 4), (768, 2), (1536, 2), (2048, 2), (512, 8)) if mode != 2
                  else ((1, 8), (2, 4), (1, 4), (2, 2), (3, 2), (4, 2), (2, 8), (3, 4)))
    nbytes = buf.numel() * buf.element_size()
    out = torch.zeros(1024, dtype=torch.int32, device=buf.device)
    best = (0.0, None)
    for blocks, unroll in shapes:
        times = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(lib.shg_stream_read_probe(buf.data_ptr(), nbytes, int(mode), int(blocks), int(unroll), int(vecs_per_frame),
                                                 out.data_ptr(), _stream()), 'shg_stream_read_probe')
            b.record()
            b.synchronize()
            times.append(a.elapsed_time(b))
        rate = nbytes / (min(times[1:]) * 1e-3) / 1e9
        if rate > best[0]:
            best = (rate, (blocks, unroll))
    return best
