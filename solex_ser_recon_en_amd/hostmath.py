"""NumPy-facing wrappers of the C++ host control plane (shg_host_*, csrc/hostmath.hip).

The stage composites (stages.py) call these routines from C; this module exposes them one by one so that
the CPU test-suite can hold each against the NumPy / SciPy call it restates, and so that the Python stage
functions that still run step by step (de-vignetted and stubborn scans) share the same arithmetic."""
import ctypes

import numpy as np

from . import _lib
from ._lib import lib

_PD = ctypes.POINTER(ctypes.c_double)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data          # (an int is accepted for a void* argument; data_as builds a ctypes object: twice the time)


def polyfit3(x, y):
    """np.polyfit(x, y, 3), bit for bit (same LAPACK routine on the same matrices)."""
    x, y = _f64(x), _f64(y)
    out = np.empty(4)
    _lib.check(lib.shg_host_polyfit3(_ptr(x), _ptr(y), x.size, _ptr(out)), 'shg_host_polyfit3')
    return out


def detect_bord(row_means):
    row_means = _f64(row_means)
    lb, ub = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.shg_host_detect_bord(_ptr(row_means), row_means.size, ctypes.byref(lb), ctypes.byref(ub)), 'shg_host_detect_bord')
    return lb.value, ub.value


def line_fit(trace_blur, trace_sharp, ih, y1, y2, blur_offset=12):
    """-> (p ascending [4], fit [ih, 4], mask_good bool [y2 - y1])"""
    tb = np.ascontiguousarray(trace_blur, dtype=np.int32)
    ts = np.ascontiguousarray(trace_sharp, dtype=np.int32)
    p, fit = np.empty(4), np.empty((int(ih), 4))
    mask = np.zeros(max(int(y2) - int(y1), 0), dtype=np.uint8)
    _lib.check(lib.shg_host_line_fit(_ptr(tb), _ptr(ts), int(ih), int(y1), int(y2), int(blur_offset), _ptr(p), _ptr(fit),
                                     _ptr(mask)), 'shg_host_line_fit')
    return p, fit, mask.astype(bool)


def column_plan(fit, shifts, ih, iw):
    fit = _f64(fit)
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    ind_l = np.empty((sh.size, int(ih)), dtype=np.int32)
    lw, rw = np.empty(int(ih)), np.empty(int(ih))
    _lib.check(lib.shg_host_column_plan(_ptr(fit), int(ih), int(iw), _ptr(sh), sh.size, _ptr(ind_l), _ptr(lw), _ptr(rw)),
               'shg_host_column_plan')
    return ind_l, lw, rw


def flood_threshold(total, shape, mn, mx, counts):
    c = np.ascontiguousarray(counts, dtype=np.int64)
    out = ctypes.c_double()
    _lib.check(lib.shg_host_flood_threshold(float(total), int(shape[0]), int(shape[1]), float(mn), float(mx), _ptr(c),
                                            ctypes.byref(out)), 'shg_host_flood_threshold')
    return out.value


def limb_points(idx, root, h, w):
    """-> uint8 [m]: 1 where the labelled edge pixel is a limb point"""
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    root = np.ascontiguousarray(root, dtype=np.int32)
    sel = np.zeros(idx.size, dtype=np.uint8)
    n = ctypes.c_int64()
    _lib.check(lib.shg_host_limb_points(_ptr(idx), _ptr(root), idx.size, int(h), int(w), _ptr(sel), ctypes.byref(n)),
               'shg_host_limb_points')
    return sel


def fit_ellipse(points):
    pts = _f64(points)
    center = np.empty(2)
    width, height, phi = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    _lib.check(lib.shg_host_fit_ellipse(_ptr(pts), pts.shape[0], _ptr(center), ctypes.byref(width), ctypes.byref(height),
                                        ctypes.byref(phi)), 'shg_host_fit_ellipse')
    return center, width.value, height.value, phi.value


def correction_matrix(phi, r):
    inv = np.empty((2, 2))
    theta = ctypes.c_double()
    _lib.check(lib.shg_host_correction_matrix(float(phi), float(r), _ptr(inv), ctypes.byref(theta)), 'shg_host_correction_matrix')
    return inv, theta.value


def two_step(points):
    """-> (center (row, col), height, phi, ratio, kept uint8 [n], outline [100, 2])"""
    pts = _f64(points)
    center, outline = np.empty(2), np.empty((100, 2))
    kept = np.zeros(pts.shape[0], dtype=np.uint8)
    height, phi, ratio, n_kept = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(lib.shg_host_two_step(_ptr(pts), pts.shape[0], _ptr(center), ctypes.byref(height), ctypes.byref(phi),
                                     ctypes.byref(ratio), _ptr(kept), ctypes.byref(n_kept), _ptr(outline)), 'shg_host_two_step')
    return center, height.value, phi.value, ratio.value, kept, outline


def warp_geometry(phi, ratio, h, w):
    mat3, inv, origin = np.empty((3, 3)), np.empty((2, 2)), np.empty(2)
    det, theta, oh, ow = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.shg_host_warp_geometry(float(phi), float(ratio), int(h), int(w), _ptr(mat3), _ptr(inv), _ptr(origin),
                                          ctypes.byref(det), ctypes.byref(theta), ctypes.byref(oh), ctypes.byref(ow)),
               'shg_host_warp_geometry')
    return {'mat3': mat3, 'inv_mat': inv, 'origin': origin, 'det': det.value, 'theta': theta.value,
            'out_h': oh.value, 'out_w': ow.value}


def chord_bounds(circle, borders, y1, y2, w):
    count = max(int(y2) - int(y1), 1)
    xa, xb = np.zeros(count, dtype=np.int32), np.zeros(count, dtype=np.int32)
    _lib.check(lib.shg_host_chord_bounds(float(circle[0]), float(circle[1]), float(circle[2]), float(borders[0]),
                                         float(borders[2]), int(y1), int(y2), int(w), _ptr(xa), _ptr(xb)), 'shg_host_chord_bounds')
    return xa, xb


def transversalium_factors(ratios, interior, taps, tapered=True):
    r = _f64(np.atleast_2d(ratios))
    taps = _f64(taps)
    inter = None if interior is None else _f64(np.atleast_2d(interior))
    out = np.empty_like(r)
    _lib.check(lib.shg_host_transversalium_factors(_ptr(r), None if inter is None else _ptr(inter), r.shape[0], r.shape[1],
                                                   _ptr(taps), taps.size, int(bool(tapered)), _ptr(out)),
               'shg_host_transversalium_factors')
    return out


def percentile_plan(n, q):
    lo, hi, g = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
    _lib.check(lib.shg_host_percentile_plan(int(n), float(q), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(g)),
               'shg_host_percentile_plan')
    return lo.value, hi.value, g.value
