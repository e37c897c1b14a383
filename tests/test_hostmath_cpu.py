"""The C++ host control plane (csrc/hostmath.hip, shg_host_*) against NumPy / SciPy and the oracle.

These functions take host pointers only, so the whole file runs without a GPU.  The line fit must be
BIT-IDENTICAL to NumPy's (the raw disks are exact only if `fit` is).  tests/numpy_ref.py holds the NumPy / SciPy
statements the C++ is compared with."""
import ctypes
import os
import math

import numpy as np
import pytest
from numpy.polynomial.polynomial import polyval

from oracle import shg_oracle as orc
from solex_ser_recon_en_amd import _lib, hostmath, order_stats, solex_util
from tests import numpy_ref

lib = _lib.lib


def test_numpy_lapack_is_bound():
    assert _lib.LAPACK_PATH is not None and lib.shg_host_lapack_bound() == 1


@pytest.mark.parametrize('seed', range(6))
def test_polyfit3_is_numpy_polyfit_bit_for_bit(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(4, 4000))
    x = np.sort(rng.choice(5000, n, replace=False)).astype(np.float64) if seed % 2 else rng.normal(0, 50, n)
    y = 100 + 3e-6 * (x - 1000) ** 2 + rng.normal(0, 0.5, n)
    if seed % 3 == 0:
        y = np.round(y)
    np.testing.assert_array_equal(hostmath.polyfit3(x, y), np.polyfit(x, y, 3))


def test_polyfit3_qr_fallback_agrees_to_rounding():
    rng = np.random.default_rng(1)
    x = np.arange(100, 1900, dtype=np.float64)
    y = np.round(100 + 6e-6 * (x - 1000) ** 2 + rng.normal(0, 0.4, x.size))
    lib.shg_host_bind_lapack(None)
    try:
        got = hostmath.polyfit3(x, y)
    finally:
        _lib._bind_numpy_lapack()
    want = np.polyfit(x, y, 3)
    assert lib.shg_host_lapack_bound() == 1
    np.testing.assert_allclose(np.polyval(got, x), np.polyval(want, x), rtol=0, atol=1e-9)


@pytest.mark.parametrize('seed', range(8))
def test_detect_bord(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 3000))
    ymean = rng.random(n) * 100
    lo, hi = sorted(rng.integers(0, n, 2))
    ymean[lo:hi + 1] += 5000
    where = ymean > np.median(ymean) / 5
    want = (int(np.argmax(where)), int(n - 1 - np.argmax(np.flip(where))))
    assert hostmath.detect_bord(ymean) == want
    assert hostmath.detect_bord(np.zeros(7)) == (0, 6)


def numpy_line_fit(trace_blur, trace_sharp, ih, y1, y2):
    """compute_mean_return_fit's host arithmetic (solex_util.py:231-259), NumPy itself."""
    min_intensity = 12 + trace_blur.astype(np.int64)
    sharp = trace_sharp.astype(np.int64)
    rows = np.arange(y1, y2)
    rows_d = np.asarray(rows, dtype='d')
    p = np.flip(np.asarray(np.polyfit(rows, min_intensity[y1:y2], 3), dtype='d'))
    delta = polyval(rows_d, p) - min_intensity[y1:y2]
    keep = np.abs(delta / np.std(delta)) < 3
    p = np.flip(np.asarray(np.polyfit(rows[keep], min_intensity[y1:y2][keep], 3), dtype='d'))
    delta_sharp = polyval(rows_d, p) - sharp[y1:y2]
    values, counts = np.unique(np.around(delta_sharp, 1), return_counts=True)
    ind = np.argpartition(-counts, kth=2)[:2]
    mask = np.abs(delta_sharp - values[ind[0]]) < 5
    p = np.flip(np.asarray(np.polyfit(rows[mask], sharp[y1:y2][mask], 3), dtype='d'))
    curve = polyval(np.asarray(np.arange(ih), dtype='d'), p)
    return p, np.stack([np.floor(curve), curve - np.floor(curve), np.arange(ih, dtype='d'), curve], axis=1), mask


@pytest.mark.parametrize('seed', range(12))
def test_line_fit_is_numpy_bit_for_bit(seed):
    rng = np.random.default_rng(100 + seed)
    ih = int(rng.integers(300, 4200))
    iw = int(rng.integers(60, 300))
    y = np.arange(ih)
    curve = iw / 2 + rng.uniform(2e-6, 2e-5) * (y - ih / 2) ** 2 + rng.uniform(-0.01, 0.01) * (y - ih / 2)
    sigma = [0.3, 0.8, 2.0][seed % 3]
    sharp = np.clip(np.rint(curve + rng.normal(0, sigma, ih)), 0, iw - 1).astype(np.int32)
    blur = np.clip(np.rint(curve + rng.normal(0, 0.3, ih)), 12, iw - 14).astype(np.int32) - 12
    outl = rng.choice(ih, ih // 50, replace=False)
    sharp[outl] = rng.integers(0, iw, outl.size)
    blur[outl[:5]] = rng.integers(0, iw - 26, 5)
    y1, y2 = int(0.06 * ih), int(0.94 * ih)
    p, fit, mask = hostmath.line_fit(blur, sharp, ih, y1, y2)
    wp, wfit, wmask = numpy_line_fit(blur, sharp, ih, y1, y2)
    np.testing.assert_array_equal(p, wp)
    np.testing.assert_array_equal(fit, wfit)
    np.testing.assert_array_equal(mask, wmask)


def test_line_fit_mode_pick_is_numpys_argpartition():
    """values[np.argpartition(-counts, kth=2)[:2][0]] is one of the two most frequent values, which one being up to
    NumPy's selection kernel: the binding registers NumPy's own argpartition for that decision, and without a picker the
    library takes the first most frequent value."""
    rng = np.random.default_rng(5)
    ih, y1, y2 = 600, 30, 570
    y = np.arange(ih)
    for trial in range(40):
        curve = 60 + 1e-5 * (y - 300) ** 2
        sharp = np.rint(curve + rng.normal(0, 0.15 + 0.02 * trial, ih)).astype(np.int32)     # few distinct residuals: ties
        blur = np.rint(curve).astype(np.int32) - 12
        p, fit, mask = hostmath.line_fit(blur, sharp, ih, y1, y2)
        wp, wfit, wmask = numpy_line_fit(blur, sharp, ih, y1, y2)
        np.testing.assert_array_equal(fit, wfit)
        np.testing.assert_array_equal(mask, wmask)
    lib.shg_host_set_mode_pick(None)
    try:
        p0, fit0, _ = hostmath.line_fit(blur, sharp, ih, y1, y2)
    finally:
        lib.shg_host_set_mode_pick(ctypes.cast(_lib._numpy_mode_pick, ctypes.c_void_p))
    np.testing.assert_allclose(fit0[:, 3], wfit[:, 3], atol=0.05)


def test_line_fit_failures_raise_numpys_exceptions():
    ih = 400
    flat = np.zeros(ih, dtype=np.int32)
    with pytest.raises(ValueError, match=r'kth\(=2\) out of bounds'):
        hostmath.line_fit(flat, flat, ih, 20, 380)
    with pytest.raises(TypeError, match='expected non-empty vector'):
        hostmath.line_fit(flat, flat, ih, 50, 50)


@pytest.mark.parametrize('tag', ['u16_rot', 'u8_norot'])
def test_line_fit_on_the_golden_reference_run(golden, tag):
    """g8: the reference's own compute_mean_return_fit (cv2.blur := oracle box blur, captured under NumPy 1.26).
    From the same traces the C++ fit equals this host's NumPy bit for bit, and the reference's to LAPACK-build noise."""
    g = golden('g8_fit_shim')
    mean, mx = g[tag + '_mean'], g[tag + '_max']
    fit, y1, y2, p, aux = orc.line_fit(mean, mx)
    tb = (aux['min_intensity'] - 12).astype(np.int32)
    gp, gfit, gmask = hostmath.line_fit(tb, aux['sharp'].astype(np.int32), mean.shape[0], y1, y2)
    np.testing.assert_array_equal(gfit, fit)
    np.testing.assert_array_equal(gp, np.asarray(p))
    np.testing.assert_array_equal(gmask, aux['mask_good'])
    np.testing.assert_allclose(gfit[:, 3], g[tag + '_fit'][:, 3], rtol=0, atol=1e-9)
    lb, ub = hostmath.detect_bord(np.mean(orc.box_blur_u16(mx, 5, 5), axis=1))
    clip = int((ub - lb) * 0.05)
    assert (min(mx.shape[0] - 1, lb + clip), max(0, ub - clip)) == tuple(g[tag + '_y'])


@pytest.mark.parametrize('seed', range(4))
def test_column_plan(seed):
    rng = np.random.default_rng(seed)
    ih, iw = 500, 40
    curve = rng.uniform(-3, iw + 3, ih)
    fit = np.stack([np.floor(curve), curve - np.floor(curve), np.arange(ih, dtype='d'), curve], axis=1)
    shifts = [10, 0, -7, 3]
    ind_l, lw, rw = hostmath.column_plan(fit, shifts, ih, iw)
    w_ind, w_lw, w_rw = numpy_ref.column_plan(fit, shifts, ih, iw)
    np.testing.assert_array_equal(ind_l, w_ind)
    np.testing.assert_array_equal(lw, w_lw)
    np.testing.assert_array_equal(rw, w_rw)


@pytest.mark.parametrize('seed', range(10))
def test_flood_threshold(seed):
    rng = np.random.default_rng(seed)
    counts = np.sort(rng.integers(0, 5000, 20))[::-1].copy()
    rng.shuffle(counts[5:])
    counts[int(rng.integers(8, 16))] += 9000
    total, mn, mx = rng.uniform(1e3, 1e5), rng.uniform(0, 0.01), rng.uniform(0.2, 0.9)
    got = hostmath.flood_threshold(total, (300, 320), mn, mx, counts)
    assert got == numpy_ref.flood_threshold(total, (300, 320), mn, mx, counts)
    const = np.zeros(20, dtype=np.int64)                                   # constant data: np.histogram's +-0.5 range
    const[10] = 100
    assert hostmath.flood_threshold(5.0, (10, 10), 0.25, 0.25, const) == numpy_ref.flood_threshold(5.0, (10, 10), 0.25, 0.25, const)


def ring_points(rng, h, w, cy, cx, ay, ax, gap=None, blobs=0):
    """Edge pixels of an ellipse outline (raster order) with component roots, like shg_edge_components' output."""
    from scipy import ndimage
    img = np.zeros((h, w), dtype=bool)
    t = np.linspace(0, 2 * np.pi, 4000, endpoint=False)
    yy = np.rint(cy + ay * np.sin(t)).astype(int)
    xx = np.rint(cx + ax * np.cos(t)).astype(int)
    ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
    img[yy[ok], xx[ok]] = True
    if gap is not None:
        img[:, gap[0]:gap[1]] = False
    for _ in range(blobs):
        by, bx = int(rng.integers(2, h - 4)), int(rng.integers(2, w - 4))
        img[by:by + 2, bx:bx + 3] = True
    lab, nf = ndimage.label(img, np.ones((3, 3)))
    idx = np.flatnonzero(img).astype(np.int32)
    labs = lab.ravel()[idx]
    first = np.full(nf + 1, -1, dtype=np.int64)
    for i, l in zip(idx, labs):
        if first[l] < 0:
            first[l] = i
    return idx, first[labs].astype(np.int32), labs


@pytest.mark.parametrize('seed', range(8))
def test_limb_points_match_the_numpy_qhull_selection(seed):
    rng = np.random.default_rng(seed)
    h, w = 500, 520
    gap = (int(rng.integers(150, 250)), int(rng.integers(255, 300))) if seed % 2 else None
    idx, root, labs = ring_points(rng, h, w, 250 + rng.uniform(-20, 20), 260 + rng.uniform(-20, 20),
                                  rng.uniform(150, 220), rng.uniform(150, 230), gap, blobs=seed % 4)
    pts = np.stack([idx // w, idx % w], axis=1).astype(np.int64)
    lab, nf = numpy_ref.labels_from_roots(root)
    np.testing.assert_array_equal(lab, labs)
    want = numpy_ref.limb_points(pts, lab, nf, h)
    sel = hostmath.limb_points(idx, root, h, w)
    np.testing.assert_array_equal(pts[sel.astype(bool)].astype(float), want)


def test_limb_points_failures():
    from scipy.spatial import QhullError
    line = (np.arange(30) * 40 + 7).astype(np.int32)                   # one column: collinear
    with pytest.raises(QhullError):
        hostmath.limb_points(line, np.full(30, 7, dtype=np.int32), 40, 40)
    with pytest.raises(RuntimeError, match='could not find any edges'):
        hostmath.limb_points(np.zeros(0, np.int32), np.zeros(0, np.int32), 40, 40)


def one_ulp(a, b):
    return abs(a - b) <= np.spacing(abs(b))


@pytest.mark.parametrize('seed', range(40))
def test_two_step_is_the_numpy_fit_bit_for_bit(seed):
    """shg_host_two_step (NumPy's own dsyrk / dgemm / dgemv / dgesv / dgeev, called the way NumPy's matmul / inv / eig call
    them) against the oracle's lsq-ellipse restatement in NumPy: centre, height and ratio bit-identical, phi within the one
    ulp np.arctan and libm's atan may differ by.  phi and ratio steer every sample position of the warp."""
    from oracle import limb_oracle
    rng = np.random.default_rng(seed)
    n = int(rng.integers(20, 3000))
    t = rng.uniform(0, 2 * np.pi, n)
    a, b = rng.uniform(100, 2900), rng.uniform(100, 2900)
    phi = rng.uniform(-0.7, 0.7)
    cy, cx = rng.uniform(300, 3200), rng.uniform(300, 3200)
    r = cy + a * np.cos(t) * np.cos(phi) - b * np.sin(t) * np.sin(phi) + rng.normal(0, 0.7, n)
    c = cx + a * np.cos(t) * np.sin(phi) + b * np.sin(t) * np.cos(phi) + rng.normal(0, 0.7, n)
    pts = np.stack([np.rint(r / 4) * 4, np.rint(c / 4) * 4], axis=1)
    pts[:n // 20] += rng.normal(0, 30, (n // 20, 2))                    # outliers for two_step to reject
    got = hostmath.two_step(pts)
    want = limb_oracle.two_step(pts)
    np.testing.assert_array_equal(got[0], want[0])
    assert got[1] == want[1] and got[3] == want[3]
    assert one_ulp(got[2], want[2])
    np.testing.assert_array_equal(pts[got[4].astype(bool)], want[4])
    np.testing.assert_allclose(got[5], want[5], rtol=1e-12, atol=1e-9)
    center, width, height, ph = hostmath.fit_ellipse(pts)
    reg = limb_oracle.LsqEllipse().fit(pts)
    wc, ww, wh, wp = reg.as_parameters()
    assert tuple(center) == tuple(wc) and (width, height) == (ww, wh) and one_ulp(ph, wp)


@pytest.fixture
def without_numpys_blas():
    """The host whose NumPy is not a pip wheel: nothing bound, the library on its built-in routines."""
    lib.shg_host_bind_blas(None, None, None, None, None)
    lib.shg_host_bind_lapack(None)
    assert lib.shg_host_blas_bound() == 0 and lib.shg_host_lapack_bound() == 0
    yield
    _lib._bind_numpy_lapack()
    assert lib.shg_host_blas_bound() == 1 and lib.shg_host_lapack_bound() == 1


@pytest.mark.parametrize('seed', range(12))
def test_the_builtin_routines_carry_the_control_plane_when_numpys_are_not_found(seed, without_numpys_blas):
    """No OpenBLAS of a NumPy wheel to bind (conda / MKL / a distribution's NumPy): the limb geometry, the warp geometry and
    the line fit still work, on the library's own plain-loop algebra and Householder least squares, and agree with NumPy's
    to rounding (not bit for bit: _lib warns about that at import)."""
    from oracle import limb_oracle
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(40, 3000))
    t = rng.uniform(0, 2 * np.pi, n)
    a, b = rng.uniform(100, 2900), rng.uniform(100, 2900)
    phi = rng.uniform(-0.7, 0.7)
    cy, cx = rng.uniform(300, 3200), rng.uniform(300, 3200)
    r = cy + a * np.cos(t) * np.cos(phi) - b * np.sin(t) * np.sin(phi) + rng.normal(0, 0.7, n)
    c = cx + a * np.cos(t) * np.sin(phi) + b * np.sin(t) * np.cos(phi) + rng.normal(0, 0.7, n)
    pts = np.stack([np.rint(r / 4) * 4, np.rint(c / 4) * 4], axis=1)
    pts[:n // 20] += rng.normal(0, 30, (n // 20, 2))
    got = hostmath.two_step(pts)
    want = limb_oracle.two_step(pts)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-9)
    np.testing.assert_allclose([got[1], got[2], got[3]], [want[1], want[2], want[3]], rtol=1e-8, atol=1e-11)
    np.testing.assert_array_equal(pts[got[4].astype(bool)], want[4])
    g = hostmath.warp_geometry(want[2], want[3], 2000, 1800)
    inv, theta = hostmath.correction_matrix(want[2], want[3])
    stretch = np.array([[np.cos(want[2]), np.sin(want[2])], [-np.sin(want[2]), np.cos(want[2])]]) @ np.diag([want[3], 1.0]) @ \
        np.array([[np.cos(want[2]), -np.sin(want[2])], [np.sin(want[2]), np.cos(want[2])]])
    th = np.arctan(stretch[1, 0] / stretch[0, 0])
    corr = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]]) @ stretch
    corr[1, 0] = 0
    corr /= corr[1, 1]
    np.testing.assert_allclose(inv, np.linalg.inv(corr), rtol=1e-12, atol=1e-15)
    assert abs(theta - th) < 1e-14 and g['out_h'] == 2000
    # the line fit on the Householder least squares
    y = np.arange(60, 1900)
    trace = 98.0 + 2e-3 * (y - 1000) + 6e-6 * (y - 1000) ** 2 + rng.normal(0, 0.4, y.size)
    tb = np.rint(trace).astype(np.int32) - 12
    ts = np.rint(trace + rng.normal(0, 0.3, y.size)).astype(np.int32)
    full_b, full_s = np.zeros(2000, np.int32), np.zeros(2000, np.int32)
    full_b[60:1900], full_s[60:1900] = tb, ts
    p, fit, mask = hostmath.line_fit(full_b, full_s, 2000, 60, 1900)
    assert mask.sum() > 1500 and abs(np.polyval(p[::-1], 1000.0) - 98.0) < 0.5


def test_ellipse_fit_recovers_an_analytic_ellipse():
    t = np.linspace(0, 2 * np.pi, 721)[:-1]
    a, b, phi, cy, cx = 800.0, 640.0, 0.2, 1000.0, 1100.0
    pts = np.stack([cy + a * np.cos(t) * np.cos(phi) - b * np.sin(t) * np.sin(phi),
                    cx + a * np.cos(t) * np.sin(phi) + b * np.sin(t) * np.cos(phi)], axis=1)
    center, width, height, ph = hostmath.fit_ellipse(pts)
    np.testing.assert_allclose(center, [cy, cx], rtol=1e-9)
    np.testing.assert_allclose(sorted([width, height]), [b, a], rtol=1e-9)


@pytest.mark.parametrize('phi,ratio', [(0.0, 1.0), (0.1, 1.08), (-0.2, 0.93), (0.7, 1.3), (1e-9, 1.0000001), (0.0312, 1.0173)])
def test_correction_matrix_and_warp_geometry_are_numpys(phi, ratio):
    from solex_ser_recon_en_amd.ellipse_to_circle import get_correction_matrix
    inv, theta = hostmath.correction_matrix(phi, ratio)
    w_inv, w_theta = get_correction_matrix(phi, ratio)
    np.testing.assert_array_equal(inv, w_inv)
    assert one_ulp(theta, w_theta) or theta == w_theta
    # correct_image's geometry (ellipse_to_circle.py:100-114), NumPy itself
    h, w = 700, 640
    mat = w_inv
    mat3 = np.zeros((3, 3))
    mat3[:2, :2] = mat
    mat3[2, 2] = 1
    corners = np.array([[0, 0], [0, h], [w, 0], [w, h]])
    inv_mat = np.linalg.inv(mat)
    new_corners = (inv_mat @ corners.T).T
    origin = np.array([np.min(new_corners[:, 0]), np.min(new_corners[:, 1])])
    out_h = int(np.ceil(np.max(new_corners[:, 1]) - np.min(new_corners[:, 1])))
    out_w = int(np.ceil(np.max(new_corners[:, 0]) - np.min(new_corners[:, 0])))
    mat3 = mat3 @ np.array([[1, 0, origin[0]], [0, 1, origin[1]], [0, 0, 1]])
    g = hostmath.warp_geometry(phi, ratio, h, w)
    assert (g['out_h'], g['out_w']) == (out_h, out_w)
    np.testing.assert_array_equal(g['inv_mat'], inv_mat)
    np.testing.assert_allclose(g['origin'], origin, rtol=4e-16, atol=0)
    np.testing.assert_allclose(g['mat3'], mat3, rtol=4e-16, atol=1e-300)      # what shg_warp_rows_u16 samples with
    np.testing.assert_allclose(g['det'], np.linalg.det(mat), rtol=1e-15)


def test_correction_matrix_table_is_the_references(golden):
    g = golden('g7_matrix')                              # the reference's own get_correction_matrix
    for (phi, r), want, theta in zip(g['params'], g['mats'], g['thetas']):
        inv, th = hostmath.correction_matrix(float(phi), float(r))
        np.testing.assert_allclose(inv, want, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(th, theta, rtol=1e-12, atol=1e-15)


def test_limb_geometry_borders_and_circle():
    """shg_host_limb_geometry: circle of the corrected image and borders of the kept points, against ellipse_to_circle.py:
    303-314 in NumPy."""
    from oracle import limb_oracle
    from solex_ser_recon_en_amd.ellipse_to_circle import get_correction_matrix
    rng = np.random.default_rng(11)
    t = rng.uniform(0, 2 * np.pi, 1400)
    pts = np.stack([np.rint((1000 + 880 * np.sin(t) + rng.normal(0, 0.6, t.size)) / 4) * 4,
                    np.rint((1050 + 930 * np.cos(t) + rng.normal(0, 0.6, t.size)) / 4) * 4], axis=1)
    h, w = 2000, 2100
    geom = np.empty(16)
    dims = np.zeros(2, dtype=np.int64)
    kept = np.zeros(len(pts), dtype=np.uint8)
    n_kept = ctypes.c_int64()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)          # noqa: E731
    _lib.check(lib.shg_host_limb_geometry(P(pts), len(pts), h, w, P(geom), P(dims), P(kept), ctypes.byref(n_kept), None), 'geom')
    center, height, phi, ratio, X_f, _ = limb_oracle.two_step(pts)
    center = np.array([center[1], center[0]])
    mat, theta = get_correction_matrix(phi, ratio)
    inv_mat = np.linalg.inv(mat)
    corners = np.array([[0, 0], [0, h], [w, 0], [w, h]])
    nc = (inv_mat @ corners.T).T
    origin = np.array([nc[:, 0].min(), nc[:, 1].min()])
    mat3 = np.zeros((3, 3)); mat3[:2, :2] = mat; mat3[2, 2] = 1
    mat3 = mat3 @ np.array([[1, 0, origin[0]], [0, 1, origin[1]], [0, 0, 1]])
    new_center = (inv_mat @ center.T).T - origin
    new_radius = height * np.sqrt(np.abs(ratio / np.linalg.det(mat)))
    X_f3 = np.ones((X_f.shape[0], 3)); X_f3[:, 1] = X_f[:, 0]; X_f3[:, 0] = X_f[:, 1]
    tr = (np.linalg.inv(mat3) @ X_f3.T).T
    borders = [tr[:, 0].min(), tr[:, 1].min(), tr[:, 0].max(), tr[:, 1].max()]
    assert n_kept.value == len(X_f)
    np.testing.assert_allclose(geom[:5], [center[0], center[1], height, phi, ratio], rtol=3e-16)
    np.testing.assert_allclose(geom[5:7], new_center, rtol=1e-15)
    np.testing.assert_allclose(geom[7], new_radius, rtol=1e-15)
    np.testing.assert_allclose(geom[8:12], borders, rtol=1e-15)
    np.testing.assert_allclose(geom[12:15], mat3[0], rtol=4e-16, atol=1e-300)
    assert tuple(dims) == (int(np.ceil(nc[:, 1].max() - nc[:, 1].min())), int(np.ceil(nc[:, 0].max() - nc[:, 0].min())))


@pytest.mark.parametrize('seed', range(6))
def test_chord_bounds(seed):
    rng = np.random.default_rng(seed)
    w = 2100
    circle = (rng.uniform(900, 1100), rng.uniform(950, 1050), rng.uniform(700, 900))
    if seed == 0:
        circle = (1000.0, 1000.0, 845.0)                                   # Pythagorean rows: exact integer roots
    borders = [rng.uniform(0, 300), 0, rng.uniform(1800, 2099), 0]
    y1 = math.ceil(circle[1] - circle[2]) + 3
    y2 = math.floor(circle[1] + circle[2]) - 3
    xa, xb = hostmath.chord_bounds(circle, borders, y1, y2, w)
    wa, wb = numpy_ref.chord_bounds(circle, borders, y1, y2, w)
    np.testing.assert_array_equal(xa, wa)
    np.testing.assert_array_equal(xb, wb)
    # the reference's own loop (solex_util.py:389-391)
    for y in range(y1 + 1, y2, 37):
        dx = math.floor((circle[2] ** 2 - (y - circle[1]) ** 2) ** 0.5)
        s = slice(math.ceil(max(circle[0] - dx, borders[0])), math.floor(min(circle[0] + dx, borders[2]))).indices(w)
        assert (xa[y - y1], xb[y - y1]) == (s[0], max(s[0], s[1]))
    with pytest.raises(TypeError):
        hostmath.chord_bounds((1000.0, 1000.0, 10.0), [0, 0, 2000, 0], 900, 1100, w)


def test_chord_bounds_roots_near_whole_numbers():
    """shg_host_chord_bounds takes sqrt and asks libm's pow (Python's ** 0.5) only where the root lies within a few ulp of a whole
    number, where the two could floor differently.  Circles with whole-number and half-integer centres and radii make every row's
    radicand a whole number or a multiple of 1/4 -- thousands of exact and nearly exact roots -- and radii one ulp either side of
    such values put radicands one ulp from perfect squares: every row must give what the reference's expression gives."""
    rng = np.random.default_rng(3)
    w = 4000
    for trial in range(300):
        r = float(rng.integers(50, 1900)) + (0.5 if trial % 3 == 1 else 0.0)
        cy = float(rng.integers(1900, 2100)) + (0.5 if trial % 5 == 2 else 0.0)
        cx = float(rng.integers(1900, 2100))
        if trial % 7 == 3:
            r = np.nextafter(r, np.inf if trial % 2 else -np.inf)
        y1 = math.ceil(cy - r) + 1
        y2 = math.floor(cy + r) - 1
        if y2 - y1 < 3:
            continue
        xa, xb = hostmath.chord_bounds((cx, cy, float(r)), [0.0, 0, float(w - 1), 0], y1, y2, w)
        for y in range(y1 + 1, y2):
            dx = math.floor((float(r) ** 2 - (y - cy) ** 2) ** 0.5)
            s = slice(math.ceil(max(cx - dx, 0.0)), math.floor(min(cx + dx, float(w - 1)))).indices(w)
            assert (xa[y - y1], xb[y - y1]) == (s[0], max(s[0], s[1])), (trial, r, cy, y)


@pytest.mark.parametrize('k,n,strength', [(1, 1780, 301), (3, 1200, 301), (2, 200, 301), (1, 45, 301), (2, 900, 41)])
def test_transversalium_factors(k, n, strength):
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(n)
    ratios = rng.normal(0, 2e-3, (k, n)) + 1e-3 * np.sin(np.arange(n) / 50.0)
    window = solex_util.savgol_window(n, strength)
    taps = solex_util.savgol_taps(window)
    interior = correlate1d(ratios, taps[::-1], axis=-1, mode='constant')
    want = numpy_ref.transversalium_factors(ratios, strength)
    for inter in (interior, None):
        got = hostmath.transversalium_factors(ratios, inter, taps, tapered=True)
        # exp() may differ by an ulp between NumPy's SIMD loop and libm; everything before it is bit-identical
        np.testing.assert_allclose(got, want, rtol=3e-16, atol=0)
    raw = hostmath.transversalium_factors(ratios, interior, taps, tapered=False)
    np.testing.assert_allclose(raw, numpy_ref.transversalium_factors(ratios, strength, tapered=False), rtol=3e-16)
    with pytest.raises(ValueError, match='window_length'):
        hostmath.transversalium_factors(ratios[:, :5], None, solex_util.savgol_taps(7))


def test_trend_before_exp_is_bit_identical():
    """The Savitzky-Golay trend itself (interior + LAPACK edge fits) has no transcendental in it: exact."""
    from scipy.signal import savgol_filter
    rng = np.random.default_rng(2)
    r = rng.normal(0, 1e-3, 1500)
    taps = solex_util.savgol_taps(301)
    f = hostmath.transversalium_factors(r[None], None, taps, tapered=False)[0]
    det = r - savgol_filter(r, 301, 3)
    want = np.exp(-np.cumsum(det - np.mean(det)))
    np.testing.assert_allclose(f, want, rtol=3e-16)
    # log of the factors returns the cumulative sums to within exp/log rounding
    np.testing.assert_allclose(-np.log(f), np.cumsum(det - np.mean(det)), rtol=0, atol=1e-15)


@pytest.mark.parametrize('n,q', [(4000000, 99.9999), (4000000, 10), (250000, 99), (7, 50), (1, 85), (2098 * 2000, 99.9999)])
def test_percentile_plan(n, q):
    lo, hi, gamma = hostmath.percentile_plan(n, q)
    wlo, whi, _ = order_stats.lerp_order_stats(n, q)
    assert (lo, hi) == (wlo, whi) and gamma == order_stats.lerp_gamma(n, q)
    rng = np.random.default_rng(0)
    a, b = sorted(rng.random(2))
    assert lib.shg_host_lerp(a, b, gamma) == order_stats.lerp_order_stats(n, q)[2](a, b)


def test_fast_log_stays_below_one_ulp(tmp_path):
    """csrc/fast_log.h (the logarithm of k_rowpair_stats) compiled with the host compiler: below 1 ulp against logl over the
    quotients of 16-bit pixel pairs and over random normal doubles, exact for 1, and equal to the host libm's log in all but a
    few percent of the cases (where the two differ in the last bit, as two libm's do)."""
    import shutil
    import subprocess
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('no g++')
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / 'fast_log_check')
    subprocess.run([gxx, '-O2', '-ffp-contract=off', '-I', os.path.join(here, '..', 'solex_ser_recon_en_amd', 'csrc'),
                    os.path.join(here, 'c_abi', 'fast_log_check.cpp'), '-o', exe], check=True)
    pairs, every, differ, ratio, ratio_differs = (float(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split())
    assert pairs < 1.0 and every < 1.0, (pairs, every)
    assert differ < 0.08, differ
    # log_ratio_u16(a, b) (one reciprocal for the quotient and for the logarithm's own division): below 1 ulp too, and the very
    # double log_normal(a / b) gives, but for a few pairs in a million
    assert ratio < 1.0 and ratio_differs < 1e-4, (ratio, ratio_differs)


def test_ellipse_fit_agrees_with_scikit_images_independent_implementation(golden):
    """lsq-ellipse (the reference's `from ellipse import LsqEllipse`, ellipse_to_circle.py:53-59) is absent from /root/reference
    and from this image, so the fit is a restatement of Halir & Flusser's direct least squares -- which nothing but itself pinned.
    g16 holds the same published method as scikit-image 0.18.3 implements it (skimage.measure.EllipseModel: its own scaling,
    eigen-solve and conversion to centre / axes / angle), captured under the interpreter that has it, on noisy, partly open
    ellipses of the shapes a limb takes: the oracle's LsqEllipse and the C++ control plane's fit (hostmath.fit_ellipse, what every
    scan runs) must describe the same ellipse -- centre, both semi-axes and the major axis' direction."""
    from oracle import limb_oracle as limb
    g = golden('g16_ellipse_skimage')

    def canonical(xc, yc, a, b, theta):
        if a < b:
            a, b, theta = b, a, theta + np.pi / 2
        return np.array([xc, yc, a, b]), float(np.mod(theta, np.pi))

    for i in range(int(g['n_cases'])):
        pts = g['points%d' % i]
        want, want_angle = canonical(*g['params%d' % i])
        fits = []
        center, width, height, phi = limb.LsqEllipse().fit(pts).as_parameters()
        fits.append((float(center[0]), float(center[1]), float(np.real(width)), float(np.real(height)), float(np.real(phi))))
        c2, w2, h2, p2 = hostmath.fit_ellipse(pts)
        fits.append((float(c2[0]), float(c2[1]), float(w2), float(h2), float(p2)))
        for got in fits:
            vals, angle = canonical(*got)
            np.testing.assert_allclose(vals, want, rtol=1e-11, atol=0)              # (measured: 1e-15 .. 7e-14)
            roundness = (want[2] - want[3]) / want[2]                 # a nearly round ellipse has no direction to speak of
            d = abs(angle - want_angle)
            assert min(d, np.pi - d) < 1e-11 / max(roundness, 1e-3), (i, angle, want_angle)     # (measured: 2e-14 .. 2e-12)
        # and the two of this repo agree far more closely with each other than with a third party's numerics
        np.testing.assert_allclose(fits[0][:4], fits[1][:4], rtol=1e-11)
