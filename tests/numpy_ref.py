"""NumPy / SciPy statements of the host control plane, written the way the reference writes them (file:line cited).

Test infrastructure only: the product computes these steps in C++ (csrc/hostmath.hip, csrc/stages.hip); the tests hold
the C++ against the functions below, which call the very library routines the reference calls."""
import math

import numpy as np
from numpy.polynomial import Polynomial
from scipy.signal import savgol_filter
from scipy.spatial import ConvexHull

NUM_REG = 2          # ellipse_to_circle.py:31


def column_plan(fit, shifts, ih, iw):
    """Clamped left sample column per shift and the (unclamped) bilinear weights, solex_util.py:113-123."""
    fit = np.asarray(fit)
    ind_l = np.empty((len(shifts), ih), dtype=np.int32)
    for i, shift in enumerate(shifts):
        col = (fit[:, 0] + np.ones(ih) * shift).astype(int)
        col[col < 0] = 0
        col[col > iw - 2] = iw - 2
        ind_l[i] = col
    left_weights = np.ones(ih) - fit[:, 1]
    right_weights = np.ones(ih) - left_weights
    return ind_l, left_weights, right_weights


def chord_bounds(circle, borders, y1, y2, w):
    """Column slice [a, b) of every row y1 .. y2-1 (entry 0 unused), solex_util.py:389-391, with NumPy's slice rules."""
    count = max(y2 - y1, 1)
    xa = np.zeros(count, dtype=np.int32)
    xb = np.zeros(count, dtype=np.int32)
    for y in range(y1 + 1, y2):
        v = circle[2] ** 2 - (y - circle[1]) ** 2
        if v < 0:
            raise TypeError('complex chord length')                     # math.floor(complex) in the reference
        dx = math.floor(v ** 0.5)
        a, b, _ = slice(math.ceil(max(circle[0] - dx, borders[0])), math.floor(min(circle[0] + dx, borders[2]))).indices(w)
        xa[y - y1], xb[y - y1] = a, max(a, b)
    return xa, xb


def transversalium_factors(y_ratios_r, trans_strength, tapered=True):
    """Row correction factors from the row-pair log-ratios, solex_util.py:400-404 and 456-472, row by row."""
    rows = np.atleast_2d(np.asarray(y_ratios_r, dtype=np.float64))
    out = []
    for r in rows:
        trend = savgol_filter(r, min(trans_strength, len(r) // 2 * 2 - 1), 3)
        detrended = r - trend
        detrended = detrended - np.mean(detrended)
        correction = np.exp(-np.cumsum(detrended))
        if not tapered:
            out.append(correction)
            continue
        a, n = 0.05, correction.shape[0]

        def t(x):                                                        # :460-468
            if 0 <= x < a * n / 2:
                return 1 / 2 * (1 - math.cos(2 * math.pi * x / (a * n)))
            if a * n / 2 <= x <= n / 2:
                return 1
            return t(n - x)
        taper = np.array([t(x) for x in range(n)])
        out.append(np.ones(n) + (correction - np.ones(n)) * taper)
    out = np.array(out)
    return out if np.ndim(y_ratios_r) == 2 else out[0]


def flood_threshold(total, shape, mn, mx, counts):
    """thresh3 of get_flood_image (ellipse_to_circle.py:169-225) from the reduced statistics: total = np.sum(image);
    over data = blurred[blurred < very_bright]: mn, mx, counts = np.histogram(data, bins=20)[0]."""
    h, w = shape
    thresh = 0.9 * total / (h * w)
    if mn == mx:
        mn, mx = mn - 0.5, mx + 0.5
    bins = np.linspace(mn, mx, 21)
    n = np.asarray(counts, dtype=np.int64)
    d, c, b, a = Polynomial.fit(bins[1:], n, 3).convert().coef
    discriminant = 4 * b ** 2 - 12 * a * c
    thresh2 = (-2 * b + np.sqrt(discriminant)) / (6 * a) if discriminant >= 0 else thresh
    start_i = -1
    for i in range(len(bins) - 1):
        if bins[i] <= thresh2 < bins[i + 1]:
            start_i = i
    if start_i == -1:
        return thresh
    i = start_i
    while 0 < i < len(bins) - 2:
        if n[i - 1] < n[i]:
            i -= 1
        elif n[i + 1] < n[i]:
            i += 1
        else:
            break
    if i >= 1:
        i -= 1
    return bins[i]


def labels_from_roots(root):
    """Component roots (smallest linear index of each component) -> scipy.ndimage.label numbering."""
    uniq, inverse = np.unique(root, return_inverse=True)
    return inverse.astype(np.int64) + 1, len(uniq)


def limb_points(pts, lab, nf, n_rows):
    """get_edge_list after canny (ellipse_to_circle.py:251-291) on the point list: pts int [m, 2] edge pixels (row, col)
    in raster order, lab their component labels 1..nf.  -> float [n, 2] limb points."""
    sizes = np.bincount(lab, minlength=nf + 1)
    sizes[0] = -1
    size_list = sizes.tolist()
    chosen = [size_list.index(v) for v in sorted(size_list, reverse=True)[:min(nf, NUM_REG)]]
    in_chosen = np.isin(lab, chosen)
    X = pts[in_chosen]
    hull_labels = set(lab[in_chosen][ConvexHull(X).vertices].tolist())
    keep = [i for i in chosen if i in hull_labels]
    x_min, x_max = np.min(X[:, 0]), np.max(X[:, 0])
    dx = x_max - x_min
    crop = 0.017
    rows = np.zeros(n_rows, dtype=bool)
    rows[int(x_min + dx * crop):int(x_max - dx * crop)] = True
    sel = np.isin(lab, keep) & rows[pts[:, 0]]
    return np.array(pts[sel], dtype='float')
