"""Solar-limb detection and ellipse fit for ellipse_to_circle.

Works on the 4x4 block mean of the disk (1/16 of the pixels, computed on the GPU by
shg_downscale_mean_u16).  Follows the reference's get_flood_image / get_edge_list /
two_step / dofit (ellipse_to_circle.py:148-291, 53-91):

  per-pixel stages, on the GPU (ops.py -> csrc/limb.hip)
    * cv2.blur on float64 (5x5 for the canny thresholds, k x k for the flood image)
    * threshold into the 0 / 65000 flood image, skimage.feature.canny (0.18.3) up to its
      hysteresis masks: Gaussian with mask normalisation, Sobel, hypot, 4-sector NMS
  control plane, on the host (this file; scalars and point lists)
    * median / 99th percentile / 20-bin histogram / cubic fit -> the flood threshold
    * 8-connected labelling of the edge mask, the two largest regions, convex-hull filter,
      row crop -> limb points (SciPy's ndimage.label and ConvexHull, as the reference)
    * LsqEllipse -> fit_ellipse(): Halir & Flusser's direct least-squares ellipse fit
Parity for the cv2 / lsq-ellipse steps is unpinned; they are validated on analytic
ellipses and against the oracle's restatement.  There is no host implementation of the
per-pixel stages in the product.
"""
import math

import numpy as np
import torch
from numpy import polynomial
from scipy.spatial import ConvexHull

from . import ops

NUM_REG = 2          # ellipse_to_circle.py:31


# ---- get_flood_image's threshold (ellipse_to_circle.py:159-225) ---------------------------
def lerp_order_stats(n, q):
    """np.percentile(.., q) (method 'linear') on n values = lerp between two order statistics:
    returns (rank_lo, rank_hi, combine(a, b)), following NumPy's _quantile / _lerp."""
    virtual = (n - 1) * np.true_divide(q, 100)
    lo = min(max(math.floor(virtual), 0), n - 1)
    hi = min(lo + 1, n - 1)
    gamma = virtual - math.floor(virtual)

    def combine(a, b):
        diff = b - a
        return b - diff * (1 - gamma) if gamma >= 0.5 else a + diff * gamma
    return lo, hi, combine


def lerp_gamma(n, q):
    """The interpolation weight lerp_order_stats' combine() uses."""
    virtual = (n - 1) * np.true_divide(q, 100)
    return virtual - math.floor(virtual)


def median_order_stats(n):
    """np.median on n values: the middle order statistic, or the mean of the two middle ones."""
    if n % 2:
        return n // 2, n // 2, (lambda a, b: a)
    return n // 2 - 1, n // 2, (lambda a, b: (a + b) / 2)


def cubic_fit_coef(x, y):
    """== numpy.polynomial.Polynomial.fit(x, y, 3).convert().coef, bit for bit, without the Polynomial objects (the
    class arithmetic costs 0.3 ms of as_series / trimseq bookkeeping per call).  Same steps: map x from its range to
    [-1, 1], least squares there (the very polyfit Polynomial._fit is), then Horner's rule on coefficient arrays with
    np.convolve for the products -- what Polynomial.__call__ does with the line off + scl*x (every convolution term
    is a sum of at most two products, so there is no summation order to get wrong)."""
    from numpy.polynomial import polyutils as pu
    x = np.asarray(x, dtype=np.float64)
    dom = pu.getdomain(x)
    if dom[0] == dom[1]:
        dom[0] -= 1
        dom[1] += 1
    off, scl = pu.mapparms(dom, np.array([-1.0, 1.0]))
    coef = polynomial.polynomial.polyfit(off + scl * x, y, 3)
    line = np.array([0.0 + off, 1.0 * scl])                      # off + scl * identity
    acc = np.array([coef[3] + 0.0])
    for k in (2, 1, 0):
        acc = pu.trimseq(np.convolve(acc, line))
        acc[0] += coef[k]
    return pu.trimseq(acc)


def flood_threshold(total, shape, mn, mx, counts):
    """thresh3 of get_flood_image from the image statistics the GPU reduces:
    total = np.sum(image); over data = blurred[blurred < very_bright]: mn, mx = data.min(), data.max(),
    counts = np.histogram(data, bins=20)[0].  Pixels of `blurred` below thresh3 become 0, the others 65000."""
    h, w = shape
    thresh = 0.9 * total / (h * w)
    if mn == mx:                                      # np.histogram's range for constant data
        mn, mx = mn - 0.5, mx + 0.5
    bins = np.linspace(mn, mx, 21)
    n = np.asarray(counts, dtype=np.int64)
    d, c, b, a = cubic_fit_coef(bins[1:], n)
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


# ---- get_edge_list after canny (ellipse_to_circle.py:251-291) ---------------------------------
def limb_points(pts, lab, nf, n_rows):
    """pts: int [m, 2] edge pixels (row, col) in raster order; lab: their component labels 1..nf
    (scipy.ndimage.label numbering).  The reference builds full-size masks per region; these are the
    same selections on the point list (raster order is what np.argwhere of any sub-mask returns)."""
    sizes = np.bincount(lab, minlength=nf + 1)
    sizes[0] = -1
    size_list = sizes.tolist()
    # regions are picked by size VALUE: equal sizes resolve to the first such region (list.index)
    chosen = [size_list.index(v) for v in sorted(size_list, reverse=True)[:min(nf, NUM_REG)]]
    member = np.zeros(nf + 1, dtype=bool)               # label -> selected? (a table lookup instead of np.isin's sort)
    member[np.asarray(chosen, dtype=np.int64)] = True
    in_chosen = member[lab]
    X = pts[in_chosen]
    hull_labels = set(lab[in_chosen][ConvexHull(X).vertices].tolist())
    keep = [i for i in chosen if i in hull_labels]    # regions that own a convex-hull vertex
    x_min, x_max = np.min(X[:, 0]), np.max(X[:, 0])
    dx = x_max - x_min
    crop = 0.017
    rows = np.zeros(n_rows, dtype=bool)
    rows[int(x_min + dx * crop):int(x_max - dx * crop)] = True     # slice semantics of mask[int(..):int(..), :] = 1
    member[:] = False
    member[np.asarray(keep, dtype=np.int64)] = True
    sel = member[lab] & rows[pts[:, 0]]
    return np.array(pts[sel], dtype='float'), pts


def labels_from_roots(root):
    """Component roots (smallest linear index of each component) -> scipy.ndimage.label numbering:
    label k is the k-th component met in raster order, i.e. the k-th smallest root."""
    uniq, inverse = np.unique(root, return_inverse=True)
    return inverse.astype(np.int64) + 1, len(uniq)


def edge_points(small, sigma=2):
    """small: float64 GPU tensor, the 4x4 block mean of disk/65536.
    -> (X float [n, 2] limb points (row, col), raw_X int [m, 2] all canny points)."""
    h, w = small.shape
    n = h * w
    k = int(h * 0.01)
    if k <= 0:
        raise RuntimeError('ellipse fit: the scan needs at least 400 slit rows (cv2.blur kernel int(0.01 * h/4) = 0)')
    blurred = ops.box_blur_f64(small, k)
    m_lo, m_hi, median = median_order_stats(n)
    p_lo, p_hi, _ = lerp_order_stats(n, 99)
    blur5 = ops.box_blur_f64(small, 5)
    sel_d = ops.select_multi_f64([blur5, blur5, blurred, blurred], [m_lo, m_hi, p_lo, p_hi])
    # very_bright = np.percentile(img_blurred, 99) (:165) is interpolated on the device from its two order statistics, so
    # the flood statistics follow without a host round trip; one read brings everything the host needs
    stats, counts = ops.flood_stats_lerp(small, blurred, sel_d[2:4], lerp_gamma(n, 99))
    packed = torch.cat([sel_d[:2], stats, counts.to(torch.float64)]).cpu().numpy()
    low = median(packed[0], packed[1]) / 10             # low_threshold = median(blur 5x5) / 10 (:241-242)
    high = low * 1.5
    thresh3 = flood_threshold(packed[2], (h, w), packed[3], packed[4], packed[5:].astype(np.int64))
    while True:
        if sigma <= 0:
            raise RuntimeError('ellipse fit: could not find any edges of the solar disk')
        low_mask, high_mask = ops.canny_masks(blurred, thresh3, sigma, low, high)
        idx, root = ops.edge_components(low_mask, high_mask)          # hysteresis + labelling, on the GPU
        if idx.size:
            break
        sigma -= 0.5                                   # try again with less blur (:254-256)
    pts = np.stack([idx // w, idx % w], axis=1).astype(np.int64)
    lab, nf = labels_from_roots(root)
    return limb_points(pts, lab, nf, h)


# ---- LsqEllipse (Halir & Flusser) -----------------------------------------------------------
def fit_ellipse(points):
    """-> (center[2], width, height, phi, coefficients[6]) in the coordinate order of `points`."""
    pts = np.asarray(points, dtype=float)
    x, y = pts[:, 0], pts[:, 1]
    D1 = np.vstack([x ** 2, x * y, y ** 2]).T
    D2 = np.vstack([x, y, np.ones_like(x)]).T
    S1 = D1.T @ D1
    S2 = D1.T @ D2
    S3 = D2.T @ D2
    C1 = np.array([[0., 0., 2.], [0., -1., 0.], [2., 0., 0.]])
    M = np.linalg.inv(C1) @ (S1 - S2 @ np.linalg.inv(S3) @ S2.T)
    _, eigvec = np.linalg.eig(M)
    cond = 4 * np.multiply(eigvec[0, :], eigvec[2, :]) - np.power(eigvec[1, :], 2)
    a1 = eigvec[:, np.nonzero(cond > 0)[0]]
    a2 = np.linalg.inv(-S3) @ S2.T @ a1
    coef = np.vstack([a1, a2]).ravel()
    a, b, c, d, f, g = coef[0], coef[1] / 2., coef[2], coef[3] / 2., coef[4] / 2., coef[5]
    x0 = (c * d - b * f) / (b ** 2. - a * c)
    y0 = (a * f - b * d) / (b ** 2. - a * c)
    numerator = 2 * (a * f ** 2 + c * d ** 2 + g * b ** 2 - 2 * b * d * f - a * c * g)
    root = np.sqrt(1 + 4 * b * b / ((a - c) * (a - c)))
    width = np.sqrt(numerator / ((b * b - a * c) * ((c - a) * root - (c + a))))
    height = np.sqrt(numerator / ((b * b - a * c) * ((a - c) * root - (c + a))))
    phi = .5 * np.arctan((2. * b) / (a - c))
    return [x0, y0], width, height, phi, coef


def ellipse_outline(center, width, height, phi, n_points=100):
    t = np.linspace(0, 2 * np.pi, n_points)
    x = center[0] + width * np.cos(t) * np.cos(phi) - height * np.sin(t) * np.sin(phi)
    y = center[1] + width * np.cos(t) * np.sin(phi) + height * np.sin(t) * np.cos(phi)
    return np.c_[x, y]


# ---- two_step (ellipse_to_circle.py:62-91) ----------------------------------------------------
def two_step(points, correction_matrix):
    center, width, height, phi, _ = fit_ellipse(points)
    mat, _ = correction_matrix(phi, height / width)
    Xr = mat @ (points - np.array(center)).T * height
    values = np.linalg.norm(Xr, axis=0) - 1
    kept = points[values > -max(values)]
    center, width, height, phi, _ = fit_ellipse(kept)
    outline = ellipse_outline(center, width, height, phi)
    ratio = width / height
    for _ in range(2):                                  # bring phi within pi/4 of 0 by swapping the axis labels
        if phi > math.pi / 4:
            phi -= math.pi / 2
            ratio = 1 / ratio
            height = height / ratio
        if phi < -math.pi / 4:
            phi += math.pi / 2
            ratio = 1 / ratio
            height = height / ratio
    return np.array(center), height, phi, ratio, kept, outline
