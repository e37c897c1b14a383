"""Solar-limb detection and ellipse fit: the host control plane of ellipse_to_circle.

It works on the 4x4 block-mean of the disk (1/16 of the pixels; computed on the GPU by
shg_downscale_mean_u16) and produces five numbers (centre, axis, tilt, ratio).  Follows
the reference's get_flood_image / get_edge_list / two_step / dofit
(ellipse_to_circle.py:148-291, 53-91) with SciPy in place of the libraries this image lacks:

  * cv2.blur on float64        -> scipy.ndimage.uniform_filter1d, mode 'mirror' (= BORDER_REFLECT_101)
  * skimage.feature.canny      -> canny_edges() below, scikit-image 0.18.3 semantics
                                   (Gaussian with mask normalisation, Sobel, 4-sector interpolated
                                   non-maximum suppression, hysteresis by 8-connected labelling)
  * lsq-ellipse LsqEllipse     -> fit_ellipse(): Halir & Flusser's numerically stable direct
                                   least-squares fit, parameters per Wolfram MathWorld eqs. 19-23
Parity for the cv2 / lsq-ellipse steps is unpinned; they are validated on analytic
ellipses and against the oracle's restatement.
"""
import math

import numpy as np
from numpy import polynomial
from scipy import ndimage as ndi
from scipy.spatial import ConvexHull

NUM_REG = 2          # ellipse_to_circle.py:31


# ---- cv2.blur(float64 image, (k, k)) ------------------------------------------------
def box_blur_f64(img, k):
    if k <= 0:
        raise ValueError('blur kernel must be positive (image too small for the limb fit: needs >= 400 rows)')
    out = ndi.uniform_filter1d(np.asarray(img, dtype=np.float64), k, axis=0, mode='mirror')
    return ndi.uniform_filter1d(out, k, axis=1, mode='mirror')


# ---- get_flood_image (ellipse_to_circle.py:148-228) ------------------------------------
def flood_image(image):
    h, w = image.shape
    thresh = 0.9 * np.sum(image) / (h * w)
    blurred = box_blur_f64(image, int(h * 0.01))
    very_bright = np.percentile(blurred, 99)
    data = blurred.ravel()
    data = data[data < very_bright]
    n, bins = np.histogram(data, bins=20)
    d, c, b, a = polynomial.polynomial.Polynomial.fit(bins[1:], n, 3).convert().coef
    discriminant = 4 * b ** 2 - 12 * a * c
    thresh2 = (-2 * b + np.sqrt(discriminant)) / (6 * a) if discriminant >= 0 else thresh
    start_i = -1
    for i in range(len(bins) - 1):
        if bins[i] <= thresh2 < bins[i + 1]:
            start_i = i
    if start_i == -1:
        thresh3 = thresh
    else:
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
        thresh3 = bins[i]
    return np.where(blurred < thresh3, 0.0, 65000.0)


# ---- skimage.feature.canny (0.18.3) ------------------------------------------------------
def canny_edges(image, sigma, low_threshold, high_threshold):
    image = np.asarray(image, dtype=np.float64)
    h, w = image.shape
    # mask-normalised smoothing (smooth_with_function_and_mask with an all-ones mask)
    bleed = ndi.gaussian_filter(np.ones((h, w)), sigma, mode='constant')
    smoothed = ndi.gaussian_filter(image, sigma, mode='constant') / (bleed + np.finfo(float).eps)
    js = ndi.sobel(smoothed, axis=1)
    is_ = ndi.sobel(smoothed, axis=0)
    ai, aj = np.abs(is_), np.abs(js)
    mag = np.hypot(is_, js)
    interior = np.zeros((h, w), dtype=bool)
    interior[1:-1, 1:-1] = True                       # binary_erosion of a full mask with border_value=0
    ok = interior & (mag > 0)

    mp = np.pad(mag, 1)                                # mp[y+1+dy, x+1+dx] = mag[y+dy, x+dx]

    def nb(dy, dx):
        return mp[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]

    with np.errstate(divide='ignore', invalid='ignore'):
        w_ji = aj / ai
        w_ij = ai / aj
    same = ((is_ >= 0) & (js >= 0)) | ((is_ <= 0) & (js <= 0))
    opp = ((is_ <= 0) & (js >= 0)) | ((is_ >= 0) & (js <= 0))
    local = np.zeros((h, w), dtype=bool)

    def sector(pts, wgt, p1, p2, m1, m2):
        nonlocal local
        pts = ok & pts
        c_plus = nb(*p2) * wgt + nb(*p1) * (1 - wgt) <= mag
        c_minus = nb(*m2) * wgt + nb(*m1) * (1 - wgt) <= mag
        local = np.where(pts, c_plus & c_minus, local)   # later sectors overwrite earlier ones, as in skimage

    sector(same & (ai >= aj), w_ji, (1, 0), (1, 1), (-1, 0), (-1, -1))      # 0-45 degrees
    sector(same & (ai <= aj), w_ij, (0, 1), (1, 1), (0, -1), (-1, -1))      # 45-90
    sector(opp & (ai <= aj), w_ij, (0, 1), (-1, 1), (0, -1), (1, -1))       # 90-135
    sector(opp & (ai >= aj), w_ji, (-1, 0), (-1, 1), (1, 0), (1, -1))       # 135-180

    high_mask = local & (mag >= high_threshold)
    low_mask = local & (mag >= low_threshold)
    labels, count = ndi.label(low_mask, np.ones((3, 3), bool))
    if count == 0:
        return low_mask
    good = np.zeros(count + 1, dtype=bool)
    good[np.unique(labels[high_mask])] = True
    good[0] = False
    return good[labels]


# ---- get_edge_list (ellipse_to_circle.py:231-291) ----------------------------------------
def edge_points(image, sigma=2):
    """-> (X float [n, 2] limb points (row, col), raw_X int [m, 2] all canny points)."""
    low = np.median(box_blur_f64(image, 5)) / 10
    high = low * 1.5
    flooded = flood_image(image)
    while True:
        if sigma <= 0:
            raise RuntimeError('ellipse fit: could not find any edges of the solar disk')
        edges = canny_edges(flooded, sigma, low, high)
        labelled, nf = ndi.label(edges, np.ones((3, 3), int))
        if nf:
            break
        sigma -= 0.5                                   # try again with less blur (:254-256)
    raw_X = np.argwhere(edges)
    sizes = np.bincount(labelled.ravel(), minlength=nf + 1)
    sizes[0] = -1
    size_list = sizes.tolist()
    # the reference picks regions by size VALUE: equal sizes resolve to the first such region
    chosen = [size_list.index(v) for v in sorted(size_list, reverse=True)[:min(nf, NUM_REG)]]
    filt = np.isin(labelled, chosen)
    X = np.argwhere(filt)
    hull = X[ConvexHull(X).vertices]
    on_hull = np.zeros(edges.shape, dtype=bool)
    on_hull[hull[:, 0], hull[:, 1]] = True
    keep = [i for i in chosen if np.any((labelled == i) & on_hull)]
    filt = np.isin(labelled, keep)
    x_min, x_max = np.min(X[:, 0]), np.max(X[:, 0])
    dx = x_max - x_min
    crop = 0.017
    rows = np.zeros(edges.shape[0], dtype=bool)
    rows[int(x_min + dx * crop):int(x_max - dx * crop)] = True
    filt &= rows[:, None]
    return np.array(np.argwhere(filt), dtype='float'), raw_X


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
