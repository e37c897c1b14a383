"""CPU oracle for the limb-fit chain of ellipse_to_circle.  TEST INFRASTRUCTURE ONLY.

Restates, step by step and in the reference's order, ellipse_to_circle.py:148-342:
get_flood_image, get_edge_list, dofit (LsqEllipse), two_step, ellipse_to_circle.
Third-party pieces:
  * skimage.transform.downscale_local_mean, skimage.feature.canny -- restated from
    scikit-image 0.18.3 and PINNED: tests/golden/g13_limb.npz holds the outputs of the real
    library for the same inputs (oracle/capture_goldens.py G13);
  * cv2.blur(float64) -- UNPINNED (box_blur_f64 in shg_oracle.py);
  * lsq-ellipse (LsqEllipse.fit / as_parameters / return_fit) -- UNPINNED: the package is
    absent here and unpinned upstream (requirements.txt).  LsqEllipse below restates the
    published algorithm (Halir & Flusser 1998; parameters per MathWorld "Ellipse" eqs.
    19-23) and is checked on analytic ellipses.  The capture script installs it as
    `ellipse.LsqEllipse` for shim-mode runs of the reference.
Compatible with NumPy 1.26 / SciPy 1.7 (capture) and NumPy 2.x / SciPy 1.15 (tests).
"""
import math

import numpy as np
from numpy import polynomial
import scipy.ndimage as ndi
from scipy.spatial import ConvexHull

try:
    from . import shg_oracle as orc
except ImportError:                      # capture script imports the oracle as top-level modules
    import shg_oracle as orc

NUM_REG = 2                              # ellipse_to_circle.py:31


def downscale_local_mean(image, f):
    """skimage.transform.downscale_local_mean(image, (f, f)): zero-pad to a multiple of f, block mean."""
    h, w = image.shape
    ph, pw = -h % f, -w % f
    padded = np.pad(image, ((0, ph), (0, pw)), mode='constant', constant_values=0)
    blocks = padded.reshape(padded.shape[0] // f, f, padded.shape[1] // f, f)
    return np.mean(blocks, axis=(1, 3))


class LsqEllipse:
    """Direct least-squares ellipse fit (Halir & Flusser)."""

    def fit(self, X):
        X = np.asarray(X, dtype=float)
        x, y = X.T
        D1 = np.vstack([x ** 2, x * y, y ** 2]).T
        D2 = np.vstack([x, y, np.ones_like(x)]).T
        S1, S2, S3 = D1.T @ D1, D1.T @ D2, D2.T @ D2
        C1 = np.array([[0., 0., 2.], [0., -1., 0.], [2., 0., 0.]])
        M = np.linalg.inv(C1) @ (S1 - S2 @ np.linalg.inv(S3) @ S2.T)
        eigval, eigvec = np.linalg.eig(M)
        cond = 4 * np.multiply(eigvec[0, :], eigvec[2, :]) - np.power(eigvec[1, :], 2)
        a1 = eigvec[:, np.nonzero(cond > 0)[0]]
        a2 = np.linalg.inv(-S3) @ S2.T @ a1
        self.coef_ = np.vstack([a1, a2])
        return self

    @property
    def coefficients(self):
        return np.asarray(self.coef_).ravel()

    def as_parameters(self):
        a = self.coefficients[0]
        b = self.coefficients[1] / 2.
        c = self.coefficients[2]
        d = self.coefficients[3] / 2.
        f = self.coefficients[4] / 2.
        g = self.coefficients[5]
        x0 = (c * d - b * f) / (b ** 2. - a * c)
        y0 = (a * f - b * d) / (b ** 2. - a * c)
        numerator = 2 * (a * f ** 2 + c * d ** 2 + g * b ** 2 - 2 * b * d * f - a * c * g)
        denominator1 = (b * b - a * c) * ((c - a) * np.sqrt(1 + 4 * b * b / ((a - c) * (a - c))) - (c + a))
        denominator2 = (b * b - a * c) * ((a - c) * np.sqrt(1 + 4 * b * b / ((a - c) * (a - c))) - (c + a))
        width = np.sqrt(numerator / denominator1)
        height = np.sqrt(numerator / denominator2)
        phi = .5 * np.arctan((2. * b) / (a - c))
        return [x0, y0], width, height, phi

    def return_fit(self, n_points=None, t=None):
        if t is None:
            t = np.linspace(0, 2 * np.pi, n_points)
        center, width, height, phi = self.as_parameters()
        x = center[0] + width * np.cos(t) * np.cos(phi) - height * np.sin(t) * np.sin(phi)
        y = center[1] + width * np.cos(t) * np.sin(phi) + height * np.sin(t) * np.cos(phi)
        return np.c_[x, y]


def canny(image, sigma, low_threshold, high_threshold):
    """skimage.feature.canny(image, sigma, low_threshold, high_threshold) for a float image
    (dtype_max = 1, mask = all ones), scikit-image 0.18.3."""
    eps = np.finfo(float).eps
    mask = np.ones(image.shape, dtype=bool)

    def fsmooth(x):
        return ndi.gaussian_filter(x, sigma, mode='constant', cval=0, truncate=4.0)

    bleed_over = fsmooth(mask.astype(float))
    smoothed = fsmooth(np.array(image, dtype=float)) / (bleed_over + eps)
    jsobel = ndi.sobel(smoothed, axis=1)
    isobel = ndi.sobel(smoothed, axis=0)
    abs_isobel, abs_jsobel = np.abs(isobel), np.abs(jsobel)
    magnitude = np.hypot(isobel, jsobel)
    eroded = ndi.binary_erosion(mask, ndi.generate_binary_structure(2, 2), border_value=0) & (magnitude > 0)
    local_maxima = np.zeros(image.shape, bool)

    def suppress(pts, w_num, w_den, plus1, plus2, minus1, minus2):
        pts = eroded & pts
        m = magnitude[pts]
        w = w_num[pts] / w_den[pts]
        c_plus = plus2(pts) * w + plus1(pts) * (1 - w) <= m
        c_minus = minus2(pts) * w + minus1(pts) * (1 - w) <= m
        local_maxima[pts] = c_plus & c_minus

    mg = magnitude
    pos = (isobel >= 0) & (jsobel >= 0)
    neg = (isobel <= 0) & (jsobel <= 0)
    suppress((pos | neg) & (abs_isobel >= abs_jsobel), abs_jsobel, abs_isobel,
             lambda p: mg[1:, :][p[:-1, :]], lambda p: mg[1:, 1:][p[:-1, :-1]],
             lambda p: mg[:-1, :][p[1:, :]], lambda p: mg[:-1, :-1][p[1:, 1:]])
    suppress((pos | neg) & (abs_isobel <= abs_jsobel), abs_isobel, abs_jsobel,
             lambda p: mg[:, 1:][p[:, :-1]], lambda p: mg[1:, 1:][p[:-1, :-1]],
             lambda p: mg[:, :-1][p[:, 1:]], lambda p: mg[:-1, :-1][p[1:, 1:]])
    a = (isobel <= 0) & (jsobel >= 0)
    b = (isobel >= 0) & (jsobel <= 0)
    suppress((a | b) & (abs_isobel <= abs_jsobel), abs_isobel, abs_jsobel,
             lambda p: mg[:, 1:][p[:, :-1]], lambda p: mg[:-1, 1:][p[1:, :-1]],
             lambda p: mg[:, :-1][p[:, 1:]], lambda p: mg[1:, :-1][p[:-1, 1:]])
    suppress((a | b) & (abs_isobel >= abs_jsobel), abs_jsobel, abs_isobel,
             lambda p: mg[:-1, :][p[1:, :]], lambda p: mg[:-1, 1:][p[1:, :-1]],
             lambda p: mg[1:, :][p[:-1, :]], lambda p: mg[1:, :-1][p[:-1, 1:]])

    high_mask = local_maxima & (magnitude >= high_threshold)
    low_mask = local_maxima & (magnitude >= low_threshold)
    labels, count = ndi.label(low_mask, np.ones((3, 3), bool))
    if count == 0:
        return low_mask
    sums = np.array(ndi.sum(high_mask, labels, np.arange(count, dtype=np.int32) + 1), ndmin=1)
    good_label = np.zeros((count + 1,), bool)
    good_label[1:] = sums > 0
    return good_label[labels]


def get_flood_image(image):                                         # ellipse_to_circle.py:148-228
    thresh = 0.9 * np.sum(image) / (image.shape[0] * image.shape[1])
    blur_width = int(image.shape[0] * 0.01)
    img_blurred = orc.box_blur_f64(image, blur_width, blur_width)
    very_bright = np.percentile(img_blurred, 99)
    data = img_blurred.flatten()
    data = data[data < very_bright]
    n, bins = np.histogram(data, bins=20)
    coeff = polynomial.polynomial.Polynomial.fit(bins[1:], n, 3).convert().coef
    d, c, b, a = coeff
    discriminant = 4 * b ** 2 - 12 * a * c
    if discriminant >= 0:
        thresh2 = (-2 * b + np.sqrt(discriminant)) / (6 * a)
    else:
        thresh2 = thresh
    start_i = -1
    for i in range(len(bins) - 1):
        if bins[i] <= thresh2 < bins[i + 1]:
            start_i = i
    if start_i == -1:
        thresh3 = thresh
    else:
        i = start_i
        while i > 0 and i < len(bins) - 2:
            if n[i - 1] < n[i]:
                i -= 1
            elif n[i + 1] < n[i]:
                i += 1
            else:
                break
        if i >= 1:
            i -= 1
        thresh3 = bins[i]
    img_blurred[img_blurred < thresh3] = 0
    img_blurred[img_blurred >= thresh3] = 65000
    return img_blurred


def get_edge_list(image, sigma=2):                                  # ellipse_to_circle.py:231-291
    if sigma <= 0:
        raise RuntimeError('could not find any edges')
    low_threshold = np.median(orc.box_blur_f64(image, 5, 5)) / 10
    high_threshold = low_threshold * 1.5
    image_flooded = get_flood_image(image)
    edges = canny(image_flooded, sigma, low_threshold, high_threshold)
    labelled, nf = ndi.label(edges, structure=[[1, 1, 1], [1, 1, 1], [1, 1, 1]])
    if nf == 0:
        return get_edge_list(image, sigma=sigma - 0.5)
    return points_from_edges(edges)


def points_from_edges(edges):                                       # ellipse_to_circle.py:251-291
    raw_X = np.argwhere(edges)
    labelled, nf = ndi.label(edges, structure=[[1, 1, 1], [1, 1, 1], [1, 1, 1]])
    region_sizes = [-1] + [int(np.sum(labelled == i)) for i in range(1, nf + 1)]
    top = sorted(region_sizes, reverse=True)[:min(nf, NUM_REG)]
    filt = np.zeros(edges.shape)
    for size in top:
        filt[labelled == region_sizes.index(size)] = 1
    X = np.argwhere(filt)
    Xc = X[ConvexHull(X).vertices]
    Xd = np.zeros(edges.shape)
    Xd[Xc[:, 0], Xc[:, 1]] = 1
    filt = np.zeros(edges.shape)
    for size in top:
        if np.any(np.logical_and(labelled == region_sizes.index(size), Xd)):
            filt[labelled == region_sizes.index(size)] = 1
    x_min, x_max = np.min(X[:, 0]), np.max(X[:, 0])
    dx = x_max - x_min
    crop = 0.017
    mask = np.zeros(filt.shape)
    mask[int(x_min + dx * crop):int(x_max - dx * crop), :] = 1
    filt *= mask
    X = np.array(np.argwhere(filt), dtype='float')
    return X, raw_X


def dofit(points):                                                   # ellipse_to_circle.py:53-59
    reg = LsqEllipse().fit(points)
    center, width, height, phi = reg.as_parameters()
    return center, width, height, phi, reg.return_fit(n_points=100)


def two_step(points):                                                # ellipse_to_circle.py:62-91
    center, width, height, phi, _ = dofit(points)
    mat, _ = orc.correction_matrix(phi, height / width)
    Xr = mat @ (points - np.array(center)).T * height
    values = np.linalg.norm(Xr, axis=0) - 1
    points_tresholded = points[values > -max(values)]
    center, width, height, phi, ellipse_points = dofit(points_tresholded)
    ratio = width / height
    for _ in range(2):
        if phi > math.pi / 4:
            phi -= math.pi / 2
            ratio = 1 / ratio
            height = height / ratio
        if phi < -math.pi / 4:
            phi += math.pi / 2
            ratio = 1 / ratio
            height = height / ratio
    return np.array(center), height, phi, ratio, points_tresholded, ellipse_points


def ellipse_to_circle(disk_u16):                                     # ellipse_to_circle.py:294-342
    """-> (fix_img uint16, (cx, cy, r), ratio, phi, borders)."""
    image = disk_u16 / 65536
    factor = 4
    X, raw_X = get_edge_list(downscale_local_mean(image, factor))
    X, raw_X = X * factor, raw_X * factor
    center, height, phi, ratio, X_f, ellipse_points = two_step(X)
    center = np.array([center[1], center[0]])
    fix_img, new_circle, mat3 = orc.correct_image(image, phi, ratio, center, height)
    X_f3 = np.ones((X_f.shape[0], 3))
    X_f3[:, 1] = X_f[:, 0]
    X_f3[:, 0] = X_f[:, 1]
    X_f3_t = (np.linalg.inv(mat3) @ X_f3.T).T
    borders = [np.min(X_f3_t[:, 0]), np.min(X_f3_t[:, 1]), np.max(X_f3_t[:, 0]), np.max(X_f3_t[:, 1])]
    return fix_img, new_circle, ratio, phi, borders
