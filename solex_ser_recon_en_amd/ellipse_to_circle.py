"""Geometric correction: limb ellipse fit and the ellipse -> circle warp.

Same surface as the reference's ellipse_to_circle.py: get_correction_matrix (:39-50),
correct_image (:94-145) and ellipse_to_circle (:294-342).  The warp runs on the GPU
(shg_warp_rows_u16); the limb fit works on the GPU-computed 4x4 block mean and is host
control plane (limb_fit.py); the 2x2 matrix algebra stays NumPy as in the reference.
"""
import functools
import math

import numpy as np

from . import limb_fit, ops, outputs, timing
from .device import DeviceImage, to_device_u16, u16_from_unit_float
from .solex_util import logme, output_path


def rot(x):
    return np.array([[np.cos(x), np.sin(x)], [-np.sin(x), np.cos(x)]])


def get_correction_matrix(phi, r):
    """IN: tilt phi, ellipse axes ratio (height / width).  OUT: inverse correction matrix, unrotation angle."""
    stretch_matrix = rot(phi) @ np.array([[r, 0], [0, 1]]) @ rot(-phi)
    theta = np.arctan(stretch_matrix[1, 0] / stretch_matrix[0, 0])
    correction_matrix = rot(theta) @ stretch_matrix
    correction_matrix[1, 0] = 0
    correction_matrix /= correction_matrix[1, 1]
    return np.linalg.inv(correction_matrix), theta


@functools.lru_cache(maxsize=64)
def _warp_geometry(phi, ratio, h, w):
    """Everything correct_image derives from (phi, ratio) and the image shape (ellipse_to_circle.py:100-114): the disks of
    a Doppler stack share one geometry, so the 2x2 algebra runs once per file instead of once per disk."""
    mat, theta = get_correction_matrix(phi, ratio)
    mat3 = np.zeros((3, 3))
    mat3[:2, :2] = mat
    mat3[2, 2] = 1
    corners = np.array([[0, 0], [0, h], [w, 0], [w, h]])
    inv_mat = np.linalg.inv(mat)              # the reference re-inverts at each use; same input, same result
    new_corners = (inv_mat @ corners.T).T
    new_h = np.max(new_corners[:, 1]) - np.min(new_corners[:, 1])
    new_w = np.max(new_corners[:, 0]) - np.min(new_corners[:, 0])
    origin = np.array([np.min(new_corners[:, 0]), np.min(new_corners[:, 1])])
    mat3 = mat3 @ np.array([[1, 0, origin[0]], [0, 1, origin[1]], [0, 0, 1]])
    if not (mat3[1, 0] == 0 and mat3[1, 1] == 1 and mat3[1, 2] == 0 and mat3[2, 0] == 0 and mat3[2, 1] == 0
            and mat3[2, 2] == 1):
        raise RuntimeError('correct_image: the correction never moves rows (ellipse_to_circle.py:48-49); got\n%s' % mat3)
    for a in (mat, inv_mat, mat3, origin):
        a.setflags(write=False)
    return mat, theta, inv_mat, mat3, int(np.ceil(new_h)), int(np.ceil(new_w)), origin, np.linalg.det(mat)


def correct_image(image, phi, ratio, center, height, options, print_log=False):
    """image: the uint16 disk (DeviceImage / tensor / ndarray) or, as the reference passes it,
    float64 disk/65536.  Returns (uint16 DeviceImage, (cx, cy, radius), mat3)."""
    src = to_device_u16(u16_from_unit_float(image))
    h, w = src.shape
    mat, theta, inv_mat, mat3, out_h, out_w, origin, det = _warp_geometry(float(phi), float(ratio), int(h), int(w))
    fixed = ops.warp_rows_u16(src, mat3[0, 0], mat3[0, 1], mat3[0, 2], out_h, out_w)
    center = np.asarray(center)
    new_center = (inv_mat @ center.T).T - origin
    new_radius = height * np.sqrt(np.abs(ratio / det))
    if print_log and '_nolog' not in options:
        basefich0 = options['basefich0']
        print('unrotation angle theta = ' + "{:.3f}".format(math.degrees(theta)) + " degrees")
        np.set_printoptions(suppress=True)
        logme(basefich0 + '_log.txt', options, 'Y/X ratio : ' + "{:.3f}".format(ratio))
        print('Y/X ratio : ' + "{:.3f}".format(ratio))
        logme(basefich0 + '_log.txt', options, 'Tilt angle : ' + "{:.3f}".format(math.degrees(phi)) + " degrees")
        logme(basefich0 + '_log.txt', options, 'Linear transform correction matrix : \n' + str(mat))
        logme(basefich0 + '_log.txt', options, 'Disk position, radius : ' + (
            (str(new_center) + ', ' + "{:.3f}".format(new_radius)) if not height == -1.0 else 'UNKNOWN'))
        logme(basefich0 + '_log.txt', options, 'Unrotation : ' + "{:.3f}".format(math.degrees(theta)) + " degrees")
        np.set_printoptions(suppress=False)
    return DeviceImage(fixed), (new_center[0], new_center[1], new_radius), mat3.copy()


def ellipse_to_circle(image, options, basefich):
    """image: the uint16 raw disk.  Returns (fix_img, (cx, cy, r), ratio, phi, borders)."""
    src = to_device_u16(image)
    factor = 4
    with timing.stage('  limb: edges'):
        small = ops.downscale_mean_u16(src, factor)                    # downscale_local_mean(image / 65536, (4, 4))
        X, raw_X = limb_fit.edge_points(small)
    X, raw_X = X * factor, raw_X * factor                              # down-scaled, then upscaled back (:301-302)
    with timing.stage('  limb: ellipse lsq (host)'):
        center, height, phi, ratio, X_f, ellipse_points = limb_fit.two_step(X, get_correction_matrix)
    center = np.array([center[1], center[0]])
    fix_img, new_circle, mat3 = correct_image(src, phi, ratio, center, height, options, print_log=True)

    X_f3 = np.ones((X_f.shape[0], 3))
    X_f3[:, 1] = X_f[:, 0]                                             # X_f is (y, x), X_f3 is (x, y)
    X_f3[:, 0] = X_f[:, 1]
    X_f3_t = (np.linalg.inv(mat3) @ X_f3.T).T
    borders = [np.min(X_f3_t[:, 0]), np.min(X_f3_t[:, 1]), np.max(X_f3_t[:, 0]), np.max(X_f3_t[:, 1])]
    print('sun borders found:' + str(borders))
    if not options['clahe_only'] and not options['protus_only'] and '_nolog' not in options:
        outputs.submit(outputs.plot_ellipse_fit, output_path(basefich + '_ellipse_fit.png', options),
                       DeviceImage(src), fix_img, raw_X, X_f, ellipse_points, borders)
    return fix_img, new_circle, ratio, phi, borders
