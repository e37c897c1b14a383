"""Geometric correction: limb ellipse fit and the ellipse -> circle warp.

Same surface as the reference's ellipse_to_circle.py: get_correction_matrix (:39-50),
correct_image (:94-145) and ellipse_to_circle (:294-342).  The limb detection and the fit are one
stage call (stages.limb_fit -> shg_stage_limb_fit: kernels on the 4x4 block mean, then the C++ control
plane with NumPy's own BLAS / LAPACK routines); the warp is a kernel (shg_warp_rows_u16);
get_correction_matrix stays NumPy as in the reference (public surface; the product path computes the
same matrix bit for bit in shg_host_correction_matrix).
"""
import functools
import math

import numpy as np

from . import hostmath, ops, outputs, stages, timing
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
    a Doppler stack share one geometry, so the 2x2 algebra runs once per file instead of once per disk.  Computed by the
    host control plane with NumPy's own BLAS / LAPACK routines (shg_host_warp_geometry: bit-identical to the NumPy
    statement, tests/test_hostmath_cpu.py) -- the routine the limb-fit stage uses for the first disk of a file."""
    g = hostmath.warp_geometry(phi, ratio, h, w)
    mat3, inv_mat, origin = g['mat3'], g['inv_mat'], g['origin']
    for a in (inv_mat, mat3, origin):
        a.setflags(write=False)
    return g['theta'], inv_mat, mat3, g['out_h'], g['out_w'], origin, g['det']


def correct_image(image, phi, ratio, center, height, options, print_log=False):
    """image: the uint16 disk (DeviceImage / tensor / ndarray) or, as the reference passes it,
    float64 disk/65536.  Returns (uint16 DeviceImage, (cx, cy, radius), mat3)."""
    src = to_device_u16(u16_from_unit_float(image))
    h, w = src.shape
    theta, inv_mat, mat3, out_h, out_w, origin, det = _warp_geometry(float(phi), float(ratio), int(h), int(w))
    fixed = ops.warp_rows_u16(src, mat3[0, 0], mat3[0, 1], mat3[0, 2], out_h, out_w, minmax=getattr(image, 'minmax', None))
    center = np.asarray(center)
    new_center = (inv_mat @ center.T).T - origin
    new_radius = height * np.sqrt(np.abs(ratio / det))
    if print_log:
        _log_geometry(options, phi, ratio, theta, new_center, new_radius, known=not height == -1.0)
    return DeviceImage(fixed), (new_center[0], new_center[1], new_radius), mat3.copy()


def _log_geometry(options, phi, ratio, theta, new_center, new_radius, known=True):
    """The log lines of correct_image (ellipse_to_circle.py:131-143)."""
    if '_nolog' in options:
        return
    basefich0 = options['basefich0']
    print('unrotation angle theta = ' + "{:.3f}".format(math.degrees(theta)) + " degrees")
    np.set_printoptions(suppress=True)
    logme(basefich0 + '_log.txt', options, 'Y/X ratio : ' + "{:.3f}".format(ratio))
    print('Y/X ratio : ' + "{:.3f}".format(ratio))
    logme(basefich0 + '_log.txt', options, 'Tilt angle : ' + "{:.3f}".format(math.degrees(phi)) + " degrees")
    logme(basefich0 + '_log.txt', options, lambda: 'Linear transform correction matrix : \n' + str(get_correction_matrix(phi, ratio)[0]))
    logme(basefich0 + '_log.txt', options, 'Disk position, radius : ' + (
        (str(new_center) + ', ' + "{:.3f}".format(new_radius)) if known else 'UNKNOWN'))
    logme(basefich0 + '_log.txt', options, 'Unrotation : ' + "{:.3f}".format(math.degrees(theta)) + " degrees")
    np.set_printoptions(suppress=False)


def ellipse_to_circle(image, options, basefich, need_image=True):
    """image: the uint16 raw disk.  Returns (fix_img, (cx, cy, r), ratio, phi, borders).
    need_image=False (solex_process, when the ellipse-fit shift is not a requested disk and no diagnostic plot is drawn):
    the corrected image nobody looks at is not computed, fix_img is None.
    One stage call (shg_stage_limb_fit: block mean, flood image, canny ladder, labelled edges, region / hull / row
    selection, two-step ellipse fit, geometry of the corrected image, borders) and the warp kernel."""
    src = to_device_u16(image)
    plots = not options['clahe_only'] and not options['protus_only'] and '_nolog' not in options
    with timing.stage('  limb: fit'):
        g = stages.limb_fit(src, want_points=plots)
    phi, ratio = g['phi'], g['ratio']
    fix_img = None
    if need_image or plots:
        with timing.stage('  limb: warp'):
            fix_img = DeviceImage(ops.warp_rows_u16(src, g['h00'], g['h01'], g['h02'], g['out_h'], g['out_w'],
                                                    minmax=getattr(image, 'minmax', None)))
    new_circle = g['circle']
    _log_geometry(options, phi, ratio, g['theta'], np.array(new_circle[:2]), new_circle[2])
    borders = g['borders']
    print('sun borders found:' + str(borders))
    if plots:
        outputs.submit(outputs.plot_ellipse_fit, output_path(basefich + '_ellipse_fit.png', options),
                       DeviceImage(src), fix_img, g['raw_X'], g['X_f'], g['outline'], borders)
    return fix_img, new_circle, ratio, phi, borders
