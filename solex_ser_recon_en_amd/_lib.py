"""ctypes binding of libshg_hip.so (C ABI declared in include/shg_hip.h).

There is deliberately NO fallback: if the HIP library has not been built the
import raises, and every entry point raises RuntimeError on a non-zero status
(message from shg_last_error_string), which propagates out of solex_do_work
exactly like an exception inside the reference's pool worker does
(Solex_recon.py:42).
"""
import ctypes
import os
from ctypes import c_double, c_int, c_int32, c_int64, c_size_t, c_uint16, c_uint32, c_void_p

import torch  # noqa: F401  -- loads torch's own libamdhip64.so first so that ours binds to the same HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libshg_hip.so')

if not os.path.exists(LIB_PATH):
    raise ImportError(
        'libshg_hip.so is missing (%s). Build it with `python -c "import __graft_entry__ as g; g.build()"` or '
        '`make -C solex_ser_recon_en_amd/csrc`. There is no CPU fallback for the SHG hot path.' % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

P = c_void_p
PD = ctypes.POINTER(c_double)
PI64 = ctypes.POINTER(c_int64)


class ScanRequest(ctypes.Structure):
    """shg_scan_request (include/shg_hip.h); pointers as plain addresses."""
    _fields_ = [('struct_bytes', c_uint32), ('start_phase', c_int32),
                ('stack', P), ('n_frames', c_int64), ('height', c_int64), ('width', c_int64), ('frame_stride_px', c_int64),
                ('bytes_per_px', c_int32), ('flip_x', c_int32),
                ('host_shifts', P), ('host_requested', P), ('n_shifts', c_int32), ('want_fit_image', c_int32),
                ('ratio_fixe', c_double), ('slant_fix_deg', c_double),
                ('transversalium', c_int32), ('keep_detrans', c_int32), ('trans_strength', c_int64),
                ('host_taps', P), ('taps_window', c_int64),
                ('crop_square', c_int32), ('has_fixed_width', c_int32), ('fixed_width', c_int64),
                ('disk_display', c_int32), ('tiles', c_int32), ('delta_radius', c_int64), ('clip_limit', c_double),
                ('host_gauss_taps', P),
                ('mean_out', P), ('max_out', P), ('disks', P), ('disk_pitch', c_int64), ('disk_plane_stride', c_int64),
                ('minmax_slots', P), ('arena', P), ('arena_bytes', c_size_t), ('results', P), ('results_bytes', c_size_t), ('workspace', P), ('workspace_bytes', c_size_t),
                ('host_pinned', P), ('host_pinned_bytes', c_size_t),
                ('host_fit', P), ('host_trace_sharp', P), ('host_mask_good', P), ('host_points', P), ('host_flags', P),
                ('points_cap', c_int64), ('host_outline200', P), ('host_factors', P)]


class ScanResult(ctypes.Structure):
    """shg_scan_result (include/shg_hip.h)."""
    _fields_ = [('phase_done', c_int32), ('limb_fitted', c_int32), ('y1', c_int64), ('y2', c_int64), ('p4', c_double * 4),
                ('counts3', c_int64 * 3), ('geom16', c_double * 16), ('phi', c_double), ('ratio', c_double),
                ('h_first', c_double * 3), ('h_rest', c_double * 3), ('theta_first', c_double), ('theta_rest', c_double),
                ('circle3', c_double * 3), ('borders4', c_double * 4), ('circle_out3', c_double * 3),
                ('out_h', c_int64), ('out_w', c_int64), ('frame_pitch', c_int64),
                ('n_out', c_int64), ('prod_w', c_int64), ('prod_pitch', c_int64), ('window', c_int64),
                ('crop4', c_int64 * 4), ('disc3', c_int64 * 3),
                ('fit_image_off', c_int64), ('frames_off', c_int64), ('detrans_off', c_int64), ('products_off', c_int64),
                ('results_off', c_int64), ('needed_arena_bytes', c_size_t), ('needed_results_bytes', c_size_t), ('needed_workspace_bytes', c_size_t)]


SIGNATURES = {
    'shg_abi_version': (c_int, []),
    'shg_last_error_string': (ctypes.c_char_p, []),
    'shg_profile_enable': (c_int, [c_int]),
    'shg_profile_select': (c_int, [ctypes.c_char_p]),
    'shg_profile_reset': (c_int, []),
    'shg_profile_get': (c_int, [ctypes.c_char_p, ctypes.POINTER(c_double), ctypes.POINTER(c_int64)]),
    'shg_profile_total': (c_int, [ctypes.POINTER(c_double), ctypes.POINTER(c_int64)]),
    'shg_profile_dump': (c_int, [ctypes.c_char_p]),
    'shg_host_timing_enable': (c_int, [c_int]),
    'shg_host_timing_report': (c_int, [ctypes.c_char_p, c_size_t]),
    'shg_stream_read_probe': (c_int, [P, c_int64, c_int, c_int, c_int, c_int64, P, P]),
    'shg_accumulate_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    'shg_accumulate_sum_max': (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int64, P, P, P, c_size_t, P]),
    'shg_frame_pitch_bytes': (c_int64, [c_int64]),
    'shg_upload_frames': (c_int, [P, c_int64, P, c_int64, c_int64, P]),
    'shg_unpack_dib_frames': (c_int, [P, c_int64, c_int64, c_int64, c_int64, c_int, c_int64, c_int, P, P, c_int64, P]),
    'shg_finalize_mean_max': (c_int, [P, P, c_int64, c_int64, c_int64, c_int, P, P, P]),
    'shg_reduce_frame_stats': (c_int, [P, c_int, c_int64, c_int64, P, P, P]),
    'shg_accumulate_mean_max': (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int64, P, P, P, c_size_t, P]),
    'shg_box_blur_u16': (c_int, [P, c_int64, c_int64, c_int, c_int, P, P, P]),
    'shg_row_argmin_u16': (c_int, [P, c_int64, c_int64, c_int64, c_int64, P, P]),
    'shg_row_mean_u16': (c_int, [P, c_int64, c_int64, P, P]),
    'shg_blur_fits_fused': (c_int, [c_int64, c_int]),
    'shg_blur_row_mean_u16': (c_int, [P, c_int64, c_int64, c_int, c_int, P, P]),
    'shg_blur_argmin_u16': (c_int, [P, c_int64, c_int64, c_int, c_int, c_int64, c_int64, P, P, P]),
    'shg_extract_columns': (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int64, P, P, P, c_int, P, c_int64, c_int64,
                                    c_int64, c_int64, c_int, P]),
    'shg_extract_columns_minmax': (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int64, P, P, P, c_int, P, c_int64, c_int64,
                                           c_int64, c_int64, c_int, P, P]),
    'shg_extract_dense_fits': (c_int, [P, c_int]),
    'shg_extract_columns_dense': (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int64, P, P, P, P, P, c_int, P, c_int64, c_int64, c_int64, c_int64,
                                          c_int, P, P]),
    'shg_warp_rows_minmax_u16': (c_int, [P, c_int64, c_int64, c_int64, c_double, c_double, c_double, P, c_int64, c_int64,
                                         c_int64, P, P]),
    'shg_warp_rows_u16': (c_int, [P, c_int64, c_int64, c_int64, c_double, c_double, c_double, P, c_int64, c_int64,
                                  c_int64, P, P]),
    'shg_rowpair_logratio_stats': (c_int, [P, c_int64, c_int64, c_int64, c_int64, c_int64, P, P, P, P, P]),
    'shg_rowpair_logratio_stats_mirrored': (c_int, [P, c_int64, c_int64, c_int64, c_int64, c_int64, P, P, P, P, P, P]),
    'shg_line_order_stats_u16': (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int64, c_int64, P, P, P]),
    'shg_scale_rows_u16': (c_int, [P, c_int64, c_int64, c_int64, P, P, P, c_int64, P]),
    'shg_correlate1d_rows_f64': (c_int, [P, c_int64, c_int64, P, c_int, c_int, P, P]),
    'shg_lin_filter_row_sums': (c_int, [P, c_int64, c_int64, c_int64, P, P, P, P, P, c_int, P, P, P]),
    'shg_lin_filter_apply': (c_int, [P, c_int64, c_int64, c_int64, P, P, P, c_int, c_int, P, P, P, P, c_int, P, c_int64, P]),
    'shg_crop_pad_u16': (c_int, [P, c_int64, c_int64, c_int64, P, c_int64, c_int64, c_int64, c_int64, c_int64,
                                 c_int, P]),
    'shg_clahe_workspace_bytes': (c_size_t, [c_int, c_int]),
    'shg_clahe_workspace_bytes_for': (c_size_t, [c_int64, c_int64, c_int, c_int]),
    'shg_clahe': (c_int, [P, c_int64, c_int64, c_int64, c_int, c_double, c_int, P, c_int64, P, c_size_t, P]),
    'shg_hist': (c_int, [P, c_int64, c_int64, c_int64, c_int, P, P]),
    'shg_select_u16_workspace_bytes': (c_size_t, [c_int]),
    'shg_select_u16': (c_int, [P, c_int64, c_int64, c_int64, ctypes.POINTER(c_int64), c_int, P, P, c_size_t, P]),
    'shg_rescale_u16': (c_int, [P, c_int64, c_int64, c_int64, c_double, c_double, c_double, P, c_int64, P]),
    'shg_rescale_u8': (c_int, [P, c_int64, c_int64, c_int64, c_double, c_double, c_double, P, c_int64, P]),
    'shg_contrast_stats_workspace_bytes': (c_size_t, [c_int]),
    'shg_contrast_stats_workspace_bytes_for': (c_size_t, [c_int64, c_int64, c_int]),
    'shg_contrast_stats_u16': (c_int, [P, c_int64, c_int64, c_int64, c_double, c_int, P, c_int64, P, P, P, P, c_size_t, P]),
    'shg_contrast_products_u16': (c_int, [P, c_int64, P, c_int64, c_int64, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_int64, P]),
    'shg_fill_disc_u16': (c_int, [P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_uint16, P, P]),
    'shg_downscale_mean_u16': (c_int, [P, c_int64, c_int64, c_int64, c_int, P, P]),
    'shg_box_blur_f64': (c_int, [P, c_int64, c_int64, c_int, P, P, P]),
    'shg_box_blur_key_f64': (c_int, [P, c_int64, c_int64, c_int, P, P, P, P]),
    'shg_select_keys_workspace_bytes': (c_size_t, [c_int]),
    'shg_select_keys_u32': (c_int, [ctypes.POINTER(c_void_p), c_int64, ctypes.POINTER(c_int64), ctypes.POINTER(c_int), c_int, P, P, c_size_t, P]),
    'shg_select_workspace_bytes': (c_size_t, [c_int]),
    'shg_select_f64': (c_int, [P, c_int64, ctypes.POINTER(c_int64), c_int, P, P, c_size_t, P]),
    'shg_select_multi_f64': (c_int, [ctypes.POINTER(c_void_p), c_int64, ctypes.POINTER(c_int64), c_int, P, P, c_size_t, P]),
    'shg_flood_stats_f64': (c_int, [P, P, c_int64, c_double, P, P, P, P]),
    'shg_flood_stats_lerp_f64': (c_int, [P, P, c_int64, P, c_double, P, P, P, P]),
    'shg_edge_components_workspace_bytes': (c_size_t, [c_int64, c_int64]),
    'shg_edge_components': (c_int, [P, P, c_int64, c_int64, P, P, P, P, c_size_t, P]),
    'shg_canny_workspace_bytes': (c_size_t, [c_int64, c_int64]),
    'shg_canny_masks_f64': (c_int, [P, c_int64, c_int64, c_double, ctypes.POINTER(c_double), c_int, c_double, c_double, P, P, P,
                                    c_size_t, P]),
    'shg_limb_fused_fits': (c_int, [c_int64, c_int64, c_int]),
    'shg_limb_prepare_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int]),
    'shg_limb_prepare': (c_int, [P, c_int64, c_int64, c_int64, c_int, PI64, c_double, P, ctypes.POINTER(c_void_p), P, c_size_t, P]),
    'shg_limb_edges_workspace_bytes': (c_size_t, [c_int64, c_int64]),
    'shg_limb_edges': (c_int, [P, c_int64, c_int64, c_int, c_double, PD, c_int, c_double, c_double, P, P, c_size_t, P]),
    # ---- stage composites ----
    'shg_stage_mean_fit_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    'shg_stage_mean_fit_host_bytes': (c_size_t, [c_int64, c_int64]),
    'shg_stage_mean_fit': (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int64, P, P, c_int64, P, P, P, P, P, P, P, P, c_size_t, P,
                                   c_size_t, P]),
    'shg_stage_extract_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int]),
    'shg_stage_extract': (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int64, P, P, c_int, P, c_int64, c_int64, c_int64, c_int64, c_int,
                                  P, P, c_size_t, P, c_size_t, P]),
    'shg_stage_limb_points_workspace_bytes': (c_size_t, [c_int64, c_int64]),
    'shg_stage_limb_points_host_bytes': (c_size_t, [c_int64, c_int64]),
    'shg_stage_limb_points': (c_int, [P, c_int64, c_int64, c_int64, P, P, P, c_int64, P, P, c_size_t, P, c_size_t, P]),
    'shg_stage_limb_fit': (c_int, [P, c_int64, c_int64, c_int64, P, P, P, c_int64, P, P, P, P, P, c_size_t, P, c_size_t, P]),
    'shg_stage_process_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int64, c_int64, c_int]),
    'shg_stage_process_host_bytes': (c_size_t, [c_int64, c_int64]),
    'shg_stage_process_frames': (c_int, [P, c_int64, c_int64, c_int64, c_int64, c_int, P, P, P, c_int64, P, c_int64, c_int64, c_int64,
                                         c_int64, c_double, c_int, c_int64, c_int64, c_int64, P, c_int64, P, P, P, P, P, c_int64, P, c_size_t,
                                         P, c_size_t, P]),
    # ---- one scan, one call; streams ----
    'shg_scan_workspace_bytes': (c_size_t, [ctypes.POINTER(ScanRequest)]),
    'shg_scan_host_bytes': (c_size_t, [ctypes.POINTER(ScanRequest)]),
    'shg_scan_file': (c_int, [ctypes.POINTER(ScanRequest), ctypes.POINTER(ScanResult), P]),
    'shg_host_set_savgol_taps': (c_int, [P]),
    'shg_pool_create': (c_int, [P, c_int, P, c_int, ctypes.POINTER(c_void_p)]),
    'shg_pool_submit': (c_int, [P, ctypes.POINTER(ScanRequest), ctypes.POINTER(ScanResult), ctypes.POINTER(c_int64)]),
    'shg_pool_submit_after': (c_int, [P, ctypes.POINTER(ScanRequest), ctypes.POINTER(ScanResult), P, ctypes.POINTER(c_int64)]),
    'shg_scan_prelaunch': (c_int, [ctypes.POINTER(ScanRequest), P, ctypes.POINTER(c_int)]),
    'shg_pass_a_prelaunch': (c_int, [P, c_int64, c_int64, c_int64, c_int, c_int64, P, c_size_t, P, ctypes.POINTER(c_int)]),
    'shg_pass_a_forget': (c_int, [P]),
    'shg_pool_poll': (c_int, [P, c_int64]),
    'shg_pool_wait': (c_int, [P, c_int64, ctypes.POINTER(c_int), ctypes.c_char_p, c_size_t]),
    'shg_pool_destroy': (c_int, [P]),
    'shg_device_cu_count': (c_int, [ctypes.POINTER(c_int)]),
    'shg_stream_create': (c_int, [c_int, P, c_int, ctypes.POINTER(c_void_p)]),
    'shg_stream_destroy': (c_int, [P]),
    'shg_frame_pass_lane_set': (c_int, [P]),
    'shg_frame_pass_lane_get': (c_void_p, []),
    # ---- host control plane (host pointers; numpy arrays are passed by address) ----
    'shg_host_bind_lapack': (c_int, [P]),
    'shg_host_lapack_bound': (c_int, []),
    'shg_host_set_mode_pick': (c_int, [P]),
    'shg_host_polyfit3': (c_int, [P, P, c_int64, P]),
    'shg_host_detect_bord': (c_int, [P, c_int64, PI64, PI64]),
    'shg_host_line_fit': (c_int, [P, P, c_int64, c_int64, c_int64, ctypes.c_int32, P, P, P]),
    'shg_host_column_plan': (c_int, [P, c_int64, c_int64, P, c_int, P, P, P]),
    'shg_host_flood_threshold': (c_int, [c_double, c_int64, c_int64, c_double, c_double, P, PD]),
    'shg_host_limb_points': (c_int, [P, P, c_int64, c_int64, c_int64, P, PI64]),
    'shg_host_bind_blas': (c_int, [P, P, P, P, P]),
    'shg_host_blas_bound': (c_int, []),
    'shg_host_fit_ellipse': (c_int, [P, c_int64, P, PD, PD, PD]),
    'shg_host_correction_matrix': (c_int, [c_double, c_double, P, PD]),
    'shg_host_two_step': (c_int, [P, c_int64, P, PD, PD, PD, P, PI64, P]),
    'shg_host_warp_geometry': (c_int, [c_double, c_double, c_int64, c_int64, P, P, P, PD, PD, PI64, PI64]),
    'shg_host_limb_geometry': (c_int, [P, c_int64, c_int64, c_int64, P, P, P, PI64, P]),
    'shg_host_chord_bounds': (c_int, [c_double, c_double, c_double, c_double, c_double, c_int64, c_int64, c_int64, P, P]),
    'shg_host_transversalium_factors': (c_int, [P, P, c_int64, c_int64, P, c_int64, c_int, P]),
    'shg_host_percentile_plan': (c_int, [c_int64, c_double, PI64, PI64, PD]),
    'shg_host_lerp': (c_double, [c_double, c_double, c_double]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)      # AttributeError here = the library does not match include/shg_hip.h
    _fn.restype = _res
    _fn.argtypes = _args

ABI_VERSION = 16
if lib.shg_abi_version() != ABI_VERSION:
    raise ImportError('libshg_hip.so ABI %d != expected %d: rebuild it' % (lib.shg_abi_version(), ABI_VERSION))


def last_error():
    return lib.shg_last_error_string().decode('utf-8', 'replace')


def _host_error(status, what, message=None):
    """The host control plane reports the failure the reference's NumPy / SciPy call raises at that point
    (include/shg_hip.h, SHG_E_VALUE ... SHG_E_QHULL): raise that exception type."""
    msg = last_error() if message is None else message
    if status == -4:
        return ValueError(msg)
    if status == -5:
        return TypeError(msg)
    if status == -6:
        import numpy
        return numpy.linalg.LinAlgError(msg)
    if status == -10:
        return IndexError(msg)
    if status == -9:
        return AssertionError(msg)
    if status == -8:
        try:
            from scipy.spatial import QhullError
            return QhullError(msg)
        except ImportError:
            pass
    return RuntimeError('%s: %s' % (what, msg) if status != -7 else msg)


def check(status, what, message=None):
    """message: the error text when it was produced on another thread (a scan pool worker); default: this thread's last error."""
    if status != 0:
        if -10 <= status <= -4:
            raise _host_error(status, what, message)
        raise RuntimeError('%s failed (status %d): %s' % (what, status, last_error() if message is None else message))


# (library glob under site-packages, prefix, suffix of the ILP64 Fortran symbols, suffix of the ILP64 cblas symbols)
_OPENBLAS_LAYOUTS = (
    ('numpy.libs/libscipy_openblas64_*.so*', 'scipy_', '_64_', '64_'),       # NumPy 2.x wheels
    ('numpy.libs/libopenblas64_*.so*', '', '_64_', '64_'),                    # NumPy 1.2x wheels
    ('numpy/.libs/libopenblas64_*.so*', '', '_64_', '64_'),                   # older wheel layout
)


def _bind_numpy_lapack():
    """Hand the library the dgelsd NumPy itself calls (np.linalg.lstsq under np.polyfit) and the five BLAS / LAPACK entry
    points behind NumPy's matmul / inv / eig, so that the line fit and the limb geometry of the host control plane are
    bit-identical to the reference's on this host.  NumPy's wheels bundle an ILP64 OpenBLAS whose symbols carry a 64_ suffix
    (and, since 2.0, a scipy_ prefix).  A NumPy built against another BLAS (conda's MKL, a distribution's OpenBLAS) exports
    none of these: the library then uses its built-in routines (a Householder least squares, plain-loop 3 x 3 algebra), whose
    results agree with NumPy's to ~1e-12 but are not bit-identical -- said once, at import, with the names that were missing.
    -> (path of the library bound, or None; list of the symbols that could not be found)"""
    import glob
    import numpy
    import numpy.linalg           # noqa: F401  -- loads the bundled OpenBLAS
    root = os.path.dirname(os.path.dirname(os.path.abspath(numpy.__file__)))
    missing = ['dgelsd', 'cblas_dgemm', 'cblas_dsyrk', 'cblas_dgemv', 'dgesv', 'dgeev']
    for pattern, prefix, fsuffix, csuffix in _OPENBLAS_LAYOUTS:
        for path in sorted(glob.glob(os.path.join(root, pattern))):
            try:
                blas = ctypes.CDLL(path)               # already mapped: same handle, no second copy
            except OSError:
                continue
            found = {}
            for name in missing:
                sym = prefix + name + (csuffix if name.startswith('cblas_') else fsuffix)
                try:
                    found[name] = ctypes.cast(getattr(blas, sym), c_void_p)
                except AttributeError:
                    pass
            if 'dgelsd' not in found:
                continue
            lib.shg_host_bind_lapack(found['dgelsd'])
            rest = ['cblas_dgemm', 'cblas_dsyrk', 'cblas_dgemv', 'dgesv', 'dgeev']
            if all(n in found for n in rest):
                lib.shg_host_bind_blas(*[found[n] for n in rest])
            return path, [n for n in missing if n not in found]
    return None, missing


LAPACK_PATH, LAPACK_MISSING = _bind_numpy_lapack()
if LAPACK_MISSING:
    import warnings
    warnings.warn('solex_ser_recon_en_amd: NumPy\'s own %s could not be found (looked for the OpenBLAS bundled with NumPy\'s wheels); '
                  'the line fit / limb geometry run on the library\'s built-in routines: results agree with NumPy to ~1e-12 but the raw '
                  'disks and products are no longer guaranteed bit-identical to the reference\'s on this host.' % ', '.join(LAPACK_MISSING),
                  RuntimeWarning, stacklevel=2)


@ctypes.CFUNCTYPE(c_int64, ctypes.POINTER(c_int64), c_int64)
def _numpy_mode_pick(neg_counts, n):
    """np.argpartition(-counts, kth=2)[:2][0] by NumPy itself (solex_util.py:246): the one decision of the line fit whose
    outcome depends on NumPy's selection kernel.  Called once per scan from shg_host_line_fit (ctypes takes the
    interpreter lock for the call)."""
    try:
        import numpy
        a = numpy.ctypeslib.as_array(neg_counts, shape=(n,)).copy()
        return int(numpy.argpartition(a, kth=2)[:2][0])
    except Exception:      # noqa: BLE001 -- reported by the caller as an out-of-range pick
        return -1


lib.shg_host_set_mode_pick(ctypes.cast(_numpy_mode_pick, c_void_p))


_taps_error = {}          # window -> the exception SciPy raised inside the callback (re-raised by the caller of the C entry point)


@ctypes.CFUNCTYPE(c_int, c_int64, ctypes.POINTER(c_double))
def _savgol_taps(window, out):
    """scipy.signal.savgol_coeffs(window, 3) for shg_scan_file, when the window differs from the one the request carried
    taps for (a scan with few sunlit rows; solex_util.py:400).  SciPy's own exception is kept for the caller."""
    try:
        from .solex_util import savgol_taps
        taps = savgol_taps(int(window))
        ctypes.memmove(out, taps.ctypes.data, taps.size * 8)
        return 0
    except Exception as e:      # noqa: BLE001 -- the call may come from a scan pool thread: keyed by what was asked for
        _taps_error[int(window)] = e
        return -4 if isinstance(e, ValueError) else -7


lib.shg_host_set_savgol_taps(ctypes.cast(_savgol_taps, c_void_p))


def take_callback_error(window):
    """The exception the Savitzky-Golay callback met when it was asked for `window`, if any."""
    return _taps_error.pop(int(window), None)


def profile_enable(on=True, only=None):
    """only: iterable of kernel tags to time (None = all)."""
    lib.shg_profile_select(','.join(only).encode() if only else None)
    lib.shg_profile_enable(1 if on else 0)


def profile_reset():
    lib.shg_profile_reset()


def profile_get(tag):
    """-> (total milliseconds, launches) of the kernel `tag` since the last reset."""
    ms, n = c_double(0.0), c_int64(0)
    check(lib.shg_profile_get(tag.encode(), ctypes.byref(ms), ctypes.byref(n)), 'shg_profile_get')
    return ms.value, n.value


def profile_total():
    """-> (total milliseconds, entry points timed) over every tag since the last reset."""
    ms, n = c_double(0.0), c_int64(0)
    check(lib.shg_profile_total(ctypes.byref(ms), ctypes.byref(n)), 'shg_profile_total')
    return ms.value, n.value
