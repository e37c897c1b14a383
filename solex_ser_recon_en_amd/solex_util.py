"""Stage functions of the SHG pipeline on MI355X.

Same names, argument order and return tuples as the reference's solex_util.py
(SURVEY.md section 8b); the per-pixel / per-frame arithmetic runs in libshg_hip.so
(see ops.py), the 1-D control plane (polynomial fits, Savitzky-Golay trend, percentiles
from histograms) stays on the host with NumPy/SciPy exactly as the reference computes it.
Images are DeviceImage handles: they stay in HBM between stages.
"""
import datetime
import functools
import math
import os
import traceback

import numpy as np
import torch
from numpy.polynomial.polynomial import polyval

from . import dist, hostmath, ops, outputs, stages
from .device import DeviceImage, to_device_u16
from .fits_io import make_header, write_fits  # noqa: F401  (make_header is part of the surface)


# ---- log / path helpers (reference solex_util.py:29-63) ------------------------
def output_path(path, options):
    if options['output_dir'].strip() == '':
        return path
    return os.path.join(options['output_dir'], os.path.basename(path))


def _append(path, options, text, mode):
    try:
        with open(output_path(path, options), mode) as f:
            f.write(text)
    except Exception:
        traceback.print_exc()
        print('ERROR: failed to log file: ' + path)


def clearlog(path, options):
    if '_nolog' in options or '_log_off' in options:
        return      # the reference only guards logme; a file-less mode should not create an empty log either
    _append(path, options, 'start time: ' + str(datetime.datetime.now()) + '\n', 'w')


def write_complete(path, options):
    if '_nolog' in options or '_log_off' in options:
        return
    _append(path, options, 'end time: ' + str(datetime.datetime.now()) + '\n', 'a')


def logme(path, options, s):
    """s may be a callable building the line: formatting arrays costs more than the kernels it reports on."""
    if '_nolog' in options or '_log_off' in options:
        return
    _append(path, options, (s() if callable(s) else s) + '\n', 'a')


def plots_enabled(options):
    return not options['clahe_only'] and not options['protus_only'] and '_nolog' not in options


# ---- a2: mean and max frames (reference solex_util.py:174-188) -----------------------
def compute_mean_max(rdr, options, basefich0):
    logme(basefich0 + '_log.txt', options, 'Width, Height : ' + str(rdr.Width) + ' ' + str(rdr.Height))
    logme(basefich0 + '_log.txt', options, 'Number of frames : ' + str(rdr.FrameCount))
    stack = rdr.device_stack()
    n, h, w, bpp = ops.stack_geometry(stack)
    total, mx = ops.accumulate_sum_max(stack)
    if dist.is_sharded(rdr):
        total, mx = dist.exchange_frame_stats(total, mx, n_frames=n)     # integer SUM / MAX: bit-identical to one rank
    mean, mxo = ops.finalize_mean_max(total, mx, int(rdr.FrameCount), h, w, bpp)
    return DeviceImage(mean), DeviceImage(mxo)


# ---- a3: sunlit row range (reference solex_util.py:165-172) --------------------------
def detect_bord(img, axis):
    t = to_device_u16(img)
    if axis == 0:
        t = t.view(torch.int16).t().contiguous().view(torch.uint16)
    elif t.stride(0) != t.shape[1]:
        t = t.contiguous()
    ymean = ops.row_mean_u16(ops.box_blur_u16(t, 5, 5)).cpu().numpy()
    return hostmath.detect_bord(ymean)


# ---- a4: spectral line detection + cubic fit (reference solex_util.py:191-274) -------
def compute_mean_return_fit(vid_rdr, options, hdr, iw, ih, basefich0):
    """One stage call (shg_stage_mean_fit): pass A, mean / max images, detect_bord on the max image (:223-226), the
    blurred and sharp argmin traces (:228-242) and the cubic fits with their outlier logic (:233-259, C++ restatement
    of the NumPy calls, bit-identical `fit`)."""
    iw, ih = int(iw), int(ih)             # the reader exposes np.uint32 like the reference's
    logme(basefich0 + '_log.txt', options, 'Width, Height : ' + str(vid_rdr.Width) + ' ' + str(vid_rdr.Height))
    logme(basefich0 + '_log.txt', options, 'Number of frames : ' + str(vid_rdr.FrameCount))
    stack = vid_rdr.device_stack()
    plots = plots_enabled(options)
    if dist.is_sharded(vid_rdr):
        total, mx = ops.accumulate_sum_max(stack)
        total, mx = dist.exchange_frame_stats(total, mx, n_frames=int(stack.shape[0]))      # integer SUM / MAX: bit-identical to one rank
        res = stages.mean_fit(None, int(vid_rdr.FrameCount), sums=(total, mx), geometry=ops.stack_geometry(stack), want_plot_data=plots)
    else:
        res = stages.mean_fit(stack, int(vid_rdr.FrameCount), want_plot_data=plots)
    mean_img = DeviceImage(res['mean'])
    y1, y2, p, fit = res['y1'], res['y2'], res['p'], res['fit']
    if options['save_fit']:
        outputs.submit(write_fits, output_path(basefich0 + '_mean.fits', options), mean_img, hdr)
    logme(basefich0 + '_log.txt', options, 'Vertical limits y1, y2 : ' + str(y1) + ' ' + str(y2))
    logme(basefich0 + '_log.txt', options, lambda: 'Spectral line polynomial fit: ' + str(p))
    if plots:
        rows = np.arange(y1, y2)
        mask_good = res['mask_good']
        outputs.submit(outputs.plot_spectral_line, output_path(basefich0 + '_spectral_line_data.png', options),
                       mean_img, res['sharp'].astype(np.int64)[y1:y2][mask_good], rows[mask_good], fit[:, 3], ih, (y2 - y1) // 20 + 1)
    return mean_img, fit, y1, y2


# ---- a5: per-frame column extraction (reference solex_util.py:93-144) ---------------------
def extract_disks(rdr, fit, shifts, flip_x=False, want_minmax=False, owner=None):
    """-> uint16 GPU tensor [S, ih, FrameCount]; when sharded all ranks hold the full mosaic (owner None) or rank `owner`
    alone does.  One stage call (shg_stage_extract): sample columns and weights from `fit` (:113-123), upload, the extraction kernel.
    want_minmax: -> (disks, extrema slots int32 [S, 2] or None for a sharded scan)."""
    stack = rdr.device_stack()
    n_total = int(rdr.FrameCount)
    if dist.is_sharded(rdr):
        mosaic = dist.gather_columns(lambda out, k0: stages.extract(stack, fit, shifts, n_cols=n_total, k_offset=k0, flip_x=flip_x, out=out),
                                     len(shifts), int(rdr.ih), rdr.frame_range, n_total, flip_x, stack.device, dst=owner)
        return (mosaic, None) if want_minmax else mosaic
    return stages.extract(stack, fit, shifts, n_cols=n_total, k_offset=0, flip_x=flip_x, want_minmax=want_minmax)


def read_video_improved(rdr, fit, options):
    disks, mm = extract_disks(rdr, fit, options['shift'], want_minmax=True)
    return [DeviceImage(disks[i], minmax=None if mm is None else mm[i]) for i in range(disks.shape[0])], rdr.ih, rdr.iw, rdr.FrameCount


# ---- a9: transversalium (reference solex_util.py:76-86, 383-516) ---------------------------
def reject_outliers(data, m=2):
    median_value = np.median(data)
    d = np.abs(data - median_value)
    mdev = np.median(d)
    s = d / mdev if mdev else np.zeros(len(d))
    return data[s < m]


@functools.lru_cache(maxsize=64)
def _tukey_cached(n, a):
    t = _tukey_compute(n, a)
    t.setflags(write=False)
    return t


def _tukey(n, a=0.05):
    return _tukey_cached(int(n), float(a))


def _tukey_compute(n, a=0.05):
    """The reference's piecewise taper t(x), x = 0..n-1 (solex_util.py:460-470), evaluating
    math.cos only on the two ramps (everything in between is exactly 1)."""
    def ramp(x):
        return 1 / 2 * (1 - math.cos(2 * math.pi * x / (a * n)))
    taper = np.ones(n)
    for x in range(n):
        if x < a * n / 2:
            taper[x] = ramp(x)
        else:
            break
    for x in range(n - 1, -1, -1):
        # x > n/2 mirrors to t(n - x); the plateau test a*n/2 <= x <= n/2 comes first in the reference
        if x > n / 2 and (n - x) < a * n / 2:
            taper[x] = ramp(n - x)
        else:
            break
    return taper


@functools.lru_cache(maxsize=32)
def savgol_taps(window):
    from scipy.signal import savgol_coeffs
    taps = savgol_coeffs(window, 3)
    taps.setflags(write=False)
    return taps


def savgol_window(n, trans_strength):
    return min(trans_strength, n // 2 * 2 - 1)                    # solex_util.py:400


# ---- stubborn transversalium: host control plane of solex_util.py:277-375, 415-423 ------------------
LIN_LEN, LIN_HALF_WIDTH, LIN_EDGE_FUDGE = 101, 5, 20          # apply_lin_filter(img, 101, 5, ...), fix_edge_effect(..., linlen + 20)


def _spurious_rows(correction, n_rows, y1, y2):
    """Rows whose accumulated log-correction exceeds 2.5 sigma, plus their two neighbours (np.roll wraps), :416-421."""
    log_corr = np.log(correction)
    c = np.zeros(n_rows)
    c[y1:y2] = log_corr
    flag = np.abs(c) > np.std(log_corr) * 2.5
    return flag | np.roll(flag, -1) | np.roll(flag, 1)


def _nearest_unflagged(flag):
    """(up, dn): for every row the index of the nearest unflagged row above / below, -1 where there is none --
    the two sweeps of :306-317 (a flagged row becomes prev/2 + next/2) as index maps."""
    n = flag.shape[0]
    idx = np.arange(n)
    up = np.maximum.accumulate(np.where(flag, -1, idx))
    dn = np.minimum.accumulate(np.where(flag, n, idx)[::-1])[::-1]
    return up.astype(np.int32), np.where(dn == n, -1, dn).astype(np.int32)


def _limb_edge_plan(circle, h, w, linlen):
    """fix_edge_effect (:356-375) as per-row data for shg_lin_filter_apply: keep delta on [xa, xb); edge bit 0 =
    copy column xa + half over [xa, xa + half), bit 1 = copy column xb - half - 1 over [xb - half, xb)."""
    y1 = math.ceil(max(circle[1] - circle[2], 0))
    y2 = math.floor(min(circle[1] + circle[2], h - 1))
    xa = np.zeros(h, dtype=np.int32)
    xb = np.zeros(h, dtype=np.int32)
    edge = np.zeros(h, dtype=np.uint8)
    if y2 > y1:
        ys = np.arange(y1, y2, dtype=np.float64)
        v = circle[2] ** 2 - (ys - circle[1]) ** 2
        if np.any(v < 0):
            raise TypeError('stubborn transversalium: row outside the disk circle (complex chord length)')
        root = np.sqrt(v)
        dx = np.floor(root)
        for i in np.flatnonzero(np.abs(root - np.rint(root)) <= 1e-9 * np.maximum(root, 1.0)):
            dx[i] = math.floor(float(v[i]) ** 0.5)                # Python's own pow where floor() could differ
        x2 = np.floor(np.minimum(circle[0] + dx, w - 1)).astype(np.int64)
        x1 = np.ceil(np.maximum(circle[0] - dx, 0)).astype(np.int64)
        if np.any(x1 < 0) or np.any(x2 < 0):
            raise ValueError('stubborn transversalium: the circle lies outside the image')
        wide = (x2 - x1) >= linlen
        xa[y1:y2] = x1
        xb[y1:y2] = np.maximum(x2, x1)
        edge[y1:y2] = (wide & (x1 > 0)).astype(np.uint8) | ((wide & (x2 < w - 1)).astype(np.uint8) << 1)
    if 0 <= y1 <= y2 < h:
        xa[y2], xb[y2] = 0, w            # row y2 is neither in the loop nor in the cleared tail: it keeps its delta
    return xa, xb, edge, linlen // 2


def correct_transversalium2_batch(imgs, circle, borders, options, reqFlag, basefichs):
    """correct_transversalium2 for several frames that share circle / borders / shape (the disks of a Doppler
    stack): all row statistics are launched before the single device->host read, and the 1-D control plane runs
    once on the [k, n] matrix."""
    factors = [img.row_factor if isinstance(img, DeviceImage) else None for img in imgs]       # de-vignetted (float64) frames
    tensors = [img.t if rf is not None else to_device_u16(img) for img, rf in zip(imgs, factors)]
    h, w = tensors[0].shape
    if any(t.shape != (h, w) for t in tensors):
        raise ValueError('correct_transversalium2_batch: the frames must share one shape')
    y1 = math.ceil(max(circle[1] - circle[2], borders[1]))
    y2 = math.floor(min(circle[1] + circle[2], borders[3]))
    n_rows = max(y2 - y1, 1)
    window = savgol_window(n_rows, options['trans_strength'])
    taps = savgol_taps(window)                       # raises ValueError where scipy.signal.savgol_filter would
    if y2 - y1 >= 1:
        xa, xb = hostmath.chord_bounds(circle, borders, y1, y2, w)
        bounds_d = torch.from_numpy(np.stack([xa, xb])).to(tensors[0].device)          # one upload for both bound vectors
        xa_d, xb_d = bounds_d[0], bounds_d[1]
        stats = torch.stack([ops.rowpair_logratio_stats(t, y1, y2, xa_d, xb_d, rf) for t, rf in zip(tensors, factors)])
        interior = None
        if 3 < window <= stats.shape[1] and window // 2 <= 1024:         # shg_correlate1d_rows_f64 stages 2R+1 weights in LDS
            # the interior of the Savitzky-Golay trend while the statistics are still on the GPU (SciPy's own order of
            # operations); its two edges are LAPACK fits and stay on the host
            both = torch.stack([stats, ops.correlate1d_rows_f64(stats, taps[::-1])]).cpu().numpy()
            ratios, interior = both[0], both[1]
        else:
            ratios = stats.cpu().numpy()
    else:
        ratios, interior = np.zeros((len(tensors), 1)), None                           # y_ratios_r = [0], :386
    if options.get('stubborn_transversalium'):
        # :415-423: rows the smooth correction cannot follow are rebuilt from their neighbours by a line filter;
        # no correction plot and no '_transversalium_cache' on this branch
        correction = hostmath.transversalium_factors(ratios, interior, taps, tapered=False)
        taper = np.zeros(h)
        taper[y1:y2] = _tukey(y2 - y1)
        xa_e, xb_e, edge, edge_half = _limb_edge_plan(circle, h, w, LIN_LEN + LIN_EDGE_FUDGE)
        out = []
        for i, (t, rf) in enumerate(zip(tensors, factors)):
            flag = _spurious_rows(correction[i], h, y1, y2)
            up, dn = _nearest_unflagged(flag)
            out.append(DeviceImage(ops.lin_filter_u16(t, flag, up, dn, taper, xa_e, xb_e, edge, edge_half, LIN_LEN,
                                                      LIN_HALF_WIDTH, rf)))
        return out
    correction_t = hostmath.transversalium_factors(ratios, interior, taps, tapered=True)
    out = []
    for i, (t, rf) in enumerate(zip(tensors, factors)):
        c = np.ones(h)
        c[y1:y2] = correction_t[i]
        options['_transversalium_cache'] = c
        if (not reqFlag) and plots_enabled(options):
            outputs.submit(outputs.plot_transversalium, output_path(basefichs[i] + '_transversalium_correction.png', options), c)
        out.append(DeviceImage(ops.scale_rows_u16(t, c, rf)))
    return out


def correct_transversalium2(img, circle, borders, options, reqFlag, basefich):
    return correct_transversalium2_batch([img], circle, borders, options, reqFlag, [basefich])[0]


# ---- removeVignette (reference solex_util.py:590-654) ----------------------------------------------
VIGNETTE_MARGIN = 65          # pixels inside the limb where the brightness profiles start


def _profile_inside_disk(profile, centre, radius):
    """The samples of a 1-D brightness profile that lie VIGNETTE_MARGIN pixels inside the disk, and the offset (in pixels
    from int(centre)) of the first one."""
    first = max(0, int(centre - radius + VIGNETTE_MARGIN))
    stop = min(profile.shape[0], int(centre + radius + 1 - VIGNETTE_MARGIN))
    return profile[first:stop], first - int(centre)


def _fill_gaps(values):
    """NaN runs take the last finite value before them, leading NaNs the first finite value (forward, then backward fill)."""
    def forward(v):
        last_seen = np.maximum.accumulate(np.where(np.isnan(v), 0, np.arange(v.shape[0])))
        return v[last_seen]
    return forward(forward(values)[::-1])[::-1]


def removeVignette(frame_circularized, cercle0):
    """Flat-field the disk along the slit: the horizontal and the vertical brightness profile of a round sun should
    agree, so their ratio at equal distance from the centre is the row gain to take out (reference :590-654).
    Returns the reference's float64 image frame * correction_factor[:, None], held as (uint16 image, float64 row factor)
    on the GPU; or the input itself when there is too little data."""
    from scipy.ndimage import gaussian_filter1d
    from scipy.signal import savgol_filter
    from .order_stats import lerp_order_stats
    t = to_device_u16(frame_circularized)
    h, w = t.shape
    cx, cy, radius = cercle0
    # np.percentile(frame, 85, axis): two order statistics per line on the GPU, NumPy's lerp on the host
    lo0, hi0, mix0 = lerp_order_stats(h, 85)
    lo1, hi1, mix1 = lerp_order_stats(w, 85)
    cols = torch.stack(ops.line_order_stats_u16(t, 0, lo0, hi0)).view(torch.int16).cpu().numpy().view(np.uint16).astype(np.float64)
    rows = torch.stack(ops.line_order_stats_u16(t, 1, lo1, hi1)).view(torch.int16).cpu().numpy().view(np.uint16).astype(np.float64)
    along_x, x_offset = _profile_inside_disk(mix0(cols[0], cols[1]), cx, radius)       # _lerp is elementwise
    along_y, y_offset = _profile_inside_disk(mix1(rows[0], rows[1]), cy, radius)
    if along_x.shape[0] < 20 or along_y.shape[0] < 20:
        print("no de-vignette, due to not enough data")
        return frame_circularized
    print("vignette shapes:", along_x.shape, along_y.shape)
    span = int(min(along_x.shape[0] // 2.75, along_y.shape[0] // 2.75)) // 2 * 2 - 1
    smooth_x = savgol_filter(along_x, min(801, span), 3)
    smooth_y = savgol_filter(along_y, min(801, span), 3)
    # both profiles on one axis: distance from the disk centre
    nearest = min(x_offset, y_offset)
    farthest = max(x_offset + along_x.shape[0], y_offset + along_y.shape[0])
    horizontal = np.full(farthest - nearest, np.nan)
    vertical = np.full(farthest - nearest, np.nan)
    horizontal[x_offset - nearest: x_offset - nearest + along_x.shape[0]] = smooth_x
    vertical[y_offset - nearest: y_offset - nearest + along_y.shape[0]] = smooth_y
    with np.errstate(divide='ignore', invalid='ignore'):
        gain = horizontal / vertical
    gain[(horizontal == 0) | (vertical == 0)] = np.nan
    row_gain = np.full(h, np.nan)
    row_gain[np.arange(nearest, farthest) + int(cy)] = gain          # (negative rows wrap, as the reference's fancy index does)
    row_gain = gaussian_filter1d(_fill_gaps(row_gain), max(2, min(150, span // 4)))
    return DeviceImage(t, row_factor=torch.from_numpy(np.ascontiguousarray(row_gain)).to(t.device))


def as_uint16_image(img):
    """frame.astype(np.uint16) (solex_util.py:528) for a factored float64 frame: trunc(img * row_factor)."""
    if isinstance(img, DeviceImage) and img.row_factor is not None:
        ones = torch.ones(img.t.shape[0], dtype=torch.float64, device=img.t.device)
        return DeviceImage(ops.scale_rows_u16(img.t, ones, img.row_factor))
    return img


# ---- a12: rescale_brightness (reference solex_util.py:519-525) ------------------------------
def rescale_brightness(img, lo, hi, alpha=1.0):
    if getattr(img, 'dtype', None) in (torch.uint8, np.uint8):
        # 8-bit images only reach this from clahe_apply.py: sat = 255; a NumPy image comes back as NumPy
        assert 255 >= hi > lo
        if isinstance(img, np.ndarray):
            from .device import default_device
            return ops.rescale_u8(torch.from_numpy(np.ascontiguousarray(img)).to(default_device()), lo, hi, alpha).cpu().numpy()
        return ops.rescale_u8(img, lo, hi, alpha)
    assert 65535 >= hi > lo                                                            # :521
    return DeviceImage(ops.rescale_u16(to_device_u16(img), lo, hi, alpha))


def percentile_from_hist(hist, q):
    """np.percentile(values, q) (method 'linear') from the exact histogram of integer values.
    Follows NumPy's _quantile: virtual index (n-1)*(q/100), neighbours by order statistic, _lerp."""
    hist = np.asarray(hist, dtype=np.int64)
    n = int(hist.sum())
    if n == 0:
        raise ValueError('percentile of an empty image')
    cum = np.cumsum(hist)
    quantile = np.true_divide(q, 100)
    virtual = (n - 1) * quantile
    prev = math.floor(virtual)
    gamma = virtual - prev
    prev = min(max(prev, 0), n - 1)
    nxt = min(prev + 1, n - 1)
    a = float(np.searchsorted(cum, prev + 1))            # value of the prev-th order statistic (0-based)
    b = float(np.searchsorted(cum, nxt + 1))
    diff = b - a
    return b - diff * (1 - gamma) if gamma >= 0.5 else a + diff * gamma


def max_from_hist(hist):
    return int(np.flatnonzero(np.asarray(hist))[-1])


def _rot90(t, k):
    k %= 4
    if k == 0:
        return t
    return torch.rot90(t.view(torch.int16), k, dims=(0, 1)).contiguous().view(torch.uint16)


# ---- a11: CLAHE + contrast products + writers (reference solex_util.py:527-588) ---------------
def image_process_batch(frames, cercle, options, header, basefichs):
    """image_process for several frames of one shape: CLAHE and the order statistics of every frame are launched
    before the single device->host read of their 5 scalars each."""
    from .order_stats import lerp_order_stats
    tensors = [to_device_u16(as_uint16_image(f)) for f in frames]                       # frame.astype(np.uint16), :528
    stats = torch.empty((len(tensors), 5), dtype=torch.float64, device=tensors[0].device)
    cl1s = []
    for i, frame_t in enumerate(tensors):
        n_px = frame_t.shape[0] * frame_t.shape[1]
        b_lo, b_hi, b_mix = lerp_order_stats(n_px, 99.9999)
        d_lo, d_hi, d_mix = lerp_order_stats(n_px, 10)
        # CLAHE, then the two order statistics np.percentile(frame, 99.9999) needs and, on the CLAHE image, those of
        # np.percentile(cl1, 10) and np.max (the last one): one C call
        cl1 = ops.contrast_stats_u16(frame_t, [b_lo, b_hi], [d_lo, d_hi, n_px - 1], stats[i])
        cl1s.append((cl1, b_mix, d_mix))
    stats = stats.cpu().numpy()
    results = []
    for i, (frame_t, (cl1, b_mix, d_mix)) in enumerate(zip(tensors, cl1s)):
        basefich = basefichs[i]
        bright = b_mix(stats[i, 0], stats[i, 1])                        # basically the same as max
        dark_clahe = d_mix(stats[i, 2], stats[i, 3])
        bright_clahe = int(stats[i, 4])
        frame_raw = frame_t
        assert 65535 >= bright > bright * 0.25 and 65535 >= bright * 0.18 > 0 and 65535 >= bright_clahe > dark_clahe
        disc = None
        if not cercle == (-1, -1, -1) and options['disk_display']:
            r = int(cercle[2]) + options['delta_radius']
            if r > 0:
                disc = (int(cercle[0]), int(cercle[1]), r)
        # frame_HC = rescale(frame, .25 bright, bright); frame_protus = rescale(frame, 0, .18 bright) + cv2.circle(80);
        # cc = rescale(cl1, dark_clahe, bright_clahe) (:539-547): one C call
        frame_HC, frame_protus, cc = ops.contrast_products_u16(
            frame_t, cl1, [bright * 0.25, bright, 0, bright * 0.18, dark_clahe, bright_clahe], disc)

        results.append(write_products(frame_raw, cl1, frame_HC, frame_protus, cc, options, header, basefich))
    return results


def write_products(frame_raw, cl1, frame_HC, frame_protus, cc, options, header, basefich):
    """The tail of image_process (solex_util.py:550-588): rot90 by img_rotate, the four PNGs and the CLAHE FITS (background
    encoders).  -> (cc, frame_protus)"""
    k = options['img_rotate'] // 90
    frame_raw, frame_HC, frame_protus, cc = (_rot90(x, k) for x in (frame_raw, frame_HC, frame_protus, cc))
    if '_nolog' not in options:
        if options['clahe_only'] or not options['protus_only']:
            print('saving image to:' + basefich + '_clahe.png')
            outputs.submit(outputs.write_png16, output_path(basefich + '_clahe.png', options), DeviceImage(cc))
        if options['protus_only'] or not options['clahe_only']:
            outputs.submit(outputs.write_png16, output_path(basefich + '_protus.png', options), DeviceImage(frame_protus))
        if not options['clahe_only'] and not options['protus_only']:
            outputs.submit(outputs.write_png16, output_path(basefich + '_uncontrasted.png', options), DeviceImage(frame_raw))
            outputs.submit(outputs.write_png16, output_path(basefich + '_high_contrast.png', options), DeviceImage(frame_HC))
    if options['save_fit']:
        outputs.submit(write_fits, output_path(basefich + '_clahe.fits', options), DeviceImage(cl1), header)
    return DeviceImage(cc), DeviceImage(frame_protus)


def image_process(frame, cercle, options, header, basefich):
    return image_process_batch([frame], cercle, options, header, [basefich])[0]
