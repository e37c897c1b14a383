"""The oracle (oracle/shg_oracle.py) against vectors produced by the reference's
own functions (oracle/capture_goldens.py).  CPU only."""
import numpy as np
import pytest

from oracle import shg_oracle as orc


@pytest.mark.parametrize('tag', ['u16_rot', 'u16_norot', 'u8_rot', 'u8_norot', 'u16_odd'])
def test_mean_max_bit_exact(golden, tag):
    g = golden('g1_mean_max')
    rdr = orc.SerReader(g[tag + '_frames'])
    ih, iw, n, rot = g[tag + '_dims']
    assert (rdr.ih, rdr.iw, rdr.FrameCount, int(rdr.flag_rotate)) == (ih, iw, n, rot)
    mean, mx = orc.compute_mean_max(rdr)
    assert mean.dtype == np.uint16 and mx.dtype == np.uint16
    np.testing.assert_array_equal(mean, g[tag + '_mean'])
    np.testing.assert_array_equal(mx, g[tag + '_max'])


@pytest.mark.parametrize('tag', ['u16_rot', 'u16_norot', 'u8_rot', 'u16_odd'])
@pytest.mark.parametrize('stag', ['s2', 's21', 's3'])
def test_extract_bit_exact(golden, tag, stag):
    g = golden('g2_extract')
    shifts = [int(s) for s in g[tag + '_' + stag + '_shifts']]
    disks = orc.extract_columns(orc.SerReader(g[tag + '_frames']), g[tag + '_fit'], shifts)
    np.testing.assert_array_equal(np.stack(disks), g[tag + '_' + stag + '_disks'])


def test_shift_list_order():
    # Solex_recon.py:55
    assert orc.shift_list(10, [0]) == [10, 0]
    assert orc.shift_list(10, list(range(-10, 11))) == [10, 0] + [s for s in range(-10, 11) if s not in (10, 0)]
    assert orc.shift_list(10, [10]) == [10, 0]
    assert orc.shift_list(0, [3]) == [0, 3]


def test_correction_matrix(golden):
    g = golden('g7_matrix')
    for (phi, r), mat, theta in zip(g['params'], g['mats'], g['thetas']):
        m, t = orc.correction_matrix(phi, r)
        np.testing.assert_allclose(m, mat, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(t, theta, rtol=1e-13, atol=1e-15)
        assert m[1, 0] == 0 and m[1, 1] == 1


def test_warp_bit_exact(golden):
    g = golden('g3_warp')
    img = g['image_u16'] / 65536
    for i in range(6):
        phi, ratio, cx, cy, height = g['c%d_params' % i]
        out, circle, mat3 = orc.correct_image(img, phi, ratio, np.array([cx, cy]), height)
        assert out.shape == g['c%d_out' % i].shape
        np.testing.assert_array_equal(out, g['c%d_out' % i])
        np.testing.assert_allclose(circle, g['c%d_circle' % i], rtol=1e-12)
        np.testing.assert_allclose(mat3, g['c%d_mat3' % i], rtol=1e-12, atol=1e-14)
    out, circle, _ = orc.correct_image(img, 0.0, 1.0, np.array([-1.0, -1.0]), -1.0)
    np.testing.assert_array_equal(out, g['noellipse_out'])


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_transversalium(golden, tag):
    g = golden('g4_transversalium')
    out, c = orc.correct_transversalium2(g['image'], tuple(g['circle']), list(g['borders']), int(g[tag + '_strength']))
    np.testing.assert_allclose(c, g[tag + '_c'], rtol=1e-12)
    # the factors agree to ~1e-15 across SciPy versions; a truncation flip needs img*c within that of an integer
    assert np.count_nonzero(out != g[tag + '_out']) == 0


def test_transversalium_backup_bounds(golden):
    g = golden('g4_transversalium')
    out, c = orc.correct_transversalium2(g['image'], (0, 0, 99999), list(g['bb_borders']), 301)
    np.testing.assert_allclose(c, g['bb_c'], rtol=1e-12)
    np.testing.assert_array_equal(out, g['bb_out'])


def test_rescale_and_percentile(golden):
    g = golden('g5_rescale')
    img = g['image']
    bright = np.percentile(img, 99.9999)
    assert bright == float(g['bright'])
    assert np.percentile(img, 10) == float(g['p10'])
    np.testing.assert_array_equal(orc.rescale_brightness(img, bright * 0.25, bright), g['hc'])
    np.testing.assert_array_equal(orc.rescale_brightness(img, 0, bright * 0.18), g['protus'])
    np.testing.assert_array_equal(orc.rescale_brightness(img, 1000.0, 50000.0, alpha=0.8), g['alpha'])
    np.testing.assert_array_equal(orc.rescale_brightness(g['image8'], 10.0, 200.0), g['u8'])


@pytest.mark.parametrize('tag', ['u16_rot', 'u8_norot'])
def test_line_fit_shim_mode(golden, tag):
    """Reference compute_mean_return_fit ran unmodified with cv2.blur := orc.box_blur_u16."""
    g = golden('g8_fit_shim')
    mean, mx = orc.compute_mean_max(orc.SerReader(g[tag + '_frames']))
    np.testing.assert_array_equal(mean, g[tag + '_mean'])
    np.testing.assert_array_equal(mx, g[tag + '_max'])
    fit, y1, y2, p, _ = orc.line_fit(mean, mx)
    assert (y1, y2) == tuple(g[tag + '_y'])
    ref = g[tag + '_fit']
    away = np.abs(ref[:, 3] - np.rint(ref[:, 3])) > 1e-9      # floor() is only stable away from integers
    np.testing.assert_array_equal(fit[away, 0], ref[away, 0])
    np.testing.assert_array_equal(fit[:, 2], ref[:, 2])
    # np.polyfit differs by ~1e-13 between NumPy builds (LAPACK); curve values agree to that level
    np.testing.assert_allclose(fit[:, 3], ref[:, 3], rtol=0, atol=1e-9)
    np.testing.assert_allclose(fit[away, 1], ref[away, 1], rtol=0, atol=1e-9)


@pytest.mark.parametrize('source', ['g8_u16_rot', 'g8_u8_norot', 'g14_scan', 'c2_like', 'wide_sensor'])
def test_line_fit_does_not_depend_on_opencvs_simd_width(golden, source):
    """cv2.blur scales its window sums in float32 for the SIMD lanes and in double for the scalar tail; where that
    boundary lies depends on the vector unit OpenCV was dispatched to (8 uint16 lanes on the baseline, 16 with AVX2, 32
    with AVX-512).  The blurred images only steer decisions -- the sunlit rows and the two arg-min traces -- so the fit
    (and with it every raw disk) must be the same for every width: y1, y2, both traces, the mask and `fit` bit for bit."""
    import functools
    from solex_ser_recon_en_amd import synth
    if source.startswith('g8_'):
        g = golden('g8_fit_shim')
        tag = source[3:]
        mean, mx = g[tag + '_mean'], g[tag + '_max']
    else:
        if source == 'g14_scan':
            g = golden('g14_pipeline')
            frames = synth.synth_frames_numpy(int(g['param_n']), int(g['param_w']), int(g['param_h']), int(g['param_bits']),
                                              seed=int(g['param_seed']), tilt=float(g['param_tilt']), curv=float(g['param_curv']),
                                              row_gain=g['row_gain'])
        elif source == 'c2_like':
            frames = synth.synth_frames_numpy(120, 2000, 200, 16, seed=0)       # C2's frame shape (iw = 200 = 25 * 8: tails differ)
        else:
            frames = synth.synth_frames_numpy(60, 1203, 157, 16, seed=2, tilt=0.004, curv=9e-6)    # iw = 157: ragged in every width
        mean, mx = orc.compute_mean_max(orc.SerReader(frames))
    base = orc.line_fit(mean, mx)
    for lanes in (0, 16, 32, 64):
        blur = functools.partial(orc.box_blur_u16, simd_lanes=lanes)
        fit, y1, y2, p, aux = orc.line_fit(mean, mx, blur=blur)
        assert (y1, y2) == (base[1], base[2])
        np.testing.assert_array_equal(aux['min_intensity'], base[4]['min_intensity'])
        np.testing.assert_array_equal(aux['sharp'], base[4]['sharp'])
        np.testing.assert_array_equal(aux['mask_good'], base[4]['mask_good'])
        np.testing.assert_array_equal(fit, base[0])
    # the blurred images themselves do differ between widths on a handful of pixels, by one grey level
    kh = max(1, int((base[2] - base[1]) * 0.01))
    d = orc.box_blur_u16(mean, 25, kh, simd_lanes=0).astype(int) - orc.box_blur_u16(mean, 25, kh, simd_lanes=64).astype(int)
    assert np.abs(d).max() <= 1


# ---- known-answer tests for the UNPINNED third-party primitives ------------
def test_box_blur_known_answers():
    img = np.zeros((9, 16), np.uint16)
    img[4, 8] = 900
    b = orc.box_blur_u16(img, 3, 3)
    assert b[3:6, 7:10].tolist() == [[100] * 3] * 3 and b.sum() == 900
    const = np.full((12, 24), 1234, np.uint16)
    assert (orc.box_blur_u16(const, 25, 7) == 1234).all()         # reflect-101 borders keep a constant
    ramp = np.tile(np.arange(32, dtype=np.uint16) * 10, (6, 1))
    b = orc.box_blur_u16(ramp, 5, 1)
    assert (b[:, 2:-2] == ramp[:, 2:-2]).all()                    # symmetric window on a ramp
    assert b[0, 0] == round((0 + 10 + 20 + 10 + 20) / 5)          # reflect-101: cols -2,-1 -> 2,1
    # even kernel: anchor k//2 -> window [-2, 1]
    b = orc.box_blur_u16(ramp, 4, 1)
    assert b[0, 10] == round((80 + 90 + 100 + 110) / 4)


def test_clahe_known_answers():
    const = np.full((40, 60), 30000, np.uint16)
    out = orc.clahe(const, 0.8, 2)
    assert len(np.unique(out)) == 1
    rng = np.random.default_rng(0)
    img = rng.integers(0, 65536, (64, 64)).astype(np.uint16)
    out = orc.clahe(img, 0.8, 2)
    assert out.dtype == np.uint16 and out.shape == img.shape
    # with clip = 1 every bin is <= 1 + redistribution: the LUT is ~the identity ramp
    assert np.abs(out.astype(int) - img.astype(int)).max() < 4000
    img8 = rng.integers(0, 256, (31, 45)).astype(np.uint8)       # ragged: padded tiles
    out8 = orc.clahe(img8, 0.8, 3)
    assert out8.dtype == np.uint8 and out8.shape == img8.shape
    # monotone: CLAHE LUTs are non-decreasing, and so is the blend of them at one pixel position
    a = np.full((32, 32), 100, np.uint16); b = a.copy(); b[5, 5] = 200
    assert orc.clahe(b, 0.8, 2)[5, 5] >= orc.clahe(a, 0.8, 2)[5, 5]


def test_filled_circle_known_answers():
    img = np.zeros((21, 21), np.uint16)
    orc.filled_circle(img, 10, 10, 5, 80)
    assert img[10, 5] == 80 and img[10, 15] == 80 and img[10, 4] == 0 and img[5, 10] == 80 and img[4, 10] == 0
    assert (img == img.T).all() and (img == img[::-1]).all()
    area = int((img == 80).sum())
    assert abs(area - np.pi * 25) < 12
    img = np.zeros((10, 10), np.uint16)
    orc.filled_circle(img, 0, 0, 4, 7)       # clipped at the image corner
    assert img[0, 0] == 7 and img[0, 4] == 7 and img[0, 5] == 0


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_remove_vignette_pinned(golden, tag):
    g = golden('g6_vignette')
    img = g[tag + '_image']
    out = orc.remove_vignette(img, tuple(g[tag + '_circle']))
    assert out.dtype == np.float64 and out.shape == tuple(g[tag + '_out_sha256_shape'])
    # savgol / gaussian_filter1d agree to ~1e-15 across SciPy versions
    np.testing.assert_allclose(out[:, 0] / img[:, 0], g[tag + '_factor'], rtol=1e-12)
    np.testing.assert_allclose(out[::17, ::13], g[tag + '_row_sample'], rtol=1e-12)
    np.testing.assert_array_equal(np.percentile(img, 85, axis=0), g[tag + '_p85_cols'])
    small = np.full((90, 90), 1000, np.uint16)
    assert orc.remove_vignette(small, (45.0, 45.0, 40.0)) is small


def _crop_cases(g):
    for k in range(int(g['n'])):
        cx, cy, r, fw, sq = g['args_%d' % k]
        cercle = (-1, -1, -1) if cx == -1 else (float(cx), float(cy), float(r))
        yield k, g['in_%d' % k], cercle, (None if fw < 0 else int(fw)), bool(sq)


def test_crop_center_vs_reference(golden):
    """G11: the crop / pad block of single_image_process (Solex_recon.py:155-171), reference run unmodified."""
    g = golden('g11_crop')
    for k, img, cercle, fw, sq in _crop_cases(g):
        out, c2 = orc.crop_center(img, cercle, fw, sq)
        np.testing.assert_array_equal(out, g['out_%d' % k], err_msg='case %d' % k)
        np.testing.assert_array_equal(np.array(c2, dtype=np.float64), g['cercle_%d' % k], err_msg='case %d' % k)


def test_shift_order_and_raw_disks_vs_reference(golden):
    """G12: options['shift'] after solex_read (Solex_recon.py:53-55) and the disks it returns, in that order."""
    import hashlib
    from solex_ser_recon_en_amd import synth
    g = golden('g12_shift_order')
    frames = synth.synth_frames_numpy(int(g['param_n']), int(g['param_w']), int(g['param_h']), int(g['param_bits']),
                                      seed=int(g['param_seed']), tilt=float(g['param_tilt']), curv=float(g['param_curv']))
    assert hashlib.sha256(frames.tobytes()).digest() == g['frames_sha256'].tobytes()
    rdr = orc.SerReader(frames)
    mean, mx = orc.compute_mean_max(rdr)
    fit, y1, y2, _, _ = orc.line_fit(mean, mx)
    for k in range(int(g['n'])):
        shifts = orc.shift_list(int(g['efs_%d' % k]), [int(s) for s in g['request_%d' % k]])
        assert shifts == [int(s) for s in g['shift_%d' % k]]
        assert (y1, y2) == tuple(int(b) for b in g['bounds_%d' % k])
        disks = orc.extract_columns(orc.SerReader(frames), fit, shifts)
        got = np.stack([np.frombuffer(hashlib.sha256(np.ascontiguousarray(d).tobytes()).digest(), np.uint8) for d in disks])
        np.testing.assert_array_equal(got, g['disk_sha256_%d' % k])


def close_u16(got, want, max_flips):
    assert got.shape == want.shape and got.dtype == want.dtype
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert d.max() <= 1 and np.count_nonzero(d) <= max_flips, (d.max(), np.count_nonzero(d))


def test_stubborn_transversalium_vs_reference_shim(golden):
    """G15: the reference's stubborn branch run unmodified with cv2.filter2D := scipy.ndimage.correlate (mirror).
    The restatement sums in float64 and rounds once; scipy rounds the products first: a last-bit difference of
    delta can flip a truncation, hence <= 1 LSB on a handful of the 99 000 pixels."""
    g = golden('g15_stubborn')
    img, circle, borders = g['image'], tuple(g['circle']), list(g['borders'])
    out, flag = orc.correct_transversalium2_stubborn(img, circle, borders, 301)
    assert flag.sum() >= 3
    close_u16(out, g['u16_out'], 8)
    assert np.count_nonzero(out != orc.correct_transversalium2(img, circle, borders, 301)[0]) > 1000     # a different filter
    out, flag = orc.correct_transversalium2_stubborn(img * g['row_factor'][:, None], circle, borders, 301)
    close_u16(out, g['f64_out'], 8)
    out, flag = orc.correct_transversalium2_stubborn(img, (0, 0, 99999), list(g['bb_borders']), 41)
    assert flag.sum() >= 3
    close_u16(out, g['bb_out'], 8)


def test_row_box_sums_known_answers():
    a = np.arange(12, dtype=np.float64).reshape(2, 6)
    s = orc.row_box_sums_reflect101(a, 3)
    np.testing.assert_array_equal(s[0], [1 + 0 + 1, 0 + 1 + 2, 1 + 2 + 3, 2 + 3 + 4, 3 + 4 + 5, 4 + 5 + 4])   # gfedcb|abcdefgh|gfedcba
    np.testing.assert_array_equal(orc.row_box_sums_reflect101(np.ones((3, 9)), 7), np.full((3, 9), 7.0))


@pytest.mark.parametrize('kw,kh', [(5, 5), (25, 17), (25, 16), (3, 1), (7, 7), (9, 13)])
def test_box_blur_restatement_against_scipys_uniform_filter(kw, kh):
    """cv2.blur is absent here (parity unpinned); its restatement -- exact integer window sums with BORDER_REFLECT_101, OpenCV's
    float32 / double scaling, round half to even -- is held against an independent implementation of the same box filter:
    scipy.ndimage.uniform_filter (mode='mirror' is REFLECT_101; the window sizes the path uses: 5 x 5, 25 x h/100, k x k).
    OpenCV scales a window sum in float32 where it vectorises (24 bits for sums of up to 2.8e7), SciPy averages in float64: equal
    on all but a thousandth of the pixels (measured: 0 for the small windows, 22 of 33 127 for 25 x 17), never more than one grey
    level apart, and equal everywhere on a constant image."""
    from scipy import ndimage
    rng = np.random.default_rng(kw * 100 + kh)
    yy, xx = np.mgrid[0:211, 0:157]
    img = np.clip(20000 + 15000 * np.sin(yy / 17.0) * np.cos(xx / 23.0) + rng.normal(0, 900, yy.shape), 0, 65535).astype(np.uint16)
    got = orc.box_blur_u16(img, kw, kh).astype(np.int64)
    ref = np.rint(ndimage.uniform_filter(img.astype(np.float64), size=(kh, kw), mode='mirror')).astype(np.int64)
    d = np.abs(got - ref)
    assert d.max() <= 1 and np.count_nonzero(d) <= d.size // 1000, (d.max(), np.count_nonzero(d))
    flat = np.full((40, 50), 12345, dtype=np.uint16)
    np.testing.assert_array_equal(orc.box_blur_u16(flat, kw, kh), flat)
