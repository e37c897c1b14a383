"""Host-side logic of the product (no GPU): limb fit vs the oracle, percentile from
histograms, FITS / PNG encoders, CLI flag table, SER header parsing."""
import json
import os

import numpy as np
import pytest

from oracle import limb_oracle as limb
from oracle import shg_oracle as orc
from solex_ser_recon_en_amd import CLI_handler, fits_io, hostmath, order_stats, png_io, synth
from solex_ser_recon_en_amd.ellipse_to_circle import get_correction_matrix
from solex_ser_recon_en_amd.solex_util import max_from_hist, percentile_from_hist
from tests import numpy_ref


def points_via_scipy_label(edges):
    """The product's limb-point selection (shg_host_limb_points) fed the way the GPU labelling feeds it: raster-ordered
    edge pixels + component roots.  -> (X float [n, 2], raw int [m, 2])"""
    from scipy import ndimage as ndi
    labelled, nf = ndi.label(edges, np.ones((3, 3), int))
    pts = np.argwhere(edges)
    lab_img = labelled[pts[:, 0], pts[:, 1]]
    lin = pts[:, 0] * edges.shape[1] + pts[:, 1]
    first = np.full(nf + 1, np.iinfo(np.int64).max)
    np.minimum.at(first, lab_img, lin)                       # root = smallest linear index of the component
    lab, n = numpy_ref.labels_from_roots(first[lab_img])
    assert n == nf and np.array_equal(lab, lab_img)          # root order == scipy's raster numbering
    sel = hostmath.limb_points(lin, first[lab_img], edges.shape[0], edges.shape[1])
    X = np.array(pts[sel.astype(bool)], dtype='float')
    np.testing.assert_array_equal(X, numpy_ref.limb_points(pts, lab, nf, edges.shape[0]))
    return X, pts


def flood_threshold_numpy(small, blurred):
    """The product's flood threshold (shg_host_flood_threshold) fed with the statistics the GPU would reduce."""
    n = small.size
    lo, hi, p99 = order_stats.lerp_order_stats(n, 99)
    srt = np.sort(blurred.ravel())
    very_bright = p99(srt[lo], srt[hi])
    assert very_bright == np.percentile(blurred, 99)
    data = blurred.ravel()[blurred.ravel() < very_bright]
    counts, _ = np.histogram(data, bins=20)
    got = hostmath.flood_threshold(np.sum(small), small.shape, data.min(), data.max(), counts)
    assert got == numpy_ref.flood_threshold(np.sum(small), small.shape, data.min(), data.max(), counts)
    return got


def test_order_stat_helpers_are_numpy():
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 10, 11, 1000, 12345):
        v = rng.random(n)
        srt = np.sort(v)
        lo, hi, med = order_stats.median_order_stats(n)
        assert med(srt[lo], srt[hi]) == np.median(v)
        for q in (0, 1, 50, 99, 99.9999, 100):
            lo, hi, f = order_stats.lerp_order_stats(n, q)
            assert f(srt[lo], srt[hi]) == np.percentile(v, q)


def test_limb_control_plane_matches_oracle(golden):
    """The host half of the limb fit (threshold choice, hysteresis, region selection, ellipse LSQ)
    against the oracle, fed with the arrays the GPU half would produce."""
    g = golden('g13_limb')
    small = g['small']
    k = int(small.shape[0] * 0.01)
    blurred = orc.box_blur_f64(small, k, k)
    thresh3 = flood_threshold_numpy(small, blurred)
    np.testing.assert_array_equal(np.where(blurred < thresh3, 0.0, 65000.0), g['flooded'])
    edges = g['canny0']
    X, raw = points_via_scipy_label(edges)
    np.testing.assert_array_equal(raw, np.argwhere(edges))
    Xo, rawo = limb.get_edge_list(small.copy())
    np.testing.assert_array_equal(X, Xo)
    np.testing.assert_array_equal(np.argwhere(edges), rawo)
    got = hostmath.two_step(X * 4)
    want = limb.two_step(Xo * 4)
    for a, b in zip(got[:4], want[:4]):
        np.testing.assert_allclose(a, b, rtol=3e-16)
    np.testing.assert_array_equal((X * 4)[got[4].astype(bool)], want[4])


def test_limb_region_selection_cases():
    """Two largest regions by size value (ties -> first), hull filter, 1.7 % row crop: mask form vs point form."""

    def ring(shape, cy, cx, r, arc=(0, 2 * np.pi)):
        m = np.zeros(shape, bool)
        t = np.linspace(arc[0], arc[1], 2000)
        m[np.clip(np.rint(cy + r * np.sin(t)).astype(int), 0, shape[0] - 1), np.clip(np.rint(cx + r * np.cos(t)).astype(int), 0, shape[1] - 1)] = True
        return m
    cases = []
    a = ring((120, 140), 60, 70, 50)                                  # one closed limb
    cases.append(a)
    b = ring((120, 140), 60, 70, 50, (0.2, 3.0)) | ring((120, 140), 60, 70, 50, (3.4, 6.1))   # limb in two arcs
    b[60, 70] = True; b[61, 71] = True                                # a small blob inside: third region, dropped
    cases.append(b)
    c = ring((120, 140), 60, 70, 50, (0.1, 6.2)) | ring((120, 140), 60, 70, 20)               # inner ring: 2nd largest, off the hull
    cases.append(c)
    d = np.zeros((90, 90), bool)                                      # equal sizes: list.index picks the first twice
    d[10, 10:30] = True; d[11:15, 10] = True; d[40, 10:34] = True; d[70, 10:34] = True; d[80, 50:60] = True
    cases.append(d)
    for edges in cases:
        X, raw = points_via_scipy_label(edges)
        Xo, rawo = limb.points_from_edges(edges)
        np.testing.assert_array_equal(X, Xo)
        np.testing.assert_array_equal(raw, rawo)


def test_gaussian_taps_are_scipys():
    from scipy import ndimage as ndi
    from solex_ser_recon_en_amd.ops import gaussian_taps
    for sigma in (2, 1.5, 1.0, 0.5):
        taps, radius = gaussian_taps(sigma)
        impulse = np.zeros(4 * radius + 1); impulse[2 * radius] = 1.0
        resp = ndi.gaussian_filter1d(impulse, sigma, mode='constant')
        np.testing.assert_array_equal(resp[radius:3 * radius + 1], taps)


def test_correction_matrix_golden(golden):
    g = golden('g7_matrix')
    for (phi, r), mat, theta in zip(g['params'], g['mats'], g['thetas']):
        m, t = get_correction_matrix(phi, r)
        np.testing.assert_allclose(m, mat, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(t, theta, rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize('q', [0, 10, 50, 99, 99.9999, 100, 33.3])
def test_percentile_from_hist_is_np_percentile(q):
    rng = np.random.default_rng(int(q * 10))
    for n in (1, 2, 7, 1000, 54321):
        img = rng.integers(0, 65536, n).astype(np.uint16)
        if n > 100:
            img[: n // 3] = 40000          # heavy ties
        hist = np.bincount(img, minlength=65536)
        assert percentile_from_hist(hist, q) == np.percentile(img, q)
        assert max_from_hist(hist) == img.max()


def test_percentile_golden(golden):
    g = golden('g5_rescale')
    hist = np.bincount(g['image'].ravel(), minlength=65536)
    assert percentile_from_hist(hist, 99.9999) == float(g['bright'])
    assert percentile_from_hist(hist, 10) == float(g['p10'])


def test_column_plan_matches_oracle(golden):
    g = golden('g2_extract')
    fit = g['u16_rot_fit']
    shifts = [10, 0, -25, 7]
    ind_l, lw, rw = hostmath.column_plan(fit, shifts, fit.shape[0], 40)
    cols, olw, orw = orc.column_indices(fit, shifts, 40)
    np.testing.assert_array_equal(ind_l, np.stack([c[0] for c in cols]))
    np.testing.assert_array_equal(lw, olw)
    np.testing.assert_array_equal(rw, orw)


def test_fits_writer_matches_astropy_bytes(golden):
    g = golden('g9_fits')

    class R:
        iw, ih = 7, 5
    hdr = fits_io.make_header(R)
    hdr['NAXIS1'] = 9
    assert fits_io.fits_bytes(g['array'], hdr) == g['fits_bytes'].tobytes()


def test_fits_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 65536, (33, 47)).astype(np.uint16)
    path = str(tmp_path / 'x.fits')
    fits_io.write_fits(path, img, {'BIN1': 1})
    back, cards = fits_io.read_fits_u16(path)
    np.testing.assert_array_equal(back, img)
    assert cards['BITPIX'] == '16' and cards['BZERO'] == '32768' and os.path.getsize(path) % 2880 == 0


@pytest.mark.parametrize('dtype', [np.uint8, np.uint16])
def test_png_round_trip(tmp_path, dtype):
    rng = np.random.default_rng(1)
    img = rng.integers(0, np.iinfo(dtype).max + 1, (19, 23)).astype(dtype)
    path = str(tmp_path / 'x.png')
    png_io.write_png(path, img, 0)
    np.testing.assert_array_equal(png_io.read_png_gray(path), img)
    try:
        from PIL import Image
    except ImportError:
        return
    np.testing.assert_array_equal(np.array(Image.open(path)), img)          # an independent decoder agrees
    Image.fromarray(img).save(path)                                          # filtered rows from an independent encoder
    np.testing.assert_array_equal(png_io.read_png_gray(path), img)


def test_cli_flag_table_matches_reference(golden, capsys):
    table = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'g10_cli.json')))
    base = {'shift': [0], 'flag_display': False, 'ratio_fixe': None, 'slant_fix': None, 'save_fit': False,
            'clahe_only': False, 'protus_only': False, 'disk_display': True, 'delta_radius': 0,
            'crop_width_square': False, 'transversalium': True, 'flip_x': False, 'fixed_width': None}
    for arg, want in table.items():
        opts = dict(base)
        CLI_handler.treat_flag_at_cli(opts, arg)
        assert opts == want, arg
    capsys.readouterr()


def test_cli_detached_values_and_files(capsys):
    opts = {'shift': [0], 'fixed_width': None}
    files = CLI_handler.handle_CLI(opts, ['-w', '-10:10:1', '-r', '900', 'a.ser', 'b.SER', 'notes.txt', '-cf'])
    assert files == ['a.ser', 'b.SER']
    assert opts['shift'] == list(range(-10, 11)) and opts['fixed_width'] == 900
    assert opts['clahe_only'] is True and opts['save_fit'] is True
    with pytest.raises(ValueError):
        CLI_handler.handle_CLI({'shift': [0]}, ['-w'])
    capsys.readouterr()


def test_ser_header_parse(tmp_path):
    from solex_ser_recon_en_amd.video_reader import video_reader
    frames = synth.synth_frames_numpy(5, 40, 12, 8, seed=0)
    path = str(tmp_path / 'a.ser')
    synth.write_ser(path, frames)
    rdr = video_reader(path)
    assert (rdr.Width, rdr.Height, rdr.FrameCount, rdr.infilebytes) == (40, 12, 5, 1)
    assert rdr.flag_rotate and (rdr.ih, rdr.iw) == (40, 12) and rdr.count == 480
    ref = orc.SerReader(path)
    k = 0
    while rdr.has_frames():
        np.testing.assert_array_equal(rdr.next_frame(), ref.next_frame())
        k += 1
    assert k == 5
    with pytest.raises(Exception, match='neither is SER nor AVI'):
        video_reader(str(tmp_path / 'a.txt'))
    with pytest.raises(FileNotFoundError):
        video_reader(str(tmp_path / 'a.avi'))
    open(str(tmp_path / 'short.ser'), 'wb').write(b'LUCAM')
    with pytest.raises(Exception, match='truncated'):
        video_reader(str(tmp_path / 'short.ser'))


def test_stubborn_control_plane_matches_oracle():
    """Host side of the stubborn transversalium branch (solex_util.py:356-375, 416-421): vectorised row maps
    against the oracle's loops, including wrap-around dilation, leading / trailing flagged runs and circles that
    touch or leave the image."""
    from solex_ser_recon_en_amd import solex_util as su
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 64):
        for _ in range(20):
            flag = rng.random(n) < rng.choice([0.0, 0.2, 0.7, 1.0])
            up, dn = su._nearest_unflagged(flag)
            rup, rdn = orc.neighbour_rows(flag)
            sel = flag                                         # only read where flagged
            np.testing.assert_array_equal(up[sel], rup[sel])
            np.testing.assert_array_equal(dn[sel], rdn[sel])
    for trial in range(30):
        n = int(rng.integers(5, 60))
        y1 = int(rng.integers(0, 4))
        y2 = n - int(rng.integers(0, 4))
        corr = np.exp(0.01 * rng.standard_normal(y2 - y1))
        corr[rng.integers(0, y2 - y1)] *= 1.5
        if trial % 3 == 0:
            corr[0] *= 2.0                                     # flags row y1: the dilation wraps when y1 == 0
        np.testing.assert_array_equal(su._spurious_rows(corr, n, y1, y2), orc.spurious_rows(corr, n, y1, y2))
    for circle, h, w in [((160.5, 149.3, 132.0), 300, 330), ((40.0, 30.0, 80.0), 200, 260), ((130.2, 100.0, 100.0), 201, 260),
                         ((0, 0, 99999), 120, 300), ((50.0, 60.0, 20.0), 120, 130), ((100.0, 100.0, 60.5), 150, 200),
                         ((65.0, 50.0, 50.0), 101, 131)]:
        xa, xb, edge, half = su._limb_edge_plan(circle, h, w, 121)
        rxa, rxb, left, right, rhalf = orc.edge_plan(circle, h, w, 121)
        assert half == rhalf == 60
        np.testing.assert_array_equal(xa, rxa)
        np.testing.assert_array_equal(xb, rxb)
        np.testing.assert_array_equal(edge & 1, left.astype(np.uint8))
        np.testing.assert_array_equal(edge >> 1, right.astype(np.uint8))


AVI_CASES = [('Y800', {}), ('pal8', {}), ('pal8', {'bottom_up': False}), ('pal8', {'audio_every': 3, 'junk_bytes': 37}),
             ('Y800', {'rec_lists': True}), ('pal8', {'palette': 'random'}), ('bgr24', {}), ('bgr24', {'bottom_up': False})]


def avi_case(path, layout, kw, n=7, h=9, w=14, seed=0):
    rng = np.random.default_rng(seed)
    kw = dict(kw)
    if kw.get('palette') == 'random':
        kw['palette'] = rng.integers(0, 256, (256, 3)).astype(np.uint8)
    shape = (n, h, w, 3) if layout == 'bgr24' else (n, h, w)
    frames = rng.integers(0, 256, shape).astype(np.uint8)
    synth.write_avi(path, frames, layout, **kw)
    return frames


@pytest.mark.parametrize('layout,kw', AVI_CASES)
@pytest.mark.parametrize('h,w', [(9, 14), (14, 9), (5, 5)])
def test_avi_index_and_host_frames_match_oracle(tmp_path, layout, kw, h, w):
    """Uncompressed AVI: the product walks the 'movi' chunks, the oracle follows 'idx1'; both must land on the same
    pixels, in the reference's orientation (video_reader.py:111-122), for every row order / padding / palette."""
    from solex_ser_recon_en_amd.video_reader import video_reader
    path = str(tmp_path / 'scan.avi')
    frames = avi_case(path, layout, kw, h=h, w=w)
    rdr, ref = video_reader(path), orc.AviReader(path)
    assert (rdr.Width, rdr.Height, rdr.FrameCount, rdr.ih, rdr.iw, rdr.flag_rotate) == \
           (ref.Width, ref.Height, ref.FrameCount, ref.ih, ref.iw, ref.flag_rotate) == (w, h, 7, max(h, w), min(h, w), w > h)
    assert rdr.infilebytes == 1 and rdr.infiledatatype == 'uint8' and rdr.AVI_flag and not rdr.SER_flag
    if layout != 'bgr24' and 'palette' not in kw:
        np.testing.assert_array_equal(ref.raw_frames(), frames)          # grey in, grey out: BGR2GRAY is the identity
    k = 0
    while rdr.has_frames():
        assert ref.has_frames()
        got, want = rdr.next_frame(), ref.next_frame()
        assert got.dtype == np.uint16 and got.shape == (max(h, w), min(h, w))
        np.testing.assert_array_equal(got, want)
        k += 1
    assert k == 7 and not ref.has_frames()


def test_avi_compressed_or_broken_files_raise(tmp_path):
    from solex_ser_recon_en_amd.video_reader import video_reader
    path = str(tmp_path / 'scan.avi')
    avi_case(path, 'Y800', {})
    data = bytearray(open(path, 'rb').read())
    at = data.find(b'strf') + 8 + 16
    assert data[at:at + 4] == b'Y800'
    data[at:at + 4] = b'MJPG'
    open(path, 'wb').write(data)
    with pytest.raises(Exception, match='codec'):
        video_reader(path)
    open(path, 'wb').write(b'RIFF\x04\x00\x00\x00WAVE')
    with pytest.raises(Exception, match='not a RIFF AVI'):
        video_reader(path)
    avi_case(path, 'pal8', {})
    data = open(path, 'rb').read()
    open(path, 'wb').write(data[:len(data) // 2])                        # truncated inside 'movi'
    with pytest.raises(Exception):
        rdr = video_reader(path)
        while rdr.has_frames():
            rdr.next_frame()


def test_bgr2gray_fixed_point():
    from solex_ser_recon_en_amd.avi_io import bgr_to_gray_u8
    v = np.arange(256, dtype=np.uint8)
    np.testing.assert_array_equal(bgr_to_gray_u8(v, v, v), v)            # identity on grey
    b, g, r = np.array([255, 0, 0], np.uint8), np.array([0, 255, 0], np.uint8), np.array([0, 0, 255], np.uint8)
    np.testing.assert_array_equal(bgr_to_gray_u8(b, g, r), [29, 150, 76])   # OpenCV's well-known primaries


def test_savgol_trend_is_scipys():
    """The trend inside shg_host_transversalium_factors (interior correlation + LAPACK edge fits) == scipy's
    savgol_filter bit for bit: with a zero mean removed and no taper, -log(factor) returns its cumulative sum."""
    from scipy.signal import savgol_filter
    from solex_ser_recon_en_amd import solex_util as su
    rng = np.random.default_rng(0)
    for n, win in [(280, 279), (1800, 301), (40, 21), (5, 5), (302, 301)]:
        y = rng.standard_normal((3, n)) * 0.01
        got = hostmath.transversalium_factors(y, None, su.savgol_taps(win), tapered=False)
        for r, g in zip(y, got):
            d = r - savgol_filter(r, win, 3)
            np.testing.assert_allclose(g, np.exp(-np.cumsum(d - np.mean(d))), rtol=4e-16, atol=0)


def test_folder_argument_expands_to_its_scans(tmp_path):
    from solex_ser_recon_en_amd import SHG_MAIN
    for name in ('b.ser', 'a.SER', 'c.avi', 'notes.txt', 'd.png'):
        (tmp_path / name).write_bytes(b'x')
    (tmp_path / 'sub.ser').mkdir()
    assert [os.path.basename(f) for f in SHG_MAIN.scans_in(str(tmp_path))] == ['a.SER', 'b.ser', 'c.avi']


def test_upload_reader_direct_and_buffered_reads_agree(tmp_path, monkeypatch):
    """video_reader's chunk reader (file -> the upload service's pinned buffer): the O_DIRECT path reads the enclosing 4 KiB-aligned
    span (the SER header is 178 bytes: no frame starts on a block) and reports where the wanted bytes begin; a chunk that ends at
    the end of the file, a file system that refuses O_DIRECT and SHG_READ_DIRECT=0 all give the very bytes the buffered read gives."""
    import mmap
    from solex_ser_recon_en_amd import video_reader as vr
    data = os.urandom(3 * (1 << 20) + 12345)
    path = str(tmp_path / 'blob.ser')
    with open(path, 'wb') as f:
        f.write(data)
        f.flush()
        os.fsync(f.fileno())
    buf = mmap.mmap(-1, (1 << 20) + 2 * vr._DIRECT_ALIGN)     # page-aligned, like a pinned buffer
    mv = memoryview(buf)
    try:
        for mode in ('0', '1', 'auto'):
            monkeypatch.setenv('SHG_READ_DIRECT', mode)
            fd = os.open(path, os.O_RDONLY)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)  # cold where the file system can forget it
            os.close(fd)
            h = vr._FileHandles()
            for offset, n in ((178, 1 << 20), (178 + (1 << 20), 1 << 20), (len(data) - 70001, 70001), (4096, 8192), (0, 1)):
                h.open(path, offset, n)
                if mode == '0':
                    assert h.dfd < 0
                skip = vr._read_chunk(h, mv, offset, n)
                assert 0 <= skip < vr._DIRECT_ALIGN and bytes(mv[skip:skip + n]) == data[offset:offset + n], (mode, offset, n)
            h.close()
            assert h.fd < 0 and h.dfd < 0
        share = vr._page_cache_share
        fd = os.open(path, os.O_RDONLY)
        os.pread(fd, 1 << 20, 0)
        assert 0.0 <= share(fd, 0, 1 << 20) <= 1.0 and share(-1, 0, 4096) == 1.0      # (a descriptor that cannot be mapped: "cached")
        os.close(fd)
    finally:
        mv.release()
        buf.close()


def test_bench_starts_its_ranks_and_reports_a_rank_that_fails():
    """bench.py --gpus 2 with no launcher around it starts two rank processes itself (before it imports torch): here, without a
    GPU, both ranks stop with 'needs a GPU' and bench.py exits non-zero with them, printing no JSON line."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['HIP_VISIBLE_DEVICES'] = ''
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, cwd=repo, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count('bench.py needs a GPU') == 2, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
