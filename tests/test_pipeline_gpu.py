"""End-to-end drop-in parity on the GPU: solex_read / solex_process / the CLI against the
CPU oracle and against the reference's own shim-mode run (g14)."""
import os

import numpy as np
import pytest

from tests.conftest import flips

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

from oracle import pipeline_oracle as po          # noqa: E402
from solex_ser_recon_en_amd import synth          # noqa: E402

SCENARIOS = {'A': {}, 'B': {'shift': [-2, 0, 3], 'flip_x': True, 'crop_width_square': True},
             'C': {'ratio_fixe': 1, 'fixed_width': 300, 'disk_display': False, 'img_rotate': 90},
             'D': {'de-vignette': True, 'shift': [0, 4]},
             'E': {'de-vignette': True, 'transversalium': False, 'crop_width_square': True},
             'F': {'stubborn_transversalium': True, 'trans_strength': 41},
             'G': {'stubborn_transversalium': True, 'de-vignette': True}}
PRODUCT_KEY = {'clahe': 'cc', 'protus': 'protus', 'uncontrasted': 'raw', 'high_contrast': 'hc'}


@pytest.fixture(scope='module')
def pkg():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, outputs
    return SHG_MAIN, Solex_recon, outputs


@pytest.fixture(scope='module')
def scan(golden, tmp_path_factory):
    g = golden('g14_pipeline')
    frames = synth.synth_frames_numpy(int(g['param_n']), int(g['param_w']), int(g['param_h']), int(g['param_bits']),
                                      seed=int(g['param_seed']), tilt=float(g['param_tilt']), curv=float(g['param_curv']),
                                      row_gain=g['row_gain'])
    path = str(tmp_path_factory.mktemp('scan') / 'scan.ser')
    synth.write_ser(path, frames)
    return g, frames, path


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def close_to_capture(label, got, want, max_flips, lsb):
    """Against the REFERENCE'S OWN capture of a stubborn-transversalium product (another host's exp / log / float32 filter sums):
    the one stated allowance of this file.  Everything else is held to the flip counts observed, which are 0 (conftest.flips)."""
    assert got.shape == want.shape
    diff = np.abs(np.asarray(got).astype(np.int64) - np.asarray(want).astype(np.int64))
    print('PARITY %-60s %9d px  %d differ  max %d LSB (capture of another host: <= %d px, <= %d LSB allowed)' % (
        label, diff.size, np.count_nonzero(diff), diff.max(), max_flips, lsb))
    assert diff.max() <= lsb and np.count_nonzero(diff) <= max_flips, (diff.max(), np.count_nonzero(diff))


@pytest.mark.parametrize('tag', ['A', 'B', 'C', 'D', 'E', 'F', 'G'])
def test_solex_read_process_vs_oracle_and_reference(pkg, scan, tag):
    SHG_MAIN, Solex_recon, outputs = pkg
    g, frames, path = scan
    opts = SHG_MAIN.default_options()
    opts.update(SCENARIOS[tag], _nolog=True)
    disk_list, bounds, hdr = Solex_recon.solex_read(path, opts)
    want = po.run(frames, SCENARIOS[tag])
    # integer stages: bit exact
    assert opts['shift'] == want['read']['shifts'] == [int(s) for s in g[tag + '_shifts']]
    assert tuple(int(b) for b in bounds) == (want['read']['y1'], want['read']['y2'])
    for got, ref in zip(disk_list, want['read']['disks']):
        np.testing.assert_array_equal(np.asarray(got), ref)
    results = Solex_recon.solex_process(opts, disk_list, bounds, hdr)
    outputs.flush()
    requested = [s for s in opts['shift'] if s in opts['shift_requested']]
    assert len(results) == len(requested)
    if tag != 'C':
        np.testing.assert_allclose(opts['ratio_fixe'], want['geometry']['ratio'], rtol=1e-9)
        np.testing.assert_allclose(opts['slant_fix'], np.degrees(want['geometry']['phi']), rtol=1e-6, atol=1e-9)
    for shift, (cc, protus) in zip(requested, results):
        ref = want['results'][shift]
        # float stages (warp f64, transversalium f64 + device log, CLAHE f32) against the oracle on this host: all seven
        # scenarios come out bit-identical, stubborn ones included -- held to that
        flips('scenario %s shift %d cc' % (tag, shift), cc, ref['cc'])
        flips('scenario %s shift %d protus' % (tag, shift), protus, ref['protus'])
        for product in ('clahe', 'protus'):
            key = '%s_s%d_%s' % (tag, shift, product)
            # The reference's own output for this product, captured under NumPy 1.26 / SciPy 1.7 (oracle/capture_goldens.py).
            # Stubborn scenarios: that host's exp / log / float32 filter sums leave ONE pixel of the filtered frame a grey
            # level away from this host's, and the CLAHE slope there (about 3) stretches it -- measured 3 LSB on 1 of
            # 168 000 px in scenario F, 0 in G; every other scenario is exact.
            if key in g.files:
                stretched = tag in 'FG' and product == 'clahe'
                if stretched:
                    close_to_capture(key, {'clahe': cc, 'protus': protus}[product], g[key], max_flips=2, lsb=4)
                else:
                    flips(key + ' vs the reference capture', {'clahe': cc, 'protus': protus}[product], g[key])


def test_cli_writes_the_reference_file_layout(pkg, scan, tmp_path):
    """CLI -> same file names / dtypes / shapes as SURVEY.md section 8b's output table."""
    SHG_MAIN, Solex_recon, outputs = pkg
    g, frames, path = scan
    work = str(tmp_path / 'scan.ser')
    os.link(path, work) if hasattr(os, 'link') else None
    if not os.path.exists(work):
        synth.write_ser(work, frames)
    from solex_ser_recon_en_amd import fits_io, png_io
    assert SHG_MAIN.main(['-f', work]) == 0
    base = work[:-4]
    expect = ['_log.txt', '_mean.fits', '_spectral_line_data.png', '_shift=0_raw.fits', '_shift=10_ellipse_fit.png',
              '_shift=0_circular.fits', '_shift=0_transversalium_correction.png', '_shift=0_detransversaliumed.fits',
              '_shift=0_clahe.png', '_shift=0_protus.png', '_shift=0_uncontrasted.png', '_shift=0_high_contrast.png',
              '_shift=0_clahe.fits']
    for suffix in expect:
        assert os.path.exists(base + suffix), 'missing output ' + suffix
    assert not os.path.exists(base + '_shift=10_raw.fits')            # not requested (Solex_recon.py:78-82)
    want = po.run(frames, {})
    raw, cards = fits_io.read_fits_u16(base + '_shift=0_raw.fits')
    np.testing.assert_array_equal(raw, want['read']['disks'][1])
    mean, _ = fits_io.read_fits_u16(base + '_mean.fits')
    np.testing.assert_array_equal(mean, want['read']['mean'])
    flips('CLI clahe.png vs the reference capture', png_io.read_png_gray(base + '_shift=0_clahe.png'), g['A_s0_clahe'])
    flips('CLI protus.png vs the reference capture', png_io.read_png_gray(base + '_shift=0_protus.png'), g['A_s0_protus'])
    flips('CLI uncontrasted.png vs the reference capture', png_io.read_png_gray(base + '_shift=0_uncontrasted.png'), g['A_s0_uncontrasted'])
    flips('CLI high_contrast.png vs the reference capture', png_io.read_png_gray(base + '_shift=0_high_contrast.png'), g['A_s0_high_contrast'])
    flips('CLI clahe.fits vs the oracle', fits_io.read_fits_u16(base + '_shift=0_clahe.fits')[0], want['results'][0]['cl1'])
    log = open(base + '_log.txt').read()
    for needle in ('start time', 'Pixel shift : [0]', 'Width, Height : 400 32', 'Number of frames : 400',
                   'Vertical limits y1, y2 : 41 359', 'Spectral line polynomial fit', 'Transversalium correction : 301',
                   'Y/X ratio', 'Tilt angle', 'Disk position, radius', 'end time'):
        assert needle in log, needle


def test_failure_propagates_like_the_reference(pkg, tmp_path):
    """A scan whose sunlit span is too short makes cv2.blur raise in the reference (solex_util.py:229-230);
    here the C ABI returns an error code and solex_do_work raises; the front door catches (SHG_MAIN.py:136-143)."""
    SHG_MAIN, Solex_recon, outputs = pkg
    frames = synth.synth_frames_numpy(12, 90, 32, 16, seed=0)
    path = str(tmp_path / 'tiny.ser')
    synth.write_ser(path, frames)
    opts = SHG_MAIN.default_options()
    opts['_nolog'] = True
    with pytest.raises(RuntimeError, match='shg_box_blur_u16'):
        Solex_recon.solex_do_work([(path, opts)], True)
    assert SHG_MAIN.handle_files([path], SHG_MAIN.default_options(), True) is False


def test_apply_clahe_tool(pkg, tmp_path):
    from oracle import shg_oracle as orc
    from solex_ser_recon_en_amd import clahe_apply, png_io
    rng = np.random.default_rng(2)
    yy, xx = np.mgrid[0:120, 0:150]
    img = np.clip((0.5 + 0.4 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 0.03 * rng.standard_normal((120, 150))) * 65535, 0, 65535).astype(np.uint16)
    path = str(tmp_path / 'sun.png')
    png_io.write_png(path, img)
    for tile in (1, 2, 3, 4):
        out = clahe_apply.apply_clahe(path, dict(clahe_apply.options, tile_size=tile), write_file=(tile == 2))
        np.testing.assert_array_equal(out, orc.clahe(img, 0.8, tile))
    np.testing.assert_array_equal(png_io.read_png_gray(str(tmp_path / 'sun_clahe.png')), orc.clahe(img, 0.8, 2))
    o = dict(clahe_apply.options, do_stretch=True, lo=1, hi=99.5, sat=80)
    want = orc.rescale_brightness(orc.clahe(img, 0.8, 2), np.percentile(img, 1), np.percentile(img, 99.5), alpha=0.8)
    np.testing.assert_array_equal(clahe_apply.apply_clahe(path, o, write_file=False), want)
    img8 = (img >> 8).astype(np.uint8)
    np.testing.assert_array_equal(clahe_apply.apply_clahe(img8, dict(clahe_apply.options), write_file=False), orc.clahe(img8, 0.8, 2))
    # the 8-bit stretch (sat = 255) also runs on the GPU
    want8 = orc.rescale_brightness(orc.clahe(img8, 0.8, 2), np.percentile(img8, 1), np.percentile(img8, 99.5), alpha=0.8)
    got8 = clahe_apply.apply_clahe(img8, o, write_file=False)
    assert got8.dtype == np.uint8
    np.testing.assert_array_equal(got8, want8)


def test_folder_of_files_with_prefetch(pkg, scan, tmp_path):
    """Several files through solex_do_work: file k+1 is decoded into HBM while file k is processed;
    every file gets the same products as when processed alone."""
    SHG_MAIN, Solex_recon, outputs = pkg
    g, frames, path = scan
    from solex_ser_recon_en_amd import png_io
    files = []
    for i in range(3):
        f = str(tmp_path / ('scan%d.ser' % i))
        synth.write_ser(f, frames if i != 1 else frames[::-1].copy())          # file 1: the scan reversed
        files.append(f)
    opts = SHG_MAIN.default_options()
    opts['clahe_only'] = True
    Solex_recon.solex_do_work(SHG_MAIN.precheck_files(files, opts), True)
    a = png_io.read_png_gray(files[0][:-4] + '_shift=0_clahe.png')
    b = png_io.read_png_gray(files[1][:-4] + '_shift=0_clahe.png')
    c = png_io.read_png_gray(files[2][:-4] + '_shift=0_clahe.png')
    flips('clahe of scenario A vs the reference capture', a, g['A_s0_clahe'])
    np.testing.assert_array_equal(a, c)
    assert a.shape[0] == b.shape[0] and not np.array_equal(a, b)
    # a missing file in the middle stops the batch when its turn comes (README: "will halt if a file is unsuitable")
    with pytest.raises(Exception):
        Solex_recon.solex_do_work([(files[0], SHG_MAIN.default_options()), (str(tmp_path / 'nope.ser'), SHG_MAIN.default_options())], True)


def test_scan_workers_give_the_serial_products_bit_for_bit(pkg, tmp_path):
    """A batch of different scans (files and device-resident stacks, different shapes, depths and options) through one,
    two and four scan workers: every raw disk and every product is identical to the one-at-a-time order, whatever the
    interleaving on the GPU -- files are independent, each worker has its own stream, workspace and staging buffers."""
    SHG_MAIN, Solex_recon, outputs = pkg
    from solex_ser_recon_en_amd.video_reader import array_reader
    specs = [(300, 400, 32, 16, 3, {}), (260, 520, 40, 16, 4, {'shift': [-2, 0, 3]}), (300, 400, 32, 8, 5, {'flip_x': True}),
             (280, 32, 400, 16, 6, {'crop_width_square': True}), (300, 400, 32, 16, 7, {'transversalium': False}),
             (320, 480, 36, 16, 8, {'de-vignette': True}), (300, 400, 32, 16, 9, {'fixed_width': 300, 'img_rotate': 90}),
             (300, 400, 32, 16, 10, {}), (300, 400, 32, 16, 11, {'stubborn_transversalium': True, 'trans_strength': 41})]
    sources = []
    for i, (n, w, h, bits, seed, extra) in enumerate(specs):
        frames = synth.synth_frames_numpy(n, w, h, bits, seed=seed, tilt=0.01, curv=5e-5)
        if i % 2:
            path = str(tmp_path / ('scan%d.ser' % i))
            synth.write_ser(path, frames)
            sources.append((path, extra))
        else:
            sources.append((torch.from_numpy(frames).cuda(), extra))

    def run(workers):
        tasks = []
        for src, extra in sources:
            opts = SHG_MAIN.default_options()
            opts.update(extra, _nolog=True)
            tasks.append((src if isinstance(src, str) else array_reader(src), opts))
        res = Solex_recon.solex_do_work(tasks, True, return_results=True, workers=workers)
        outputs.flush()
        return [[(np.asarray(cc), np.asarray(pr)) for cc, pr in per_file] for per_file in res], [t[1] for t in tasks]
    serial, opts1 = run(1)
    for workers in (2, 4):
        piped, optsw = run(workers)
        assert len(piped) == len(serial) == len(specs)
        for a, b, oa, ob in zip(serial, piped, opts1, optsw):
            assert len(a) == len(b) and oa['ratio_fixe'] == ob['ratio_fixe'] and oa['slant_fix'] == ob['slant_fix']
            for (cc1, p1), (cc2, p2) in zip(a, b):
                np.testing.assert_array_equal(cc1, cc2)
                np.testing.assert_array_equal(p1, p2)
    # a failure in the middle of a pipelined batch: the batch stops and the error of the lowest failing task is raised
    # (a reader per task: a reader gives its stack up once its scan has read it, it cannot serve two scans at once)
    bad = [task for _ in range(3) for task in ((array_reader(sources[0][0]), dict(SHG_MAIN.default_options(), _nolog=True)),
                                               (str(tmp_path / 'nope.ser'), dict(SHG_MAIN.default_options(), _nolog=True)))]
    with pytest.raises(Exception, match='nope|No such file'):
        Solex_recon.solex_do_work(bad, True, workers=4)
    torch.cuda.synchronize()


def test_stage_functions_keep_the_reference_call_surface(pkg, scan):
    """The second caller of the stage functions (spectralAnalyserUI.py:155-175, 345-359): all_video_reader ->
    compute_mean_return_fit -> read_video_improved -> ellipse_to_circle -> correct_image(disk / 65536) ->
    single_image_process, with options['_nolog'] (no files)."""
    SHG_MAIN, Solex_recon, outputs = pkg
    from oracle import shg_oracle as orc
    from solex_ser_recon_en_amd import ellipse_to_circle as e2c
    from solex_ser_recon_en_amd import solex_util
    from solex_ser_recon_en_amd.video_reader import all_video_reader
    g, frames, path = scan
    rdr = all_video_reader(path)
    assert rdr.frames.shape == (400, 400, 32) and rdr.means.shape == (400,)
    options = SHG_MAIN.default_options()
    options.update(_nolog=True, shift=[options['ellipse_fit_shift']], basefich0=path[:-4])
    hdr = solex_util.make_header(rdr)
    mean_img, fit, y1, y2 = solex_util.compute_mean_return_fit(rdr, options, hdr, rdr.iw, rdr.ih, path[:-4])
    ref_mean, ref_max = orc.compute_mean_max(orc.SerReader(frames))
    np.testing.assert_array_equal(np.asarray(mean_img), ref_mean)
    ref_fit, ry1, ry2, _, _ = orc.line_fit(ref_mean, ref_max)
    assert (y1, y2) == (ry1, ry2) and fit.shape == (400, 4)
    np.testing.assert_allclose(fit, ref_fit, rtol=0, atol=1e-9)
    assert solex_util.detect_bord(mean_img, axis=1) == orc.detect_bord(ref_mean, axis=1)
    assert solex_util.detect_bord(mean_img, axis=0) == orc.detect_bord(ref_mean, axis=0)
    rdr.reset()
    disk_list, ih, iw, n = solex_util.read_video_improved(rdr, fit, options)
    assert (ih, iw, n, len(disk_list)) == (400, 32, 400, 1) and disk_list[0].dtype == np.uint16
    want = orc.extract_columns(orc.SerReader(frames), fit, options['shift'])
    np.testing.assert_array_equal(np.asarray(disk_list[0]), want[0])
    fix_img, circle, ratio, phi, borders = e2c.ellipse_to_circle(disk_list[0], options, path[:-4])
    again, circle2, mat3 = e2c.correct_image(np.asarray(disk_list[0]) / 65536, phi, ratio, np.array([-1.0, -1.0]), -1.0, options)
    np.testing.assert_array_equal(np.asarray(again), np.asarray(fix_img))          # float disk/65536 input, as the reference passes it
    with pytest.raises(ValueError):
        e2c.correct_image(np.random.default_rng(0).random((8, 8)), 0.0, 1.0, np.array([-1.0, -1.0]), -1.0, options)
    cc, protus = Solex_recon.single_image_process(fix_img, hdr, options, circle, borders, path[:-4] + '_x', (y1, y2))
    assert np.asarray(cc).shape == np.asarray(fix_img).shape and np.asarray(cc).dtype == np.uint16
    # rescale_brightness keeps its signature, including the 8-bit form clahe_apply uses
    img8 = (np.asarray(fix_img) >> 8).astype(np.uint8)
    np.testing.assert_array_equal(solex_util.rescale_brightness(img8, 10.0, 200.0), orc.rescale_brightness(img8, 10.0, 200.0))
    np.testing.assert_array_equal(np.asarray(solex_util.rescale_brightness(fix_img, 100.0, 50000.0, alpha=0.9)),
                                  orc.rescale_brightness(np.asarray(fix_img), 100.0, 50000.0, alpha=0.9))
    np.testing.assert_array_equal(solex_util.reject_outliers(np.array([1.0, 1.1, 0.9, 50.0])), orc.reject_outliers(np.array([1.0, 1.1, 0.9, 50.0])))


@pytest.mark.parametrize('flags', ['-cf', '-f'])
def test_two_ranks_doppler_stack_is_dealt_and_identical(pkg, scan, tmp_path, flags):
    """Frame-sharded Doppler stack (-w-3:3:1, 7 requested disks): after the gather rank 0 fits the limb and
    broadcasts the geometry, the disks are dealt round-robin, each rank writes its own products; the union of the
    files equals the single-process run bit for bit."""
    import subprocess
    import sys
    g, frames, path = scan
    from solex_ser_recon_en_amd import fits_io, png_io
    one = tmp_path / 'one'
    two = tmp_path / 'two'
    one.mkdir(); two.mkdir()
    for d in (one, two):
        synth.write_ser(str(d / 'scan.ser'), frames)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=repo, SHG_DIST_BACKEND='gloo', MPLBACKEND='Agg')
    subprocess.run([sys.executable, '-m', 'solex_ser_recon_en_amd.SHG_MAIN', flags, '-w-3:3:1', str(one / 'scan.ser')], check=True,
                   env=env, cwd=repo, stdout=subprocess.DEVNULL, timeout=900)
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                    '--master-port', str(free_port()), '-m', 'solex_ser_recon_en_amd.SHG_MAIN', flags, '-w-3:3:1', str(two / 'scan.ser')],
                   check=True, env=env, cwd=repo, stdout=subprocess.DEVNULL, timeout=900)
    names = sorted(os.listdir(str(one)))
    assert names == sorted(os.listdir(str(two)))
    assert sum(n.endswith('_clahe.png') for n in names) == 7
    for name in names:
        a, b = str(one / name), str(two / name)
        if name.endswith('_clahe.png') or name.endswith('_protus.png') or name.endswith('contrast.png') or name.endswith('contrasted.png'):
            np.testing.assert_array_equal(png_io.read_png_gray(a), png_io.read_png_gray(b), err_msg=name)
        elif name.endswith('.fits'):
            np.testing.assert_array_equal(fits_io.read_fits_u16(a)[0], fits_io.read_fits_u16(b)[0], err_msg=name)


def test_two_ranks_sharded_scan_equals_one_rank(pkg, scan, tmp_path):
    """The sharded orchestration end to end: two processes (gloo, both on this GPU) each decode and reduce half
    of the frames of ONE file, exchange sum/max and disk columns, rank 0 post-processes and writes.  Every
    product must equal the single-process run bit for bit (integer reductions are order independent)."""
    import subprocess
    import sys
    SHG_MAIN, Solex_recon, outputs = pkg
    g, frames, path = scan
    from solex_ser_recon_en_amd import fits_io, png_io
    one = tmp_path / 'one'
    two = tmp_path / 'two'
    one.mkdir(); two.mkdir()
    for d in (one, two):
        synth.write_ser(str(d / 'scan.ser'), frames)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=repo, SHG_DIST_BACKEND='gloo', MPLBACKEND='Agg')
    args = ['-cf', '-w-2,0', str(one / 'scan.ser')]
    subprocess.run([sys.executable, '-m', 'solex_ser_recon_en_amd.SHG_MAIN'] + args, check=True, env=env, cwd=repo,
                   stdout=subprocess.DEVNULL, timeout=600)
    args2 = ['-cf', '-w-2,0', str(two / 'scan.ser')]
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                    '--master-port', str(free_port()), '-m', 'solex_ser_recon_en_amd.SHG_MAIN'] + args2, check=True, env=env, cwd=repo,
                   stdout=subprocess.DEVNULL, timeout=600)
    names = sorted(os.listdir(str(one)))
    assert names == sorted(os.listdir(str(two))), 'rank 1 must not write anything, rank 0 everything'
    assert 'scan_shift=-2_clahe.png' in names and 'scan_shift=0_raw.fits' in names and 'scan_mean.fits' in names
    for name in names:
        a, b = str(one / name), str(two / name)
        if name.endswith('.png'):
            np.testing.assert_array_equal(png_io.read_png_gray(a), png_io.read_png_gray(b), err_msg=name)
        elif name.endswith('.fits'):
            np.testing.assert_array_equal(fits_io.read_fits_u16(a)[0], fits_io.read_fits_u16(b)[0], err_msg=name)


def test_two_ranks_series_of_sharded_scans_overlaps_and_equals_one_rank(pkg, scan, tmp_path):
    """SHG_DISTRIBUTE=frames with several files: every file's frames are sharded over the two ranks; file k's owner (rank k mod 2)
    post-processes it on a thread of its own while both ranks already read files k + 1 and k + 2 (two reading threads, their
    collectives in one order).  Three different files must come out like the single-process run, each written by its owner only."""
    import subprocess
    import sys
    g, frames, path = scan
    from solex_ser_recon_en_amd import png_io
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=repo, SHG_DIST_BACKEND='gloo', MPLBACKEND='Agg', SHG_DISTRIBUTE='frames')
    dirs = {}
    for tag in ('one', 'two'):
        d = tmp_path / tag
        d.mkdir()
        for j in range(3):
            synth.write_ser(str(d / ('scan%d.ser' % j)), np.roll(frames, 3 * j, axis=0) if j else frames)
        dirs[tag] = d
    files = lambda d: [str(d / ('scan%d.ser' % j)) for j in range(3)]      # noqa: E731
    subprocess.run([sys.executable, '-m', 'solex_ser_recon_en_amd.SHG_MAIN', '-c'] + files(dirs['one']), check=True,
                   env=dict(env, SHG_DISTRIBUTE='auto'), cwd=repo, stdout=subprocess.DEVNULL, timeout=900)
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                    '--master-port', str(free_port()), '-m', 'solex_ser_recon_en_amd.SHG_MAIN', '-c'] + files(dirs['two']),
                   check=True, env=env, cwd=repo, stdout=subprocess.DEVNULL, timeout=900)
    names = sorted(os.listdir(str(dirs['one'])))
    assert names == sorted(os.listdir(str(dirs['two'])))
    assert sum(n.endswith('_clahe.png') for n in names) == 3
    for name in names:
        if name.endswith('.png'):
            np.testing.assert_array_equal(png_io.read_png_gray(str(dirs['one'] / name)), png_io.read_png_gray(str(dirs['two'] / name)), err_msg=name)


def test_two_ranks_series_returns_one_entry_per_scan_and_two_collectives_each(pkg, scan, tmp_path):
    """solex_do_work(distribute='frames', return_results=True) over five different scans on two ranks (gloo, one GPU): the returned
    list has one entry per task on every rank -- scan k's products on rank k mod 2, None on the other (round 5 dropped the Nones:
    the entries could not be matched to files) -- the products are the one-process run's bit for bit, and the series took two
    collectives per scan plus the one word at its end (two scans being read at a time, tests/series_worker.py)."""
    import subprocess
    import sys
    SHG_MAIN, Solex_recon, outputs = pkg
    g, frames, path = scan
    files = []
    for j in range(5):
        f = str(tmp_path / ('scan%d.ser' % j))
        synth.write_ser(f, np.roll(frames, 5 * j, axis=0) if j else frames)
        files.append(f)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=repo, SHG_DIST_BACKEND='gloo', MPLBACKEND='Agg')
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                    '--master-port', str(free_port()), os.path.join(repo, 'tests', 'series_worker.py'), str(tmp_path)] + files,
                   check=True, env=env, cwd=repo, timeout=900)
    ranks = [np.load(str(tmp_path / ('series_rank%d.npz' % r))) for r in range(2)]
    for r, rep in enumerate(ranks):
        assert int(rep['n_entries']) == 5
        assert list(rep['held']) == [i for i in range(5) if i % 2 == r]
        assert int(rep['collectives']) == 2 * 5 + 1
    for i, f in enumerate(files):
        opts = SHG_MAIN.default_options()
        opts.update(_nolog=True)
        (res,) = Solex_recon.solex_do_work([(f, opts)], True, return_results=True)
        (cc, protus), = res
        np.testing.assert_array_equal(ranks[i % 2]['cc_%d' % i], np.asarray(cc))
        np.testing.assert_array_equal(ranks[i % 2]['protus_%d' % i], np.asarray(protus))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='the RCCL (nccl) branch needs two GPUs; gloo covers the logic on one')
def test_rccl_two_gpus_sharded_doppler_and_folder(pkg, scan, tmp_path):
    """torch.distributed.run with the default backend (nccl = RCCL over xGMI), one process per GPU: a frame-sharded scan,
    a dealt Doppler stack and a folder of scans.  Every rank's stack and products live on its own device (LOCAL_RANK),
    and the products equal the single-GPU run bit for bit (tests/rccl_worker.py)."""
    import subprocess
    import sys
    SHG_MAIN, Solex_recon, outputs = pkg
    g, frames, path = scan
    second = str(tmp_path / 'other.ser')
    synth.write_ser(second, frames[::-1].copy())                              # the same scan taken in the other direction
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=repo, MPLBACKEND='Agg')
    env.pop('SHG_DIST_BACKEND', None)
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                    '--master-port', str(free_port()), os.path.join(repo, 'tests', 'rccl_worker.py'), str(tmp_path), path, second],
                   check=True, env=env, cwd=repo, timeout=900)
    r0, r1 = np.load(str(tmp_path / 'rank0.npz')), np.load(str(tmp_path / 'rank1.npz'))

    def single(file, **kw):
        opts = SHG_MAIN.default_options()
        opts.update(_nolog=True, **kw)
        (res,) = Solex_recon.solex_do_work([(file, opts)], True, return_results=True)
        return [np.asarray(cc) for cc, _ in res]
    np.testing.assert_array_equal(r0['sharded_cc'], single(path)[0])
    stack = single(path, shift=[-2, 0, 3])                                     # processing order 0, -2, 3: dealt to ranks 0, 1, 0
    assert int(r0['doppler_n']) == 2 and int(r1['doppler_n']) == 1
    np.testing.assert_array_equal(r0['doppler_cc_0'], stack[0])
    np.testing.assert_array_equal(r1['doppler_cc_0'], stack[1])
    np.testing.assert_array_equal(r0['doppler_cc_1'], stack[2])
    np.testing.assert_array_equal(r0['folder_cc_0'], single(path)[0])          # file 0 -> rank 0, file 1 -> rank 1
    np.testing.assert_array_equal(r1['folder_cc_0'], single(second)[0])


@pytest.mark.parametrize('n,w,h,bits,extra', [
    (220, 800, 120, 8, {}),                                            # BASELINE configs[0] shape, stored rotated
    (220, 120, 800, 8, {'shift': [0, 5]}),                             # the same shape stored un-rotated (Width < Height)
    (330, 640, 64, 16, {'flip_x': True, 'fixed_width': 500, 'delta_radius': 7}),
    (260, 720, 96, 16, {'transversalium': False, 'img_rotate': 270, 'disk_display': False}),
])
def test_whole_flow_other_shapes_vs_oracle(pkg, n, w, h, bits, extra):
    SHG_MAIN, Solex_recon, outputs = pkg
    from solex_ser_recon_en_amd.video_reader import array_reader
    frames = synth.synth_frames_numpy(n, w, h, bits, seed=n, tilt=0.004, curv=3e-5)
    opts = SHG_MAIN.default_options()
    opts.update(extra, _nolog=True)
    disk_list, bounds, hdr = Solex_recon.solex_read(array_reader(torch.from_numpy(frames).cuda()), opts)
    results = Solex_recon.solex_process(opts, disk_list, bounds, hdr)
    with np.errstate(all='ignore'):
        want = po.run(frames, extra)
    for got, ref in zip(disk_list, want['read']['disks']):
        np.testing.assert_array_equal(np.asarray(got), ref)
    requested = [s for s in opts['shift'] if s in opts['shift_requested']]
    for shift, (cc, protus) in zip(requested, results):
        flips('cc', cc, want['results'][shift]['cc'])
        flips('protus', protus, want['results'][shift]['protus'])


def test_bad_files_raise_like_the_reference(pkg, tmp_path):
    SHG_MAIN, Solex_recon, outputs = pkg
    from solex_ser_recon_en_amd.video_reader import video_reader
    frames = synth.synth_frames_numpy(6, 64, 16, 16, seed=0)
    good = str(tmp_path / 'good.ser')
    synth.write_ser(good, frames)
    raw = open(good, 'rb').read()
    cut = str(tmp_path / 'cut.ser')
    open(cut, 'wb').write(raw[:-100])                                  # shorter than its header says
    with pytest.raises(Exception, match='shorter than its header'):
        video_reader(cut).device_stack()
    empty = str(tmp_path / 'empty.ser')
    open(empty, 'wb').write(synth.ser_header(64, 16, 16, 0))           # FrameCount = 0
    with pytest.raises(Exception, match='no frames'):
        video_reader(empty).device_stack()
    stack = video_reader(good).device_stack()
    np.testing.assert_array_equal(stack.cpu().numpy(), frames)
    part = video_reader(good, frame_range=(2, 5)).device_stack()      # a rank's block of a sharded scan
    np.testing.assert_array_equal(part.cpu().numpy(), frames[2:5])


def test_crop_to_width_vs_reference(pkg, golden):
    """G11 through the product: crop_to_width = the crop / pad block of single_image_process (Solex_recon.py:155-171)."""
    from tests.test_oracle_golden import _crop_cases
    _, Solex_recon, _ = pkg
    g = golden('g11_crop')
    for k, img, cercle, fw, sq in _crop_cases(g):
        opts = {'fixed_width': fw, 'crop_width_square': sq}
        (out,), c2 = Solex_recon.crop_to_width([img], cercle, opts)
        np.testing.assert_array_equal(np.asarray(out), g['out_%d' % k], err_msg='case %d' % k)
        np.testing.assert_array_equal(np.array(c2, dtype=np.float64), g['cercle_%d' % k], err_msg='case %d' % k)


def test_solex_read_shift_order_vs_reference(pkg, golden, tmp_path):
    """G12 through the product: options['shift'], options['shift_requested'], bounds and the raw disks of solex_read."""
    import hashlib
    SHG_MAIN, Solex_recon, _ = pkg
    g = golden('g12_shift_order')
    frames = synth.synth_frames_numpy(int(g['param_n']), int(g['param_w']), int(g['param_h']), int(g['param_bits']),
                                      seed=int(g['param_seed']), tilt=float(g['param_tilt']), curv=float(g['param_curv']))
    path = str(tmp_path / 'scan.ser')
    synth.write_ser(path, frames)
    for k in range(int(g['n'])):
        opts = SHG_MAIN.default_options()
        opts.update(shift=[int(s) for s in g['request_%d' % k]], ellipse_fit_shift=int(g['efs_%d' % k]), _nolog=True)
        disks, bounds, hdr = Solex_recon.solex_read(path, opts)
        assert opts['shift'] == [int(s) for s in g['shift_%d' % k]]
        assert list(opts['shift_requested']) == [int(s) for s in g['shift_requested_%d' % k]]
        assert tuple(int(b) for b in bounds) == tuple(int(b) for b in g['bounds_%d' % k])
        got = np.stack([np.frombuffer(hashlib.sha256(np.ascontiguousarray(np.asarray(d)).tobytes()).digest(), np.uint8) for d in disks])
        np.testing.assert_array_equal(got, g['disk_sha256_%d' % k])


# ---- uncompressed AVI input (video_reader.py:68-80, 111-113) -------------------------------------------------
def test_avi_device_stack_matches_oracle(pkg, tmp_path):
    """Every uncompressed layout through the raw upload + shg_unpack_dib_frames, against the oracle's idx1-based decode;
    includes a sharded frame range and chunked uploads (chunk_bytes smaller than the file)."""
    from oracle import shg_oracle as orc
    from solex_ser_recon_en_amd import ops
    from solex_ser_recon_en_amd.video_reader import video_reader
    from tests.test_host_cpu import AVI_CASES, avi_case
    path = str(tmp_path / 'scan.avi')
    for layout, kw in AVI_CASES:
        for h, w in [(9, 14), (37, 21)]:
            avi_case(path, layout, kw, n=11, h=h, w=w, seed=h)
            want = orc.AviReader(path).raw_frames()
            got = ops.stack_to_host(video_reader(path).device_stack())
            assert got.dtype == np.uint8
            np.testing.assert_array_equal(got, want, err_msg='%s %s %dx%d' % (layout, kw, h, w))
            rdr = video_reader(path, frame_range=(3, 9))
            part = ops.stack_to_host(rdr.device_stack(chunk_bytes=2 * h * ((w * 3 + 3) // 4 * 4)))
            np.testing.assert_array_equal(part, want[3:9])


@pytest.mark.parametrize('layout', ['Y800', 'pal8'])
def test_solex_read_from_avi_equals_ser(pkg, tmp_path, layout):
    """The same 8-bit scan as SER and as uncompressed AVI: identical raw disks, bounds and products; both equal the oracle."""
    SHG_MAIN, Solex_recon, outputs = pkg
    frames = synth.synth_frames_numpy(400, 400, 32, 8, seed=5, tilt=0.01, curv=5e-5)
    ser, avi = str(tmp_path / 'scan.ser'), str(tmp_path / 'scan.avi')
    synth.write_ser(ser, frames)
    synth.write_avi(avi, frames, layout)
    want = po.run(frames, {})
    results = {}
    for path in (ser, avi):
        opts = SHG_MAIN.default_options()
        opts.update(_nolog=True)
        disks, bounds, hdr = Solex_recon.solex_read(path, opts)
        for got, ref in zip(disks, want['read']['disks']):
            np.testing.assert_array_equal(np.asarray(got), ref)
        assert tuple(int(b) for b in bounds) == (want['read']['y1'], want['read']['y2'])
        out = Solex_recon.solex_process(opts, disks, bounds, hdr)
        outputs.flush()
        results[path] = [np.asarray(x) for x in out[0]]
    for a, b in zip(results[ser], results[avi]):
        np.testing.assert_array_equal(a, b)
    flips('AVI scan cc', results[avi][0], want['results'][0]['cc'])


VARIANTS = {
    'off_centre': dict(scene=dict(cx=230.0, cy=180.0, ax=150.0, ay=150.0)),
    'elongated_scan': dict(scene=dict(ax=0.30 * 400, ay=0.44 * 400), options={'shift': [1]}),          # fast scan: ratio well below 1
    'slow_scan': dict(n=520, scene=dict(ax=0.46 * 520, ay=0.36 * 400)),                               # ratio above 1
    'noisy': dict(scene=dict(noise=0.012, depth=0.6), options={'crop_width_square': True}),
    'faint_8bit': dict(bits=8, scene=dict(gain=0.5, sky=0.04)),
    'tilted_line': dict(tilt=0.03, curv=-4e-5, options={'shift': [-3, 3]}),
    'rotated_180': dict(options={'img_rotate': 180, 'delta_radius': 3}),
    'rotated_270_no_disc': dict(options={'img_rotate': 270, 'disk_display': False, 'fixed_width': 380}),
    'no_transversalium_flip': dict(options={'transversalium': False, 'flip_x': True, 'trans_strength': 51}),
    'short_trend_window': dict(options={'trans_strength': 21, 'shift': [0, 10]}),
    # an active sun: two spots, a plage, a prominence off the limb, dust lines on the slit (transversalium)
    'active_sun_dusty_slit': dict(scene=dict(spots=[(150.0, 170.0, 9.0, 9.0, 0.7), (260.0, 240.0, 5.0, 6.0, 0.5),
                                                    (210.0, 120.0, 14.0, 10.0, -0.25), (372.0, 200.0, 6.0, 14.0, -4.0)]),
                                  row_gain='dust'),
}


@pytest.mark.parametrize('name', sorted(VARIANTS))
def test_scene_variants_vs_oracle(pkg, name):
    """Scans that differ from the golden scene (disk position / shape, noise, bit depth, line tilt): the whole
    flow on the GPU against the CPU oracle.  Raw disks and bounds bit-exact, products <= 1 LSB on <= 4 pixels."""
    SHG_MAIN, Solex_recon, outputs = pkg
    from solex_ser_recon_en_amd.video_reader import array_reader
    v = VARIANTS[name]
    row_gain = None
    if v.get('row_gain') == 'dust':
        row_gain = 1 + 0.005 * np.random.default_rng(5).standard_normal(400)
        row_gain[[97, 98, 181, 260, 261, 262]] *= [0.93, 0.95, 0.9, 0.96, 0.92, 0.97]
    frames = synth.synth_frames_numpy(v.get('n', 400), 400, 32, v.get('bits', 16), seed=11, tilt=v.get('tilt', 0.01),
                                      curv=v.get('curv', 5e-5), scene=v.get('scene'), row_gain=row_gain)
    extra = v.get('options', {})
    want = po.run(frames, extra)
    opts = SHG_MAIN.default_options()
    opts.update(extra, _nolog=True)
    rdr = array_reader(torch.from_numpy(frames).cuda())
    disks, bounds, hdr = Solex_recon.solex_read(rdr, opts)
    assert opts['shift'] == want['read']['shifts']
    assert tuple(int(b) for b in bounds) == (want['read']['y1'], want['read']['y2'])
    for got, ref in zip(disks, want['read']['disks']):
        np.testing.assert_array_equal(np.asarray(got), ref)
    results = Solex_recon.solex_process(opts, disks, bounds, hdr)
    outputs.flush()
    np.testing.assert_allclose(opts['ratio_fixe'], want['geometry']['ratio'], rtol=1e-9)
    requested = [s for s in opts['shift'] if s in opts['shift_requested']]
    for shift, (cc, protus) in zip(requested, results):
        flips('cc', cc, want['results'][shift]['cc'])
        flips('protus', protus, want['results'][shift]['protus'])


@pytest.mark.parametrize('name', ['zeros', 'noise_only', 'ten_frames', 'half_scan', 'tiny_disk'])
def test_pathological_scans_fail_like_the_reference(pkg, name):
    """Scans the reference cannot process (solex_util.py:245-246 needs >= 3 distinct residuals, ellipse_to_circle.py:
    245-263 needs a closed limb) must raise the same exception type here -- no hang, no garbage -- and a half
    scan must still come out like the oracle's."""
    SHG_MAIN, Solex_recon, outputs = pkg
    from solex_ser_recon_en_amd.video_reader import array_reader
    base = synth.synth_frames_numpy(400, 400, 32, 16, seed=1, tilt=0.01, curv=5e-5)
    frames = {'zeros': np.zeros_like(base),
              'noise_only': np.random.default_rng(0).integers(0, 3000, base.shape).astype(np.uint16),
              'ten_frames': base[195:205].copy(), 'half_scan': base[:200].copy(),
              'tiny_disk': synth.synth_frames_numpy(400, 400, 32, 16, seed=1, scene=dict(ax=20.0, ay=20.0))}[name]
    opts = SHG_MAIN.default_options()
    opts.update(_nolog=True)
    try:
        with np.errstate(all='ignore'):
            want = po.run(frames, {})
        expected = None
    except Exception as e:          # noqa: BLE001
        want, expected = None, type(e)
    task = [(array_reader(torch.from_numpy(frames).cuda()), opts)]
    if expected is not None:
        with pytest.raises(expected):
            Solex_recon.solex_do_work(task, True, return_results=True)
        outputs.flush()
        torch.cuda.synchronize()                                    # the device is still usable
    else:
        (results,) = Solex_recon.solex_do_work(task, True, return_results=True)
        outputs.flush()
        flips('cc', results[0][0], want['results'][0]['cc'])
    assert (expected is None) == (name == 'half_scan')


def test_cli_accepts_an_uncompressed_avi(pkg, tmp_path):
    """The front door with an .avi argument (CLI_handler.py:123 lets SER and AVI through): same CLAHE product as the SER."""
    SHG_MAIN, Solex_recon, outputs = pkg
    from solex_ser_recon_en_amd import png_io
    frames = synth.synth_frames_numpy(400, 400, 32, 8, seed=7, tilt=0.01, curv=5e-5)
    ser, avi = str(tmp_path / 'a.ser'), str(tmp_path / 'b.avi')
    synth.write_ser(ser, frames)
    synth.write_avi(avi, frames, 'pal8')
    assert SHG_MAIN.main(['-c', ser, avi]) == 0
    a = png_io.read_png_gray(str(tmp_path / 'a_shift=0_clahe.png'))
    b = png_io.read_png_gray(str(tmp_path / 'b_shift=0_clahe.png'))
    assert a.dtype == np.uint16 and a.shape[0] == 400
    np.testing.assert_array_equal(a, b)


ALL_PNG = ['_shift=0_clahe.png', '_shift=0_protus.png', '_shift=0_uncontrasted.png', '_shift=0_high_contrast.png']
PLOTS = ['_spectral_line_data.png', '_shift=10_ellipse_fit.png', '_shift=0_transversalium_correction.png']


@pytest.mark.parametrize('extra,present,absent', [
    ({'clahe_only': True}, ['_shift=0_clahe.png'], ALL_PNG[1:] + PLOTS),
    ({'protus_only': True}, ['_shift=0_protus.png'], [ALL_PNG[0]] + ALL_PNG[2:] + PLOTS),
    ({'clahe_only': True, 'protus_only': True}, ALL_PNG[:2], ALL_PNG[2:] + PLOTS),
    ({'disk_display': False}, ALL_PNG + PLOTS, []),                      # the CLI's -p: products unchanged, no black disc
])
def test_product_selection_options(pkg, scan, tmp_path, extra, present, absent):
    """clahe_only / protus_only choose the products as solex_util.py:556-566 and switch the diagnostics off (:263,
    :482, ellipse_to_circle.py:316); the CLI's -p is disk_display (CLI_handler.py:17), not protus_only."""
    SHG_MAIN, Solex_recon, outputs = pkg
    g, frames, path = scan
    work = str(tmp_path / 'scan.ser')
    synth.write_ser(work, frames)
    opts = SHG_MAIN.default_options()
    opts.update(extra)
    Solex_recon.solex_do_work([(work, opts)], True)
    base = work[:-4]
    for suffix in present + ['_log.txt']:
        assert os.path.exists(base + suffix), 'missing ' + suffix
    for suffix in absent:
        assert not os.path.exists(base + suffix), 'unexpected ' + suffix


def test_output_dir_redirects_every_file(pkg, scan, tmp_path):
    """options['output_dir'] (solex_util.py:60-63): same file names, other folder; nothing is written next to the scan."""
    SHG_MAIN, Solex_recon, outputs = pkg
    g, frames, path = scan
    src, dst = tmp_path / 'in', tmp_path / 'out'
    src.mkdir()
    dst.mkdir()
    work = str(src / 'scan.ser')
    synth.write_ser(work, frames)
    opts = SHG_MAIN.default_options()
    opts.update(output_dir=str(dst), clahe_only=True, save_fit=True)
    Solex_recon.solex_do_work([(work, opts)], True)
    assert sorted(os.listdir(str(src))) == ['scan.ser']
    names = sorted(os.listdir(str(dst)))
    for expected in ('scan_log.txt', 'scan_mean.fits', 'scan_shift=0_raw.fits', 'scan_shift=0_clahe.png', 'scan_shift=0_clahe.fits'):
        assert expected in names, (expected, names)


@pytest.mark.parametrize('tag', ['A', 'B'])
def test_log_file_text_matches_the_reference(pkg, scan, tmp_path, tag):
    """<base>_log.txt line for line against the reference's own run (g14, shim mode), time stamps aside: same lines,
    same order, same number formatting (NumPy's str of the fit / matrix / centre)."""
    SHG_MAIN, Solex_recon, outputs = pkg
    g, frames, path = scan
    work = str(tmp_path / 'scan.ser')
    synth.write_ser(work, frames)
    opts = SHG_MAIN.default_options()
    opts.update(SCENARIOS[tag])
    Solex_recon.solex_do_work([(work, opts)], True)
    text = open(work[:-4] + '_log.txt').read().splitlines()
    assert text[0].startswith('start time: ') and text[-1].startswith('end time: ')
    got = [ln for ln in text if not ln.startswith(('start time', 'end time'))]
    want = str(g[tag + '_log']).splitlines()
    assert got == want, '\n'.join(['--- got'] + got + ['--- want'] + want)


ONE_CALL_CASES = {
    'plain': {},
    'doppler_flip_square': {'shift': [-2, 0, 3], 'flip_x': True, 'crop_width_square': True},
    'fit_shift_requested': {'shift': [10, 0, 1]},                   # the ellipse-fit disk is a product too
    'fixed_ratio': {'ratio_fixe': 1, 'fixed_width': 300, 'disk_display': False, 'img_rotate': 90},
    'fixed_slant': {'slant_fix': 1.5, 'shift': [0, 2]},
    'fixed_both': {'ratio_fixe': 0.97, 'slant_fix': -2.0, 'delta_radius': 4},
    'no_transversalium': {'transversalium': False, 'fixed_width': 520},
    'short_trend': {'trans_strength': 21, 'shift': [0, 10]},
    'window_from_rows': {'trans_strength': 501},                     # fewer sunlit rows than that: not the window the request carries taps for
}


def _run_route(pkg, frames, tmp_path, name, extra, one_call, monkeypatch, files):
    SHG_MAIN, Solex_recon, outputs = pkg
    from solex_ser_recon_en_amd.video_reader import array_reader
    monkeypatch.setenv('SHG_SCAN_CALL', '1' if one_call else '0')
    opts = SHG_MAIN.default_options()
    opts.update(extra)
    if files:
        d = tmp_path / ('%s_%d' % (name, one_call))
        d.mkdir()
        work = str(d / 'scan.ser')
        synth.write_ser(work, frames)
        opts.update(save_fit=True)
        src = work
    else:
        opts.update(_nolog=True)
        src = array_reader(torch.from_numpy(frames).cuda())
    err = None
    res = None
    try:
        (res,) = Solex_recon.solex_do_work([(src, opts)], True, return_results=True)
    except Exception as e:      # noqa: BLE001
        err = e
    outputs.flush()
    torch.cuda.synchronize()
    images = None if res is None else [(np.asarray(cc), np.asarray(pr)) for cc, pr in res]
    listing, log = None, None
    if files:
        listing = sorted(os.listdir(str(d)))
        log = [ln for ln in open(work[:-4] + '_log.txt').read().splitlines() if not ln.startswith(('start time', 'end time'))]
    return images, opts, err, listing, log, (str(d) if files else None)


@pytest.mark.parametrize('name', sorted(ONE_CALL_CASES))
def test_one_call_route_equals_the_stage_route(pkg, scan, tmp_path, monkeypatch, name):
    """shg_scan_file (one C call per file, the default of solex_do_work) against the stage-by-stage route
    (SHG_SCAN_CALL=0): products bit for bit, the same `options` side effects, and -- with files on -- the same set of
    files, the same log text and byte-identical FITS / PNG products."""
    g, frames, path = scan
    extra = ONE_CALL_CASES[name]
    files = name in ('plain', 'doppler_flip_square', 'fixed_ratio')
    a_img, a_opts, a_err, a_ls, a_log, a_dir = _run_route(pkg, frames, tmp_path, name, extra, False, monkeypatch, files)
    b_img, b_opts, b_err, b_ls, b_log, b_dir = _run_route(pkg, frames, tmp_path, name, extra, True, monkeypatch, files)
    assert a_err is None and b_err is None, (a_err, b_err)
    assert len(a_img) == len(b_img) and len(a_img) == len(extra.get('shift', [0]))
    for (cc1, p1), (cc2, p2) in zip(a_img, b_img):
        np.testing.assert_array_equal(cc1, cc2)
        np.testing.assert_array_equal(p1, p2)
    for key in ('ratio_fixe', 'slant_fix', 'shift', 'shift_requested'):
        assert a_opts[key] == b_opts[key], key
    if a_opts['transversalium']:
        np.testing.assert_array_equal(a_opts['_transversalium_cache'], b_opts['_transversalium_cache'])
    if files:
        assert a_ls == b_ls and a_log == b_log
        for f in a_ls:
            if f.endswith(('.fits', '_clahe.png', '_protus.png', '_uncontrasted.png', '_high_contrast.png')):
                assert open(os.path.join(a_dir, f), 'rb').read() == open(os.path.join(b_dir, f), 'rb').read(), f


def test_one_call_route_resumes_when_its_arena_guess_was_too_small(pkg, scan, monkeypatch):
    """The corrected images' size is only known after the limb fit: a first guess that is too small makes shg_scan_file
    return SHG_E_WORKSPACE with the sizes it needs, and the second call resumes at the warp -- same products."""
    from solex_ser_recon_en_amd import stages
    g, frames, path = scan
    want, _, err, _, _, _ = _run_route(pkg, frames, None, 'resume', {'shift': [0, 2]}, True, monkeypatch, False)
    assert err is None
    monkeypatch.setattr(stages, 'FIRST_GUESS_SCALE', 0.1)
    stages._arena_hint.clear()
    got, _, err, _, _, _ = _run_route(pkg, frames, None, 'resume', {'shift': [0, 2]}, True, monkeypatch, False)
    assert err is None and len(stages._arena_hint) == 1
    for (cc1, p1), (cc2, p2) in zip(want, got):
        np.testing.assert_array_equal(cc1, cc2)
        np.testing.assert_array_equal(p1, p2)
    stages._arena_hint.clear()


@pytest.mark.parametrize('name', ['noise_only', 'tiny_disk', 'few_rows'])
def test_one_call_route_fails_like_the_stage_route(pkg, tmp_path, monkeypatch, name):
    """A scan that fails half-way: the same exception type and message from both routes, and the same log lines up to
    the failure (the one-call route writes them after the call, from what the call reached)."""
    base = synth.synth_frames_numpy(400, 400, 32, 16, seed=1, tilt=0.01, curv=5e-5)
    frames = {'noise_only': np.random.default_rng(0).integers(0, 3000, base.shape).astype(np.uint16),
              'tiny_disk': synth.synth_frames_numpy(400, 400, 32, 16, seed=1, scene=dict(ax=20.0, ay=20.0)),
              'few_rows': synth.synth_frames_numpy(400, 90, 32, 16, seed=1, tilt=0.01, curv=5e-5)}[name]
    _, _, a_err, a_ls, a_log, _ = _run_route(pkg, frames, tmp_path, name, {}, False, monkeypatch, True)
    _, _, b_err, b_ls, b_log, _ = _run_route(pkg, frames, tmp_path, name, {}, True, monkeypatch, True)
    assert a_err is not None and b_err is not None
    # (a failure inside the library names the entry point the caller went through: 'shg_stage_mean_fit failed ...')
    assert type(a_err) is type(b_err) and str(a_err).split(' failed ')[-1] == str(b_err).split(' failed ')[-1]
    assert a_log == b_log


@pytest.mark.parametrize('mode', ['folder', 'sharded'])
def test_bench_launches_its_own_ranks(mode):
    """`python bench.py --gpus 2` with NO launcher around it (no WORLD_SIZE): bench.py starts the two ranks itself, before it
    touches the GPU, and rank 0 prints the one JSON line.  Both ranks share this GPU over gloo (SHG_DIST_BACKEND); small scans.
    folder: two scans per rank, no collective on the data path, and the sharded_c3 leg (a file sharded over the ranks) beside it;
    sharded: the timed scans themselves are sharded.  The line must say what ran -- world size, backend, collectives -- and the
    sharded products must be those of one rank."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(SHG_DIST_BACKEND='gloo', PYTHONPATH=repo)
    cmd = [sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--repeats', '2', '--frames', '250',
           '--width', '640', '--height', '48', '--mode', mode, '--no-extra', '--e2e-files', '2', '--c3-scans', '2', '--c3-frames', '600']
    r = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['config']['world_size'] == 2 and line['config']['backend'] == 'gloo'
    assert line['steps'] == 4 and line['warmup'] == 2 and line['value'] > 0 and line['cpu_baseline'] is None
    if mode == 'sharded':
        assert line['scaling'] == 'strong' and line['config']['mode'] == 'sharded' and line['config']['collectives_per_scan'] > 0
        par = line['parity_vs_one_rank']
        assert par['images_compared'] == 2 and par['images_that_differ'] == 0, par
    else:
        assert line['scaling'] == 'weak' and line['config']['mode'] == 'folder' and line['config']['collectives_per_scan'] == 0
        c3 = line['sharded_c3']
        assert c3['world_size'] == 2 and c3['backend'] == 'gloo' and c3['collectives_per_scan'] > 0, c3
        assert c3['parity_vs_one_rank']['images_compared'] == 2 and c3['parity_vs_one_rank']['images_that_differ'] == 0, c3
        assert line['e2e']['value'] > 0 and line['value_decode_inclusive'] == line['e2e']['value'] and line['e2e']['h2d_ceiling_GBps'] > 0
