"""Generate tests/golden/*.npz by running the REFERENCE's own functions.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg /opt/conda/bin/python3.9 oracle/capture_goldens.py [G1 G2 ...]

/opt/conda python3.9 has NumPy 1.26.4, SciPy 1.7.1, scikit-image 0.18.3, astropy
4.3.1 -- everything the reference imports except cv2, FreeSimpleGUI and
lsq-ellipse, which are stubbed as empty modules.
Reference functions that never touch those modules run UNMODIFIED ("pinned").
For functions that call cv2.blur / cv2.createCLAHE / cv2.circle / LsqEllipse the
stub module is given this repo's own restatement of that primitive ("shim
mode"): the logic around the primitive is then pinned byte for byte, the
primitive itself is not (it is marked UNPINNED in oracle/shg_oracle.py).

Only inputs and expected outputs are stored; no reference source is copied.

SHG_PIN_REAL=1 (tools/pin_cv2.py, for an interpreter that HAS opencv-python and lsq-ellipse): nothing of cv2 / ellipse is stubbed
or shimmed -- the reference then runs on the real cv2.blur / createCLAHE / circle / filter2D and LsqEllipse (only cv2.imwrite is
intercepted, to capture the arrays it is handed).  SHG_GOLDEN_OUT / SHG_REFERENCE redirect the output folder / the reference.
"""
import io
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.environ.get('SHG_GOLDEN_OUT') or os.path.join(REPO, 'tests', 'golden')
REF = os.environ.get('SHG_REFERENCE') or '/root/reference'
REAL = os.environ.get('SHG_PIN_REAL') == '1'        # the real opencv-python / lsq-ellipse instead of this repo's restatements

for _n in ('FreeSimpleGUI', 'skimage.data._fetchers') + (() if REAL else ('cv2', 'ellipse')):
    sys.modules[_n] = types.ModuleType(_n)
if not REAL:
    sys.modules['ellipse'].LsqEllipse = object
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import shg_oracle as orc                                  # noqa: E402  (shim primitives only)
import limb_oracle as limb                                # noqa: E402  (LsqEllipse shim only)
if not REAL:
    sys.modules['ellipse'].LsqEllipse = limb.LsqEllipse   # bound by `from ellipse import LsqEllipse`
from solex_ser_recon_en_amd import synth                  # noqa: E402  (input generator only)

import cv2                                                # noqa: E402  (the stub)
import video_reader as ref_vr                             # noqa: E402
import solex_util as ref_su                               # noqa: E402
import ellipse_to_circle as ref_e2c                       # noqa: E402
import Solex_recon as ref_sr                              # noqa: E402
import CLI_handler as ref_cli                             # noqa: E402

QUIET = {'output_dir': '', '_nolog': True, 'flag_display': False, 'clahe_only': True,
         'protus_only': False, 'save_fit': False}


def save(name, **arrays):
    path = os.path.join(GOLD, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', path, os.path.getsize(path), 'bytes')


def tmp_ser(frames):
    fd, path = tempfile.mkstemp(suffix='.ser')
    os.close(fd)
    synth.write_ser(path, frames)
    return path


def install_blur_shim():
    if REAL:
        return

    def blur(img, ksize):
        kw, kh = ksize
        if img.dtype == np.uint16:
            return orc.box_blur_u16(img, kw, kh)
        return orc.box_blur_f64(img, kw, kh)
    cv2.blur = blur


# --------------------------------------------------------------------------
def g1_mean_max():
    """Reference compute_mean_max + video_reader, unmodified (pinned)."""
    out = {}
    cases = [('u16_rot', 24, 64, 20, 16), ('u16_norot', 24, 20, 64, 16),
             ('u8_rot', 24, 64, 20, 8), ('u8_norot', 24, 20, 64, 8),
             ('u16_odd', 7, 37, 11, 16)]       # odd sizes: exercises the non-vector path
    for tag, n, w, h, bits in cases:
        frames = synth.synth_frames_numpy(n, w, h, bits, seed=11)
        path = tmp_ser(frames)
        rdr = ref_vr.video_reader(path)
        mean, mx = ref_su.compute_mean_max(rdr, QUIET, path[:-4])
        os.remove(path)
        out[tag + '_frames'] = frames
        out[tag + '_mean'] = mean
        out[tag + '_max'] = mx
        out[tag + '_dims'] = np.array([rdr.ih, rdr.iw, rdr.FrameCount, int(rdr.flag_rotate)])
    save('g1_mean_max', **out)


def g2_extract():
    """Reference read_video_improved, unmodified (pinned), incl. clamped columns and S=21."""
    out = {}
    for tag, n, w, h, bits in [('u16_rot', 20, 96, 40, 16), ('u16_norot', 20, 40, 96, 16),
                               ('u8_rot', 20, 96, 40, 8), ('u16_odd', 9, 37, 29, 16)]:
        frames = synth.synth_frames_numpy(n, w, h, bits, seed=5)
        path = tmp_ser(frames)
        rdr = ref_vr.video_reader(path)
        ih, iw = rdr.ih, rdr.iw
        # a curve that runs off both spectral edges so that the index clamps engage
        curve = np.linspace(-6.3, iw + 4.7, ih) + 0.37 * np.sin(np.arange(ih))
        fit = np.array([[np.floor(c), c - np.floor(c), y, c] for y, c in enumerate(curve)])
        for stag, shifts in [('s2', [10, 0]), ('s21', list(dict.fromkeys([10, 0] + list(range(-10, 11))))),
                             ('s3', [10, 0, -25])]:
            opts = dict(QUIET, shift=shifts)
            disks, _, _, _ = ref_su.read_video_improved(ref_vr.video_reader(path), fit, opts)
            out[tag + '_' + stag + '_disks'] = np.stack(disks)
            out[tag + '_' + stag + '_shifts'] = np.array(shifts)
        os.remove(path)
        out[tag + '_frames'] = frames
        out[tag + '_fit'] = fit
    save('g2_extract', **out)


def _disk_image(h, w, seed, ratio=0.8, tilt=0.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = w / 2 + 1.5, h / 2 - 0.7
    u = (xx - cx) + tilt * (yy - cy)
    r2 = (u / (0.40 * w)) ** 2 + ((yy - cy) / (0.40 * w * ratio)) ** 2
    img = np.where(r2 < 1, 0.35 + 0.65 * np.sqrt(np.clip(1 - r2, 0, 1)), 0.02)
    img = img * (1 + 0.05 * rng.standard_normal((h, w)))
    return np.clip(np.rint(img * 0.8 * 65535), 1, 65535).astype(np.uint16)


def g3_warp():
    """Reference correct_image (skimage 0.18.3 warp), unmodified (pinned)."""
    out = {}
    cases = [(0.0, 1.0), (0.0, 0.83), (0.05, 0.9), (-0.08, 1.12), (0.2, 0.75), (0.0, 1.25)]
    img = _disk_image(60, 84, 3)
    out['image_u16'] = img
    for i, (phi, ratio) in enumerate(cases):
        center = np.array([43.5, 29.3])
        fixed, circle, mat3 = ref_e2c.correct_image(img / 65536, phi, ratio, center, 24.0, dict(QUIET), print_log=False)
        out['c%d_params' % i] = np.array([phi, ratio, center[0], center[1], 24.0])
        out['c%d_out' % i] = fixed
        out['c%d_circle' % i] = np.array(circle)
        out['c%d_mat3' % i] = mat3
    # the no-ellipse call form of Solex_recon.py:123
    fixed, circle, mat3 = ref_e2c.correct_image(img / 65536, 0.0, 1.0, np.array([-1.0, -1.0]), -1.0, dict(QUIET))
    out['noellipse_out'] = fixed
    out['noellipse_circle'] = np.array(circle)
    save('g3_warp', **out)


def g4_transversalium():
    """Reference correct_transversalium2 (non-stubborn), unmodified (pinned)."""
    out = {}
    rng = np.random.default_rng(8)
    img = _disk_image(150, 160, 4, ratio=1.0)
    gain = 1 + 0.02 * rng.standard_normal(150)
    img = np.clip(img * gain[:, None], 1, 65535).astype(np.uint16)
    out['image'] = img
    circle = (81.5, 74.3, 62.0)
    borders = [18.2, 12.7, 143.9, 136.1]
    for tag, ts in [('a', 301), ('b', 21)]:
        opts = dict(QUIET, trans_strength=ts, stubborn_transversalium=False)
        res = ref_su.correct_transversalium2(img, circle, borders, opts, 0, '/tmp/x')
        out[tag + '_out'] = res
        out[tag + '_c'] = opts['_transversalium_cache']
        out[tag + '_strength'] = np.array(ts)
    out['circle'] = np.array(circle)
    out['borders'] = np.array(borders)
    # backup-bounds call form, Solex_recon.py:146
    opts = dict(QUIET, trans_strength=301, stubborn_transversalium=False)
    bb = [0, 20 + 20, img.shape[1] - 1, 130 - 20]
    res = ref_su.correct_transversalium2(img, (0, 0, 99999), bb, opts, 0, '/tmp/x')
    out['bb_out'] = res
    out['bb_c'] = opts['_transversalium_cache']
    out['bb_borders'] = np.array(bb)
    save('g4_transversalium', **out)


def g5_rescale():
    """Reference rescale_brightness + np.percentile (pinned)."""
    out = {}
    img = _disk_image(70, 90, 6)
    out['image'] = img
    bright = np.percentile(img, 99.9999)
    out['bright'] = np.array(bright)
    out['p10'] = np.array(np.percentile(img, 10))
    out['hc'] = ref_su.rescale_brightness(img, bright * 0.25, bright)
    out['protus'] = ref_su.rescale_brightness(img, 0, bright * 0.18)
    out['alpha'] = ref_su.rescale_brightness(img, 1000.0, 50000.0, alpha=0.8)
    img8 = (img >> 8).astype(np.uint8)
    out['image8'] = img8
    out['u8'] = ref_su.rescale_brightness(img8, 10.0, 200.0)
    save('g5_rescale', **out)


def g7_matrix():
    """Reference get_correction_matrix table (pinned)."""
    params = [(0.0, 1.0), (0.0, 0.8), (0.1, 0.9), (-0.2, 1.1), (0.7, 0.6), (-0.78, 1.4), (0.3, 1.0)]
    mats = np.array([ref_e2c.get_correction_matrix(p, r)[0] for p, r in params])
    thetas = np.array([ref_e2c.get_correction_matrix(p, r)[1] for p, r in params])
    save('g7_matrix', params=np.array(params), mats=mats, thetas=thetas)


def g8_fit_shim():
    """Reference compute_mean_return_fit UNMODIFIED with cv2.blur := orc.box_blur (shim mode)."""
    install_blur_shim()
    out = {}
    for tag, n, w, h, bits in [('u16_rot', 16, 180, 48, 16), ('u8_norot', 16, 48, 180, 8)]:
        frames = synth.synth_frames_numpy(n, w, h, bits, seed=2, tilt=0.031, curv=2e-4)
        path = tmp_ser(frames)
        rdr = ref_vr.video_reader(path)
        hdr = ref_su.make_header(rdr)
        mean, fit, y1, y2 = ref_su.compute_mean_return_fit(ref_vr.video_reader(path), dict(QUIET), hdr,
                                                           rdr.iw, rdr.ih, path[:-4])
        _, mx = ref_su.compute_mean_max(ref_vr.video_reader(path), dict(QUIET), path[:-4])
        os.remove(path)
        out[tag + '_frames'] = frames
        out[tag + '_mean'] = mean
        out[tag + '_max'] = mx
        out[tag + '_fit'] = fit
        out[tag + '_y'] = np.array([y1, y2])
    save('g8_fit_shim', **out)


def g9_fits():
    """astropy FITS bytes for a uint16 array with the reference's make_header cards."""
    from astropy.io import fits

    class R:
        iw, ih = 7, 5
    hdr = ref_su.make_header(R)
    hdr['NAXIS1'] = 9                                     # Solex_recon.py:65
    arr = (np.arange(45, dtype=np.uint16).reshape(5, 9) * 1500 + 7).astype(np.uint16)
    buf = io.BytesIO()
    fits.PrimaryHDU(arr, header=hdr).writeto(buf)
    save('g9_fits', array=arr, fits_bytes=np.frombuffer(buf.getvalue(), dtype=np.uint8))


def g10_cli():
    """Reference CLI_handler.treat_flag_at_cli: argv -> options table (pinned)."""
    import json
    base = {'shift': [0], 'flag_display': False, 'ratio_fixe': None, 'slant_fix': None, 'save_fit': False,
            'clahe_only': False, 'protus_only': False, 'disk_display': True, 'delta_radius': 0,
            'crop_width_square': False, 'transversalium': True, 'flip_x': False, 'fixed_width': None}
    cases = ['-w5', '-w-10:10:1', '-w1,2,-3', '-w-3:3', '-dcf', '-x', '-t', '-p', '-s', '-m', '-r1200',
             '-cfw3,4', '-r800s', '-fw-2:2:2t', '-mw7']
    table = {}
    stdout = sys.stdout
    for arg in cases:
        opts = dict(base)
        sys.stdout = io.StringIO()
        try:
            ref_cli.treat_flag_at_cli(opts, arg)
        finally:
            sys.stdout = stdout
        table[arg] = opts
    with open(os.path.join(GOLD, 'g10_cli.json'), 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print('wrote g10_cli.json')


def g6_vignette():
    """Reference removeVignette, unmodified (pinned)."""
    out = {}
    rng = np.random.default_rng(21)
    for tag, (h, w, cx, cy, r) in {'a': (300, 330, 160.4, 148.9, 120.3), 'b': (260, 420, 230.0, 120.0, 110.0)}.items():
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        r2 = ((xx - cx) ** 2 + (yy - cy) ** 2) / r ** 2
        img = np.where(r2 < 1, 0.35 + 0.65 * np.sqrt(np.clip(1 - r2, 0, 1)), 0.02) * (1 + 0.15 * (yy - cy) / h)
        img = np.clip(img * 0.7 * 65535 * (1 + 0.02 * rng.standard_normal((h, w))), 1, 65535).astype(np.uint16)
        res = ref_su.removeVignette(img, (cx, cy, r))
        out[tag + '_image'] = img
        out[tag + '_circle'] = np.array([cx, cy, r])
        out[tag + '_factor'] = res[:, 0] / img[:, 0]
        out[tag + '_out_sha256_shape'] = np.array(res.shape)
        out[tag + '_row_sample'] = res[::17, ::13]
        out[tag + '_p85_cols'] = np.percentile(img, 85, axis=0)
        out[tag + '_p85_rows'] = np.percentile(img, 85, axis=1)
    small = np.full((90, 90), 1000, np.uint16)
    assert ref_su.removeVignette(small, (45.0, 45.0, 40.0)) is small        # not enough data: returned unchanged
    save('g6_vignette', **out)


def g13_limb():
    """Real scikit-image 0.18.3: downscale_local_mean and canny on a flooded disk (pinned)."""
    import skimage.feature
    from skimage.transform import downscale_local_mean
    out = {}
    disk = _disk_image(811, 920, 12, ratio=0.9)
    out['disk_crop'] = disk[:41, :53]
    out['small_crop'] = downscale_local_mean(disk[:41, :53] / 65536, (4, 4))
    small = downscale_local_mean(disk / 65536, (4, 4))
    out['small'] = small
    install_blur_shim()
    flooded = limb.get_flood_image(small.copy())
    out['flooded'] = flooded
    for i, (sigma, lo, hi) in enumerate([(2, 0.03, 0.045), (1.5, 0.03, 0.045), (2, 20000.0, 40000.0)]):
        out['canny%d' % i] = skimage.feature.canny(image=flooded, sigma=sigma, low_threshold=lo, high_threshold=hi)
        out['canny%d_params' % i] = np.array([sigma, lo, hi])
    rng = np.random.default_rng(3)
    noisy = np.clip(small + 0.05 * rng.standard_normal(small.shape), 0, 1)
    out['noisy'] = noisy
    out['canny_noisy'] = skimage.feature.canny(image=noisy, sigma=1.0, low_threshold=0.05, high_threshold=0.12)
    save('g13_limb', **out)


def g14_pipeline():
    """The reference's solex_read + solex_process run END TO END, unmodified, in shim mode:
    cv2.blur / createCLAHE / circle and ellipse.LsqEllipse are this repo's restatements,
    cv2.imwrite captures the arrays.  Pins the orchestration (shift list, geometry state,
    stage order, crop, flip, percentiles, rescale) around the unpinned primitives.  Scenarios F and G take the
    stubborn transversalium branch with cv2.filter2D := scipy.ndimage.correlate(mode='mirror')."""
    import hashlib
    import shutil
    install_blur_shim()
    captured = {}

    class _Clahe:
        def __init__(self, clipLimit, tileGridSize):
            self.clip, self.tiles = clipLimit, tileGridSize[0]

        def apply(self, img):
            return orc.clahe(img, self.clip, self.tiles)

    if not REAL:
        cv2.createCLAHE = lambda clipLimit, tileGridSize: _Clahe(clipLimit, tileGridSize)
        cv2.circle = lambda img, center, r, color, thickness: orc.filled_circle(img, center[0], center[1], r, color)
        cv2.IMWRITE_PNG_COMPRESSION = 16

    def imwrite(path, img, params=None):
        captured[os.path.basename(path)] = np.array(img)
        return True
    cv2.imwrite = imwrite
    cv2.destroyAllWindows = lambda: None
    from scipy import ndimage
    if not REAL:
        cv2.filter2D = lambda src, ddepth, kernel: ndimage.correlate(src, kernel, mode='mirror')      # scenarios F, G (see g15_stubborn)

    base = {'shift': [0], 'flag_display': False, 'ratio_fixe': None, 'slant_fix': None, 'save_fit': False,
            'clahe_only': False, 'protus_only': False, 'disk_display': True, 'delta_radius': 0,
            'crop_width_square': False, 'transversalium': True, 'stubborn_transversalium': False,
            'trans_strength': 301, 'img_rotate': 0, 'flip_x': False, 'fixed_width': None, 'output_dir': '',
            'ellipse_fit_shift': 10, 'de-vignette': False}
    params = dict(n=400, w=400, h=32, bits=16, seed=3, tilt=0.01, curv=5e-5)
    row_gain = 1 + 0.01 * np.random.default_rng(77).standard_normal(400)
    frames = synth.synth_frames_numpy(params['n'], params['w'], params['h'], params['bits'], seed=params['seed'],
                                      tilt=params['tilt'], curv=params['curv'], row_gain=row_gain)
    out = {'row_gain': row_gain, 'frames_sha256': np.frombuffer(hashlib.sha256(frames.tobytes()).digest(), np.uint8)}
    out.update({'param_' + k: np.array(v) for k, v in params.items()})
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, 'scan.ser')
    synth.write_ser(path, frames)
    scenarios = {'A': {}, 'B': {'shift': [-2, 0, 3], 'flip_x': True, 'crop_width_square': True},
                 'C': {'ratio_fixe': 1, 'fixed_width': 300, 'disk_display': False, 'img_rotate': 90},
                 'D': {'de-vignette': True, 'shift': [0, 4]}, 'E': {'de-vignette': True, 'transversalium': False, 'crop_width_square': True},
                 'F': {'stubborn_transversalium': True, 'trans_strength': 41},
                 'G': {'stubborn_transversalium': True, 'de-vignette': True}}
    keep = {'A': ['clahe', 'protus', 'uncontrasted', 'high_contrast'], 'B': ['uncontrasted', 'clahe'], 'C': ['clahe', 'protus'],
            'D': ['clahe', 'uncontrasted'], 'E': ['clahe', 'protus'], 'F': ['clahe', 'uncontrasted'], 'G': ['clahe', 'uncontrasted']}
    for tag, extra in scenarios.items():
        captured.clear()
        opts = dict(base, **extra)
        disk_list, bounds, hdr = ref_sr.solex_read(path, opts)
        ref_sr.solex_process(opts, disk_list, bounds, hdr)
        out[tag + '_shifts'] = np.array(opts['shift'])
        out[tag + '_bounds'] = np.array(bounds)
        if tag in ('A', 'B'):                             # the text of <base>_log.txt, minus its two time stamps
            lines = [ln for ln in open(os.path.join(tmp, 'scan_log.txt')).read().splitlines()
                     if not ln.startswith(('start time', 'end time'))]
            out[tag + '_log'] = np.array('\n'.join(lines))
        out[tag + '_geometry'] = np.array([np.nan if opts['ratio_fixe'] is None else opts['ratio_fixe'],
                                           np.nan if opts['slant_fix'] is None else opts['slant_fix']])
        out[tag + '_disk_sha256'] = np.stack([np.frombuffer(hashlib.sha256(np.ascontiguousarray(d).tobytes()).digest(), np.uint8)
                                              for d in disk_list])
        for name, img in sorted(captured.items()):
            # scan_shift=<s>_<product>.png
            stem = name[len('scan_'):-len('.png')]
            shift_s, product = stem.split('_', 1)
            key = '%s_%s_%s' % (tag, shift_s.replace('shift=', 's'), product)
            out[key + '_sha256'] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(img).tobytes()).digest(), np.uint8)
            out[key + '_shape'] = np.array(img.shape)
            if tag == 'D' and not (shift_s == 'shift=4' or product == 'clahe'):
                continue
            if product in keep[tag] and (tag != 'B' or (product == 'uncontrasted' and shift_s == 'shift=-2')
                                         or (product == 'clahe' and shift_s == 'shift=3')):
                out[key] = img
    shutil.rmtree(tmp)
    save('g14_pipeline', **out)


def g11_crop():
    """single_image_process's crop / pad block (Solex_recon.py:155-171), run unmodified: transversalium and
    FITS off, `image_process` replaced by a recorder so that the (frame, cercle) it is handed is the fixture."""
    rng = np.random.default_rng(11)
    seen = {}

    def record(frame, cercle, options, header, basefich):
        seen['frame'], seen['cercle'] = np.array(frame), tuple(cercle)
        return frame, frame
    ref_sr.image_process = record
    out = {}
    cases = [  # (h, w, cercle, fixed_width, crop_width_square)
        (40, 90, (45.7, 20.0, 18.0), None, True),        # square crop, circle centred
        (40, 90, (12.2, 20.0, 18.0), None, True),        # circle near the left edge: pad on the left (tx > 0)
        (40, 90, (80.9, 20.0, 18.0), None, True),        # circle near the right edge: pad on the right
        (40, 90, (-1, -1, -1), None, True),              # no circle: centre on w // 2
        (40, 90, (45.0, 20.0, 18.0), 140, False),        # wider than the image: pad both sides
        (40, 90, (45.0, 20.0, 18.0), 33, False),         # odd width
        (41, 57, (30.3, 20.0, 18.0), 58, True),          # fixed_width wins over crop_width_square
        (64, 37, (18.0, 30.0, 17.0), None, True),        # taller than wide: square crop pads
        (40, 90, (-1, -1, -1), 21, False),
        (40, 90, (0.4, 20.0, 18.0), 30, False),          # cx = 0
    ]
    for k, (h, w, cercle, fw, sq) in enumerate(cases):
        img = rng.integers(0, 65536, (h, w)).astype(np.uint16)
        opts = dict(QUIET, save_fit=False, transversalium=False, fixed_width=fw, crop_width_square=sq)
        ref_sr.single_image_process(img, {}, opts, cercle, [0, 0, 0, 0], 'x', (0, h))
        out['in_%d' % k] = img
        out['args_%d' % k] = np.array([cercle[0], cercle[1], cercle[2], -1 if fw is None else fw, int(sq)], dtype=np.float64)
        out['out_%d' % k] = seen['frame']
        out['cercle_%d' % k] = np.array(seen['cercle'], dtype=np.float64)
    out['n'] = np.array(len(cases))
    save('g11_crop', **out)


def g12_shift_order():
    """options['shift'] / options['shift_requested'] after the reference's solex_read (Solex_recon.py:53-55) for
    request lists that repeat, contain or omit the two implicit shifts; and the raw disks it returns for them."""
    import hashlib
    import shutil
    install_blur_shim()
    params = dict(n=120, w=160, h=40, bits=8, seed=12, tilt=0.02, curv=1e-4)
    frames = synth.synth_frames_numpy(params['n'], params['w'], params['h'], params['bits'], seed=params['seed'],
                                      tilt=params['tilt'], curv=params['curv'])
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, 'scan.ser')
    synth.write_ser(path, frames)
    cases = [(10, [0]), (10, [10]), (10, [10, 0]), (10, [0, 10]), (10, [-3, 5, 5]), (10, [2, -2, 0, 2]),
             (0, [0, 1]), (5, [5]), (-4, [3, -4, 0]), (10, list(range(-10, 11)))]
    out = {'frames_sha256': np.frombuffer(hashlib.sha256(frames.tobytes()).digest(), np.uint8), 'n': np.array(len(cases))}
    out.update({'param_' + k: np.array(v) for k, v in params.items()})
    for k, (efs, req) in enumerate(cases):
        opts = dict(QUIET, shift=list(req), ellipse_fit_shift=efs, flip_x=False)
        disks, bounds, hdr = ref_sr.solex_read(path, opts)
        out['efs_%d' % k] = np.array(efs)
        out['request_%d' % k] = np.array(req)
        out['shift_%d' % k] = np.array(opts['shift'])
        out['shift_requested_%d' % k] = np.array(opts['shift_requested'])
        out['bounds_%d' % k] = np.array(bounds)
        out['disk_sha256_%d' % k] = np.stack([np.frombuffer(hashlib.sha256(np.ascontiguousarray(d).tobytes()).digest(), np.uint8)
                                              for d in disks])
    shutil.rmtree(tmp)
    save('g12_shift_order', **out)


def g15_stubborn():
    """Reference correct_transversalium2 with `stubborn_transversalium`, unmodified, in shim mode: cv2.filter2D :=
    scipy.ndimage.correlate(mode='mirror') (= BORDER_REFLECT_101, exact correlation accumulated in float64 and
    stored in the source type).  Pins the spurious-row logic, the row replacement, fix_edge_effect and the taper;
    cv2's own DFT rounding stays unpinned.  Two inputs: uint16 (float32 filters) and a float64 image."""
    from scipy import ndimage
    if not REAL:
        cv2.filter2D = lambda src, ddepth, kernel: ndimage.correlate(src, kernel, mode='mirror')
    out = {}
    rng = np.random.default_rng(15)
    h, w = 300, 330
    img = _disk_image(h, w, 15, ratio=1.0)
    gain = 1 + 0.004 * rng.standard_normal(h)
    for y in (97, 98, 160, 221):
        gain[y] *= 1.25                                   # lines the smooth correction cannot follow
    img = np.clip(img * gain[:, None], 1, 65535).astype(np.uint16)
    circle = (w / 2 + 1.5, h / 2 - 0.7, 0.40 * w)
    borders = [20.3, 14.2, 310.8, 287.6]
    opts = dict(QUIET, trans_strength=301, stubborn_transversalium=True)
    out['image'] = img
    out['circle'] = np.array(circle)
    out['borders'] = np.array(borders)
    out['u16_out'] = ref_su.correct_transversalium2(img, circle, borders, opts, 0, '/tmp/x')
    rowf = 1 + 0.1 * np.sin(np.arange(h) / 40.0)
    out['row_factor'] = rowf
    out['f64_out'] = ref_su.correct_transversalium2(img * rowf[:, None], circle, borders, dict(opts), 0, '/tmp/x')
    bb = [0, 30 + 20, w - 1, 270 - 20]                    # backup-bounds call form, Solex_recon.py:146
    out['bb_borders'] = np.array(bb)
    out['bb_out'] = ref_su.correct_transversalium2(img, (0, 0, 99999), bb, dict(opts, trans_strength=41), 0, '/tmp/x')
    save('g15_stubborn', **out)


def g16_ellipse_skimage():
    """An INDEPENDENT implementation of the ellipse fit the reference takes from lsq-ellipse (absent here: restated in
    oracle/limb_oracle.py from Halir & Flusser 1998): scikit-image 0.18.3's skimage.measure.EllipseModel, which implements the
    same published method with its own numerics and its own conversion to centre / axes / angle.  Noisy partial ellipses of the
    shapes a solar limb takes; the fixture holds the points and EllipseModel's (xc, yc, a, b, theta)."""
    from skimage.measure import EllipseModel
    rng = np.random.default_rng(16)
    out = {}
    cases = [(1010.0, 995.0, 840.0, 880.0, 0.03, 0.4, 2 * np.pi, 900), (523.0, 480.7, 400.0, 310.0, -0.6, 0.8, 2 * np.pi, 500),
             (2000.0, 1280.0, 1680.0, 1120.0, 0.9, 1.5, 1.4 * np.pi, 1200), (300.0, 310.0, 250.0, 251.0, 0.2, 0.3, 2 * np.pi, 300),
             (1500.0, 700.0, 500.0, 1300.0, 0.1, 2.0, 1.7 * np.pi, 800), (640.0, 512.0, 420.0, 390.0, -1.2, 0.5, 2 * np.pi, 64)]
    for i, (xc, yc, a, b, theta, noise, arc, n) in enumerate(cases):
        t = np.sort(rng.uniform(0.3, 0.3 + arc, n))
        x = xc + a * np.cos(t) * np.cos(theta) - b * np.sin(t) * np.sin(theta) + noise * rng.standard_normal(n)
        y = yc + a * np.cos(t) * np.sin(theta) + b * np.sin(t) * np.cos(theta) + noise * rng.standard_normal(n)
        pts = np.c_[x, y]
        model = EllipseModel()
        assert model.estimate(pts)
        out['points%d' % i] = pts
        out['params%d' % i] = np.array(model.params, dtype=np.float64)          # xc, yc, a, b, theta
        out['truth%d' % i] = np.array([xc, yc, a, b, theta])
    out['n_cases'] = np.array(len(cases))
    save('g16_ellipse_skimage', **out)


ALL = dict(G16=g16_ellipse_skimage, G6=g6_vignette, G13=g13_limb, G14=g14_pipeline, G1=g1_mean_max, G2=g2_extract, G3=g3_warp, G4=g4_transversalium, G5=g5_rescale,
           G7=g7_matrix, G8=g8_fit_shim, G9=g9_fits, G10=g10_cli, G11=g11_crop, G12=g12_shift_order, G15=g15_stubborn)

if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    todo = sys.argv[1:] or list(ALL)
    for key in todo:
        ALL[key]()
