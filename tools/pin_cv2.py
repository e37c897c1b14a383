#!/usr/bin/env python3
"""Pin the primitives this repo could only restate: cv2.blur, cv2.createCLAHE, cv2.circle, cv2.filter2D (opencv-python) and
LsqEllipse (lsq-ellipse) -- the reference's call sites solex_util.py:166, 230, 532-533, 547 and ellipse_to_circle.py:57-59.

The build image has neither package, so the fixtures g8 / g12 / g13 / g14 / g15 were captured with this repo's own restatements
of those primitives injected into the reference ("shim mode", oracle/capture_goldens.py): for them "HIP == oracle == golden"
says nothing about OpenCV itself.  Whoever has an interpreter WITH both packages (and a checkout of the reference) closes
that loop with one command:

    python3 tools/pin_cv2.py --reference /path/to/Solex_ser_recon_EN [--out pin_report]

It re-runs the capture with the REAL primitives (SHG_PIN_REAL=1: nothing stubbed, nothing shimmed), records the versions, and
compares every array of the regenerated fixtures with the committed shim-mode ones: identical arrays pin the restatement;
any difference is listed with its size (largest absolute difference, number of differing elements) -- the place to look
is then the primitive's restatement in oracle/shg_oracle.py / oracle/limb_oracle.py (and its HIP counterpart).
Exit status 0: everything identical; 1: differences (report written); 2: the packages or the reference are missing.
Needs NumPy, SciPy, scikit-image and astropy like the reference; no GPU."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = {'G8': 'g8_fit_shim', 'G12': 'g12_shift_order', 'G13': 'g13_limb', 'G14': 'g14_pipeline', 'G15': 'g15_stubborn'}


def versions():
    out = {'python': sys.version.split()[0]}
    for name, mod in (('opencv-python', 'cv2'), ('lsq-ellipse', 'ellipse'), ('numpy', 'numpy'), ('scipy', 'scipy'),
                      ('scikit-image', 'skimage'), ('astropy', 'astropy')):
        try:
            m = __import__(mod)
            out[name] = getattr(m, '__version__', 'present')
        except Exception as e:      # noqa: BLE001
            out[name] = 'MISSING (%s)' % type(e).__name__
    try:
        import cv2
        info = cv2.getBuildInformation()
        out['opencv_cpu_baseline'] = next((ln.strip() for ln in info.splitlines() if 'Baseline' in ln), '')
        out['opencv_cpu_dispatch'] = next((ln.strip() for ln in info.splitlines() if 'Dispatched' in ln), '')
    except Exception:      # noqa: BLE001
        pass
    return out


def compare(new_dir, old_dir):
    import numpy as np
    report, same = {}, True
    for key, stem in FIXTURES.items():
        new_path, old_path = os.path.join(new_dir, stem + '.npz'), os.path.join(old_dir, stem + '.npz')
        if not os.path.exists(new_path):
            report[stem] = 'not regenerated'
            same = False
            continue
        new, old = np.load(new_path, allow_pickle=True), np.load(old_path, allow_pickle=True)
        entry = {}
        for name in sorted(set(new.files) | set(old.files)):
            if name not in new.files or name not in old.files:
                entry[name] = 'only in the %s fixture' % ('regenerated' if name in new.files else 'committed')
                same = False
                continue
            a, b = new[name], old[name]
            if a.shape != b.shape or a.dtype != b.dtype:
                entry[name] = 'shape / dtype %s %s against %s %s' % (a.shape, a.dtype, b.shape, b.dtype)
                same = False
            elif a.dtype.kind in 'OUS':
                if not (a == b).all():
                    entry[name] = 'text differs'
                    same = False
            elif not np.array_equal(a, b, equal_nan=a.dtype.kind == 'f'):
                d = np.abs(a.astype(np.float64) - b.astype(np.float64))
                entry[name] = {'max_abs_diff': float(np.nanmax(d)), 'differing': int(np.count_nonzero(d > 0)), 'of': int(d.size)}
                same = False
        report[stem] = entry or 'identical'
    return report, same


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--reference', required=True, help='checkout of thelondonsmiths/Solex_ser_recon_EN')
    ap.add_argument('--out', default='pin_report', help='folder for the regenerated fixtures and pin_report.json')
    args = ap.parse_args()
    v = versions()
    missing = [k for k in ('opencv-python', 'lsq-ellipse', 'numpy', 'scipy', 'scikit-image') if str(v.get(k, '')).startswith('MISSING')]
    if missing:
        print('pin_cv2: this interpreter lacks %s -- run it where the reference itself runs (pip install opencv-python lsq-ellipse)'
              % ', '.join(missing))
        return 2
    if not os.path.exists(os.path.join(args.reference, 'Solex_recon.py')):
        print('pin_cv2: %s is not a checkout of the reference (no Solex_recon.py)' % args.reference)
        return 2
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, SHG_PIN_REAL='1', SHG_GOLDEN_OUT=out, SHG_REFERENCE=os.path.abspath(args.reference), MPLBACKEND='Agg',
               PYTHONDONTWRITEBYTECODE='1')
    with tempfile.TemporaryDirectory() as cwd:
        subprocess.run([sys.executable, os.path.join(REPO, 'oracle', 'capture_goldens.py')] + list(FIXTURES), check=True, env=env, cwd=cwd)
    report, same = compare(out, os.path.join(REPO, 'tests', 'golden'))
    with open(os.path.join(out, 'pin_report.json'), 'w') as f:
        json.dump({'versions': v, 'fixtures': report, 'all_identical': same}, f, indent=1, default=str)
    print(json.dumps({'versions': v, 'fixtures': report}, indent=1, default=str))
    print('pin_cv2: %s' % ('every fixture is IDENTICAL with the real primitives: the restatements are pinned for these versions'
                           if same else 'DIFFERENCES -- see %s' % os.path.join(out, 'pin_report.json')))
    return 0 if same else 1


if __name__ == '__main__':
    sys.exit(main())
