"""Per kernel, per configuration: calls per scan, average duration, ALGORITHMIC bytes, fraction of the 8 TB/s HBM peak, and the
PMC traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) next to it -- one tracked text file per round
(profiles/<tag>_roofline_table.txt) instead of figures read off a terminal.

    roofline_table.py <tag> <label> <steps> <trace dir> [<pmc fetch dir> <pmc write dir>] -- N W H bits S k out_w

trace dir: rocprofv3 --kernel-trace --stats of tools/step_loop.py (one scan at a time); the PMC dirs: the same loop under
--pmc FETCH_SIZE and --pmc WRITE_SIZE.  Algorithmic bytes follow SURVEY.md section 8(d): pass A = N*ih*iw*B, pass B =
N*ih*(U*B + 2*S), and one u16 image pass = 2 B per pixel of the image a kernel reads or writes (P = ih * out_w pixels per requested
disk, D = ih * N pixels of a raw disk, Q = D / 16 pixels of the limb stage's quarter-size image).
FETCH_SIZE is shown as counted; on gfx950 it tallies a wide (16 B / lane) coalesced stream at half its bytes
(MI355X_MICROARCH.md, HBM), so `pmc/alg` is given for both readings: (FETCH + WRITE) and (2 * FETCH + WRITE)."""
import collections
import csv
import glob
import os
import re
import sys

PEAK = 8000e9


def newest(d, pattern):
    files = glob.glob(os.path.join(d, '**', pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def algorithmic(name, c):
    """bytes per SCAN of kernel `name` (all its launches of one scan together), or None for a control-plane kernel that moves KBs."""
    N, ih, iw, B, S, k, P, D, Q, U = (c[x] for x in ('N', 'ih', 'iw', 'B', 'S', 'k', 'P', 'D', 'Q', 'U'))
    table = [
        ('k_accumulate', N * ih * iw * B),
        ('k_finalize_rot', ih * iw * (2 * 6 + 4)),               # two slabs of (u32 sum + u16 max) in, mean + max out
        ('k_reduce_partials', ih * iw * (2 * 6 + 10)),
        ('k_blur_reduce', ih * iw * 2),
        ('k_extract', N * ih * (U * B + 2 * S)),
        ('k_limb_blur', D * 2 + 2 * Q * 4),
        ('k_limb_select', Q * 4 * 2), ('k_limb_flood', Q * 4 * 2), ('k_limb_canny_tile', Q * 4 + Q * 5), ('k_limb_border_merge', Q * 5),
        ('k_limb_emit', Q * 5),
        ('k_warp_rows', (k + c['fit_image']) * (D * 2 + P * 2)),
        ('k_rowpair_stats', k * P * 2),
        ('k_scale_rows', k * P * 4), ('k_crop_pad', k * P * 4), ('k_frame_hist', k * (D * 2 + P * 2)),
        ('k_tile_hist16', k * P * 2),
        ('k_hist_reduce_sat', None),                             # a few MB of clamped slice counters: no algorithmic figure
        ('k_hist_reduce', k * ((P + 65534) // 65535) * 131072 * 1),
        ('k_clahe_interp', k * P * 4), ('k_select16_pass', k * P * 2), ('k_products', k * P * 10),
    ]
    for prefix, b in table:
        if name.startswith(prefix):
            return b
    return None


def main():
    argv = sys.argv[1:]
    cut = argv.index('--')
    tag, label, steps, trace = argv[0], argv[1], int(argv[2]), argv[3]
    pmc_dirs = argv[4:cut]
    N, W, H, bits, S, k, out_w = (int(v) for v in argv[cut + 1:cut + 8])
    ih, iw, B = max(W, H), min(W, H), bits // 8
    U = 4 if S == 2 else S + 1
    c = dict(N=N, ih=ih, iw=iw, B=B, S=S, k=k, P=ih * out_w, D=ih * N, Q=((ih + 3) // 4) * ((N + 3) // 4), U=U, fit_image=0)
    stats = newest(trace, '*kernel_stats.csv')
    rows = []
    for r in csv.DictReader(open(stats)):
        m = re.search(r'\bk_[a-z0-9_]+(<[^>]*>)?', r['Name'])
        if not m or 'anonymous' not in r['Name'] or 'at::native' in r['Name'] or int(r['Calls']) < steps:
            continue
        rows.append((m.group(0), int(r['Calls']) / steps, float(r['AverageNs']), float(r['TotalDurationNs']) / steps))
    pmc = {}
    for d, which in zip(pmc_dirs, ('FETCH', 'WRITE')):
        f = newest(d, '*counter_collection.csv')
        if not f:
            continue
        acc, launches = collections.defaultdict(float), collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            m = re.search(r'\bk_[a-z0-9_]+(<[^>]*>)?', r['Kernel_Name'])
            if m and 'anonymous' in r['Kernel_Name']:
                acc[m.group(0)] += float(r['Counter_Value']) * 1024.0
                launches[m.group(0)] += 1
        for name, total in acc.items():                  # bytes per launch
            pmc.setdefault(name, {})[which] = total / launches[name]
    out = ['# %s  %s: N=%d %dx%d %d-bit, S=%d disks extracted, k=%d requested, corrected images %d x %d' % (tag, label, N, W, H, bits, S, k, ih, out_w),
           '# %s' % os.path.relpath(stats, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
           '%-34s %5s %9s %9s %10s %7s %6s %9s %9s %8s %8s' % ('kernel', 'calls', 'avg us', 'us/scan', 'alg MB', 'TB/s', 'frac', 'FETCH MB', 'WRITE MB',
                                                               'pmc/alg', '2F+W/alg')]
    total = 0.0
    for name, calls, avg_ns, per_scan_ns in sorted(rows, key=lambda r: -r[3]):
        total += per_scan_ns
        alg = algorithmic(name, c)
        p = pmc.get(name, {})
        f, w = (p.get('FETCH'), p.get('WRITE'))
        fetch_scan = f * calls if f is not None else None
        write_scan = w * calls if w is not None else None
        cols = ['%-34s' % name[:34], '%5.1f' % calls, '%9.1f' % (avg_ns / 1e3), '%9.1f' % (per_scan_ns / 1e3)]
        if alg:
            tbs = alg / (per_scan_ns * 1e-9)
            cols += ['%10.2f' % (alg / 1e6), '%7.2f' % (tbs / 1e12), '%6.3f' % (tbs / PEAK)]
        else:
            cols += ['%10s' % '-', '%7s' % '-', '%6s' % '-']
        cols += ['%9.2f' % (fetch_scan / 1e6) if fetch_scan is not None else '%9s' % '-',
                 '%9.2f' % (write_scan / 1e6) if write_scan is not None else '%9s' % '-']
        if alg and fetch_scan is not None and write_scan is not None:
            cols += ['%8.2f' % ((fetch_scan + write_scan) / alg), '%8.2f' % ((2 * fetch_scan + write_scan) / alg)]
        else:
            cols += ['%8s' % '-', '%8s' % '-']
        out.append(' '.join(cols))
    out.append('library kernels per scan: %.1f us in %.0f launches' % (total / 1e3, sum(r[1] for r in rows)))
    text = '\n'.join(out) + '\n'
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', '%s_roofline_table_%s.txt' % (tag, label))
    open(dst, 'w').write(text)
    sys.stdout.write(text)


if __name__ == '__main__':
    main()
