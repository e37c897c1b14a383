"""CPU oracle of the whole per-file flow.  TEST INFRASTRUCTURE ONLY.

Chains the stage oracles in the order of the reference's solex_read (Solex_recon.py:49-83),
solex_process (:93-133) and single_image_process (:136-174), including the geometry state
carried in `options` (ratio_fixe / slant_fix in degrees, :113-121).  No files are written;
every intermediate array is returned.  The shim-mode golden g14_pipeline.npz (reference run
end to end with the unpinned primitives shimmed) pins this chain.
"""
import math

import numpy as np

try:
    from . import limb_oracle as limb
    from . import shg_oracle as orc
except ImportError:
    import limb_oracle as limb
    import shg_oracle as orc

DEFAULTS = dict(shift=[0], ratio_fixe=None, slant_fix=None, disk_display=True, delta_radius=0,
                crop_width_square=False, transversalium=True, trans_strength=301, img_rotate=0,
                flip_x=False, fixed_width=None, ellipse_fit_shift=10, stubborn_transversalium=False,
                **{'de-vignette': False})


def solex_read(frames, options):
    """-> dict(mean, max, fit, y1, y2, shifts, disks)."""
    opts = dict(DEFAULTS, **options)
    shifts = orc.shift_list(opts['ellipse_fit_shift'], opts['shift'])
    mean, mx = orc.compute_mean_max(orc.SerReader(frames))
    fit, y1, y2, p, _ = orc.line_fit(mean, mx)
    disks = orc.extract_columns(orc.SerReader(frames), fit, shifts)
    if opts['flip_x']:
        disks = [np.flip(d, axis=1) for d in disks]
    return dict(mean=mean, max=mx, fit=fit, y1=y1, y2=y2, p=p, shifts=shifts, disks=disks)


def single_image_process(frame, opts, cercle0, borders, backup_bounds):
    out = dict(circular=frame)
    if opts['transversalium']:
        if not cercle0 == (-1, -1, -1):
            circle, borders_t = cercle0, borders
        else:
            circle, borders_t = (0, 0, 99999), [0, backup_bounds[0] + 20, frame.shape[1] - 1, backup_bounds[1] - 20]
        if opts['stubborn_transversalium']:
            det, out['spurious'] = orc.correct_transversalium2_stubborn(frame, circle, borders_t, opts['trans_strength'])
        else:
            det, out['c'] = orc.correct_transversalium2(frame, circle, borders_t, opts['trans_strength'])
    else:
        det = frame
    out['detrans'] = det
    cropped, cercle = orc.crop_center(det, cercle0, opts['fixed_width'], opts['crop_width_square'])
    out['cropped'] = cropped
    out['cercle'] = cercle
    out.update(orc.image_process(cropped, cercle, opts['disk_display'], opts['delta_radius'], opts['img_rotate']))
    return out


def solex_process(read, options):
    """-> dict(shift -> products dict), plus 'geometry'."""
    opts = dict(DEFAULTS, **options)
    requested = list(opts['shift'])
    ratio_fixe, slant_fix = opts['ratio_fixe'], opts['slant_fix']
    borders = [0, 0, 0, 0]
    cercle0 = (-1, -1, -1)
    results = {}
    geometry = None
    for i, shift in enumerate(read['shifts']):
        flag_requested = shift in requested
        disk = read['disks'][i]
        if ratio_fixe is None and slant_fix is None:
            frame, cercle0, ratio_fixe, phi, borders = limb.ellipse_to_circle(disk)
            slant_fix = math.degrees(phi)
            geometry = dict(circle=cercle0, ratio=ratio_fixe, phi=phi, borders=borders)
        else:
            ratio = ratio_fixe if ratio_fixe is not None else 1.0
            phi = math.radians(slant_fix) if slant_fix is not None else 0.0
            if flag_requested:
                frame = orc.correct_image(disk / 65536, phi, ratio, np.array([-1.0, -1.0]), -1.0)[0]
                if opts['de-vignette'] and not cercle0 == (-1, -1, -1):      # Solex_recon.py:124-128
                    frame = orc.remove_vignette(frame, cercle0)
        if not flag_requested:
            continue
        results[shift] = single_image_process(frame, opts, cercle0, borders, (read['y1'], read['y2']))
    return dict(results=results, geometry=geometry)


def run(frames, options):
    read = solex_read(frames, options)
    proc = solex_process(read, options)
    return dict(read=read, **proc)
