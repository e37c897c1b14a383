import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from oracle import pipeline_oracle as po
from solex_ser_recon_en_amd import synth, SHG_MAIN, Solex_recon, outputs
g = np.load('tests/golden/g14_pipeline.npz')
frames = synth.synth_frames_numpy(int(g['param_n']), int(g['param_w']), int(g['param_h']), int(g['param_bits']), seed=int(g['param_seed']), tilt=float(g['param_tilt']), curv=float(g['param_curv']), row_gain=g['row_gain'])
path='/tmp/scan_fg.ser'; synth.write_ser(path, frames)
for tag, sc in {'F': {'stubborn_transversalium': True, 'trans_strength': 41}, 'G': {'stubborn_transversalium': True, 'de-vignette': True}, 'A': {}, 'D': {'de-vignette': True, 'shift': [0, 4]}}.items():
    opts = SHG_MAIN.default_options(); opts.update(sc, _nolog=True)
    disk_list, bounds, hdr = Solex_recon.solex_read(path, opts)
    want = po.run(frames, sc)
    results = Solex_recon.solex_process(opts, disk_list, bounds, hdr); outputs.flush()
    requested = [s for s in opts['shift'] if s in opts['shift_requested']]
    for shift,(cc,protus) in zip(requested, results):
        ref = want['results'][shift]
        for name,got,exp in (('cc',cc,ref['cc']),('protus',protus,ref['protus'])):
            d=np.abs(np.asarray(got).astype(np.int64)-exp.astype(np.int64))
            print(tag, shift, name, 'max', d.max(), 'count', np.count_nonzero(d), 'of', d.size, 'hist', np.bincount(d.ravel())[:8])
        key='%s_s%d_clahe' % (tag, shift)
        if key in g.files:
            d=np.abs(np.asarray(cc).astype(np.int64)-g[key].astype(np.int64)); print('   vs reference golden', d.max(), np.count_nonzero(d))
