import sys, io, contextlib, traceback
import numpy as np, torch
sys.path.insert(0, '.')
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, synth, outputs
from solex_ser_recon_en_amd.video_reader import array_reader
from oracle import pipeline_oracle as po

def run(name, frames, extra=None):
    opts = SHG_MAIN.default_options(); opts.update(extra or {}, _nolog=True)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            r = Solex_recon.solex_do_work([(array_reader(torch.from_numpy(frames).cuda()), opts)], True, return_results=True)
        outputs.flush()
        got = 'ok %s' % (tuple(np.asarray(r[0][0][0]).shape),)
    except BaseException as e:
        got = 'raises %s: %s' % (type(e).__name__, str(e)[:90])
    try:
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all='ignore'):
            po.run(frames, extra or {})
        ref = 'ok'
    except BaseException as e:
        ref = 'raises %s: %s' % (type(e).__name__, str(e)[:90])
    print('%-16s product: %-110s | oracle: %s' % (name, got, ref), flush=True)

base = synth.synth_frames_numpy(400, 400, 32, 16, seed=1, tilt=0.01, curv=5e-5)
run('normal', base)
run('zeros', np.zeros_like(base))
run('constant', np.full_like(base, 1000))
run('noise_only', np.random.default_rng(0).integers(0, 3000, base.shape).astype(np.uint16))
run('ten_frames', base[195:205].copy())
run('half_scan', base[:200].copy())
run('saturated', np.full_like(base, 65535))
run('disk_no_line', synth.synth_frames_numpy(400, 400, 32, 16, seed=1, scene=dict(depth=0.0)))
run('tiny_disk', synth.synth_frames_numpy(400, 400, 32, 16, seed=1, scene=dict(ax=20.0, ay=20.0)))
torch.cuda.synchronize(); print('device still alive')
