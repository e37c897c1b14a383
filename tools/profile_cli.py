import cProfile, pstats, contextlib, io, os, sys, tempfile
sys.path.insert(0, '.')
from solex_ser_recon_en_amd import SHG_MAIN, outputs, synth
tmp = tempfile.mkdtemp(dir='/dev/shm')
stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=0).cpu().numpy()
paths = [synth.write_ser(os.path.join(tmp, 'scan%d.ser' % i), stack) for i in range(6)]
with contextlib.redirect_stdout(io.StringIO()):
    SHG_MAIN.main(['-c'] + paths[:1]); outputs.flush()
pr = cProfile.Profile(); pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    SHG_MAIN.main(['-c'] + paths); outputs.flush()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
import shutil; shutil.rmtree(tmp)
