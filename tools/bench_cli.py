"""Wall clock of the real CLI over a folder of synthetic SER files (decode from the page cache, PCIe upload, the GPU
path, PNG encoders): the end-to-end figure a user sees.  Never bench.py's `value` (that one has the stack in HBM)."""
import contextlib
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, outputs, synth  # noqa: E402


def main():
    files, frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    flags = sys.argv[3] if len(sys.argv) > 3 else '-c'
    mode = os.environ.get('SHG_CLI_MODE', '')
    if mode == 'sync':
        outputs.synchronous = True
    elif mode == 'noop':
        outputs.write_png16 = lambda path, img: None
        from solex_ser_recon_en_amd import solex_util
        if hasattr(solex_util, 'write_png16'):
            solex_util.write_png16 = outputs.write_png16
    tmp = tempfile.mkdtemp(dir=os.environ.get('SHG_BENCH_DIR', '/dev/shm' if os.path.isdir('/dev/shm') else None))
    stack = synth.synth_frames_torch(frames, 2000, 200, 16, seed=0).cpu().numpy()
    paths = []
    for i in range(files):
        paths.append(synth.write_ser(os.path.join(tmp, 'scan%d.ser' % i), stack))
    for label, batch in (('warm-up', paths), ('timed', paths)):       # the first batch pins the staging buffers
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            rc = SHG_MAIN.main([flags] + batch if flags else batch)
        outputs.flush()
        dt = time.perf_counter() - t0
        print('%s: %d file(s) x %d frames, flags %r: %.3f s -> %.0f frames/s, %.1f ms per file (rc %s)' % (
            label, len(batch), frames, flags, dt, len(batch) * frames / dt, dt / len(batch) * 1e3, rc))
    for root, _, names in os.walk(tmp):
        for n in names:
            os.remove(os.path.join(root, n))
    os.rmdir(tmp)


if __name__ == '__main__':          # SHG_PLOT_PROCESSES spawns workers that re-import this module
    main()
