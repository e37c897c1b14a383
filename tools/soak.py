"""Soak: many back-to-back scans in one process; device / host memory must stay flat and the product unchanged."""
import contextlib
import hashlib
import io
import os
import resource
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=int(os.environ.get('SHG_STEP_SEED', '1')), padded=True)     # (1: a 2097 px disk, the reflected-border route)
    first = None
    for i in range(steps):
        opts = SHG_MAIN.default_options()
        opts.update(_nolog=True, shift=[0] if i % 50 else [0, 3, -3])
        with contextlib.redirect_stdout(io.StringIO()):
            res = Solex_recon.solex_do_work([(array_reader(stack), opts)], True, return_results=True)
        if i % 500 == 0 or i == steps - 1:
            cc = np.asarray(res[0][0][0]) if i % 50 else np.asarray(res[0][0][0])
            digest = hashlib.sha256(cc.tobytes()).hexdigest()[:12]
            first = first or digest
            print('step %5d  device allocated %7.1f MB reserved %7.1f MB  host maxrss %7.1f MB  cc %s%s' % (
                i, torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6,
                resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3, digest, '' if digest == first else '  <-- CHANGED'), flush=True)


if __name__ == '__main__':
    main()
