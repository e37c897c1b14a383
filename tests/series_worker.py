"""Run under torch.distributed.run (SHG_DIST_BACKEND=gloo: both ranks on one GPU; default: nccl, one GPU each): a SERIES of
frame-sharded scans through solex_do_work(distribute='frames', return_results=True).  Every rank stores which entries of the
returned list it holds, their cc images, and the collectives it issued.  Usage: series_worker.py <out_dir> <scan.ser> ..."""
import contextlib
import io
import os
import sys

import numpy as np
import torch
import torch.distributed as td

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_dir, files = sys.argv[1], sys.argv[2:]
    rank, local = int(os.environ['RANK']), int(os.environ['LOCAL_RANK'])
    backend = os.environ.get('SHG_DIST_BACKEND', 'nccl')
    torch.cuda.set_device(local if backend == 'nccl' else 0)
    if backend == 'nccl':
        td.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        td.init_process_group(backend)
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, dist

    def options():
        o = SHG_MAIN.default_options()
        o.update(_nolog=True)
        return o
    before = dist.counters['collectives']
    with contextlib.redirect_stdout(io.StringIO()):
        res = Solex_recon.solex_do_work([(f, options()) for f in files], True, distribute='frames', return_results=True)
    torch.cuda.synchronize()
    report = {'n_entries': len(res), 'held': np.array([i for i, r in enumerate(res) if r is not None], dtype=np.int64),
              'collectives': dist.counters['collectives'] - before}
    for i, r in enumerate(res):
        if r is not None:
            (cc, protus), = r
            report['cc_%d' % i] = np.asarray(cc)
            report['protus_%d' % i] = np.asarray(protus)
    np.savez(os.path.join(out_dir, 'series_rank%d.npz' % rank), **report)
    td.barrier()
    td.destroy_process_group()


if __name__ == '__main__':
    main()
