"""Run under torch.distributed.run with the default backend (nccl = RCCL): one process per GPU.

Checks what the gloo runs on one GPU cannot: every rank decodes into ITS device (the decoder thread binds itself to the
caller's GPU), the RCCL branches of dist.exchange_frame_stats / dist.gather_columns run on device buffers, and the products of
a sharded scan / a folder of scans equal the single-GPU run.  Usage: rccl_worker.py <out_dir> <scan.ser> [<scan2.ser> ...]"""
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
    torch.cuda.set_device(local)
    td.init_process_group('nccl', device_id=torch.device('cuda', local))
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, dist
    from solex_ser_recon_en_amd.video_reader import video_reader
    assert dist.active() and td.get_backend() == 'nccl'

    def options(**kw):
        o = SHG_MAIN.default_options()
        o.update(_nolog=True, **kw)
        return o

    report = {}
    # 1. one scan, frames sharded over the ranks (all-reduce sum / max, all-reduce of the mosaic), mosaic on rank 0
    with contextlib.redirect_stdout(io.StringIO()):
        res = Solex_recon.solex_do_work([(files[0], options())], True, return_results=True)
    if rank == 0:
        (cc, protus), = res[0]
        assert cc.t.device.index == local
        report['sharded_cc'] = np.asarray(cc)
    else:
        assert res == [None]                                  # one entry per task: this rank holds none of the scan's products
    # 2. a Doppler stack: disks dealt to the ranks after the limb fit on rank 0
    with contextlib.redirect_stdout(io.StringIO()):
        res = Solex_recon.solex_do_work([(files[0], options(shift=[-2, 0, 3]))], True, return_results=True)
    report['doppler_n'] = len(res[0]) if res and res[0] else 0
    for i, (cc, protus) in enumerate(res[0] if res and res[0] else []):
        assert cc.t.device.index == local
        report['doppler_cc_%d' % i] = np.asarray(cc)
    # 3. folder mode: file i -> rank i mod G, no collective; the decoder thread must land on this rank's GPU
    rdr = video_reader(files[rank % len(files)])
    assert rdr.device_stack().device.index == local
    tasks = [(f, options()) for f in files]
    with contextlib.redirect_stdout(io.StringIO()):
        res = Solex_recon.solex_do_work(tasks, True, return_results=True)
    for i, per_file in enumerate(res):
        cc, protus = per_file[0]
        assert cc.t.device.index == local, 'rank %d produced its folder-mode products on cuda:%d' % (rank, cc.t.device.index)
        report['folder_cc_%d' % i] = np.asarray(cc)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), **report)
    td.barrier()
    td.destroy_process_group()


if __name__ == '__main__':
    main()
