#!/usr/bin/env python3
"""bench.py -- SER frames/s of the whole SHG hot path (frame stack resident in HBM -> CLAHE /
contrast products in HBM), plus the HBM-roofline figure of the dominant kernel and a CPU
baseline (the NumPy oracle of the same path on the host cores).

    python bench.py --gpus 1 --steps 50 --warmup 5      # the defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic scan:
  sum/max over frames -> line detect + cubic fit -> column extraction (raw disks, S = 2:
  the ellipse-fit shift 10 and the requested shift 0) -> limb ellipse fit -> ellipse->circle
  warp -> transversalium -> CLAHE + contrast products.  No file is read or written inside
  the timed region (inputs are resident in HBM; the PNG/FITS encoders are off the path).

N = 1: BASELINE.json configs[1] -- 2000 frames of 2000x200 16-bit, single H-alpha shift,
transversalium + ellipse fit on.  N > 1, default --mode folder (configs[4]'s layout, the reference's own
batch parallelism: SHG_MAIN.handle_folder + Pool over files): one such scan per rank, the files are
independent, so there is no data-path collective (weak scaling).  --mode sharded (configs[2]'s layout):
ONE scan of N x 2000 frames whose frames are sharded over the ranks -- RCCL all-reduce of the integer
sum/max frames, all-gather of the disk columns, mosaic post-processed on rank 0; its serial per-file tail
(limb fit, one requested disk) does not shard, see DESIGN.md section 6.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--frames', type=int, default=2000, help='frames per GPU')
    ap.add_argument('--width', type=int, default=2000)
    ap.add_argument('--height', type=int, default=200)
    ap.add_argument('--bits', type=int, default=16)
    ap.add_argument('--mode', choices=['folder', 'sharded'], default='folder')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-frames', type=int, default=0, help='frames of the CPU-baseline sample (0 = the whole scan)')
    ap.add_argument('--stages', action='store_true', help='print a per-stage wall-clock table to stderr')
    ap.add_argument('--shifts', default='0', help="requested pixel shifts, CLI syntax of -w: '0', 'a,b,c' or 'x:y:w' (C4 = -10:10:1)")
    return ap.parse_args()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as td

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the SHG hot path has no CPU fallback')
    backend = os.environ.get('SHG_DIST_BACKEND', 'nccl')           # 'gloo' lets two ranks share one GPU (functional tests)
    device_index = local_rank % max(torch.cuda.device_count(), 1) if backend != 'nccl' else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == 'nccl':
            td.init_process_group('nccl', device_id=torch.device('cuda', device_index))
        else:
            td.init_process_group(backend)

    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, _lib, synth, timing
    from solex_ser_recon_en_amd.video_reader import array_reader

    from solex_ser_recon_en_amd.CLI_handler import parse_shift
    requested_shifts = parse_shift(args.shifts)
    n_disks = len(dict.fromkeys([10, 0] + requested_shifts))
    sharded = world > 1 and args.mode == 'sharded'
    n_local = args.frames
    n_scan = n_local * world if sharded else n_local
    k0 = rank * n_local if sharded else 0
    t0 = time.time()
    stack = synth.synth_frames_torch(n_scan, args.width, args.height, args.bits, seed=rank if not sharded else 0,
                                     k0=k0, k1=k0 + n_local, n_total=n_scan, padded=True)
    torch.cuda.synchronize()
    log('[rank %d] synthetic stack %s %s built in %.1f s' % (rank, tuple(stack.shape), stack.dtype, time.time() - t0))

    def step():
        opts = SHG_MAIN.default_options()
        opts['_nolog'] = True
        opts['shift'] = list(requested_shifts)
        rdr = array_reader(stack, frame_count=n_scan, frame_range=(k0, k0 + n_local) if sharded else None)
        with contextlib.redirect_stdout(io.StringIO()):
            # the production entry point: sharded = collectives per scan (and the disks of a Doppler stack dealt to
            # the ranks); folder = every rank processes its own scan
            return Solex_recon.solex_do_work([(rdr, opts)], True, distribute='frames' if sharded else 'none',
                                             return_results=True)

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    _lib.profile_reset()
    _lib.profile_enable(True, only=('accumulate', 'extract'))
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t_start
    _lib.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (pass A: sum/max over the stack), live HIP events --------
    acc_ms, acc_n = _lib.profile_get('accumulate')
    ext_ms, ext_n = _lib.profile_get('extract')
    ih, iw = max(args.width, args.height), min(args.width, args.height)
    bpp = args.bits // 8
    bytes_a = n_local * ih * iw * bpp                                   # algorithmic: every sample read once
    n_shifts = n_disks
    distinct = sorted(set(s + d for s in dict.fromkeys([10, 0] + requested_shifts) for d in (0, 1)))
    u = len(distinct)                                                   # distinct samples per row (4 for S=2: c, c+1, c+10, c+11)
    bytes_b = n_local * ih * (u * bpp + 2 * n_shifts)
    ach = bytes_a / (acc_ms / acc_n * 1e-3) / 1e9 if acc_n else 0.0
    traffic = None
    tpath = os.path.join(REPO, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get('%dx%dx%dx%d' % (n_local, args.width, args.height, args.bits))
            traffic = rec['accumulate_bytes_per_launch'] if rec else None
        except Exception:      # noqa: BLE001
            traffic = None
    from solex_ser_recon_en_amd import ops as _ops
    pitch_bytes = _ops.frame_stride(stack) * bpp
    flat = torch.as_strided(stack, (stack.shape[0] * pitch_bytes // bpp,), (1,)) if stack.shape[0] > 1 else stack.reshape(-1)
    ceiling, ceiling_shape = _ops.stream_read_ceiling(flat)
    walk, walk_shape = _ops.stream_read_ceiling(flat, mode=2, vecs_per_frame=pitch_bytes // 16) if pitch_bytes % 16 == 0 else (0.0, None)
    roofline = {'kernel': 'k_accumulate_vec (pass A: sum+max over frames)', 'bound': 'hbm',
                'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                'traffic': traffic, 'algorithmic_bytes_per_launch': bytes_a, 'frame_pitch_bytes': pitch_bytes,
                'measured_read_ceiling': {'value': round(ceiling, 1), 'unit': 'GB/s', 'frac_of_it': round(ach / ceiling, 4) if ceiling else None,
                                          'how': 'best of %d launch shapes of a trivial read-only kernel (shg_stream_read_probe) over the same stack, '
                                                 'blocks x unroll = %s' % (8, ceiling_shape)},
                'frame_walk_ceiling': {'value': round(walk, 1), 'unit': 'GB/s', 'frac_of_it': round(ach / walk, 4) if walk else None,
                                       'how': 'the same XOR-only kernel with pass A\'s addresses (a lane walks the frame axis), best of 8 '
                                              '(splits x unroll) = %s' % (walk_shape,)},
                'avg_launch_ms': round(acc_ms / acc_n, 5) if acc_n else None, 'launches': acc_n,
                'secondary': {'kernel': 'k_extract (pass B)', 'algorithmic_bytes_per_launch': bytes_b,
                              'avg_launch_ms': round(ext_ms / ext_n, 5) if ext_n else None,
                              'achieved': round(bytes_b / (ext_ms / ext_n * 1e-3) / 1e9, 1) if ext_n else None}}

    if args.stages and rank == 0:
        timing.enabled = True
        timing.reset()
        for _ in range(3):
            step()
        timing.enabled = False
        log('per-stage host wall clock (ms / step, each stage fenced by a device sync):')
        for k, v in timing.totals.items():
            log('  %-28s %8.3f' % (k, v / 3 * 1e3))

    # ---- CPU baseline: the NumPy oracle of the same path on this box's host cores -------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU baseline is an N=1 figure
        from oracle import pipeline_oracle as po
        n_cpu = args.cpu_frames or n_local
        if sharded:
            n_cpu = min(n_cpu, n_local)
        sample = _ops.stack_to_host(stack[:n_cpu])
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all='ignore'):
            ref = po.run(sample, {'shift': list(requested_shifts)})
        t_cpu = time.perf_counter() - t0
        cpu = {'value': round(n_cpu / t_cpu, 1), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
               'sample': 'the first %d frames of rank 0\'s stack through the whole path (oracle/pipeline_oracle.py, '
                         'NumPy, single thread like the reference\'s frame loops), %.1f s' % (n_cpu, t_cpu),
               'host_cpus': os.cpu_count()}
        if out and not sharded and n_cpu == n_local and requested_shifts == [0]:
            cc = np.asarray(out[0][0][0])
            want = ref['results'][0]['cc']
            d = np.abs(cc.astype(np.int64) - want.astype(np.int64)) if cc.shape == want.shape else None
            cpu['parity_vs_gpu'] = 'shape mismatch' if d is None else 'max |diff| %d LSB, %d of %d px differ' % (
                d.max(), np.count_nonzero(d), d.size)

    if rank == 0:
        total_frames = n_scan * args.steps if sharded else n_local * world * args.steps
        line = {
            'metric': 'SER frames/sec end-to-end (decode->clahe), stack resident in HBM',
            'value': round(total_frames / elapsed, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u16' if bpp == 2 else 'u8', 'data': 'synthetic',
            'config': {'workload': '%d-frame %d-bit SER, %dx%d frames, %s (S=%d disks), '
                                   'transversalium+ellipse on%s' % (
                                       n_scan, args.bits, args.width, args.height,
                                       'single H-alpha shift' if requested_shifts == [0] else 'shifts -w %s' % args.shifts, n_disks,
                                       '' if world == 1 else (', frames sharded over %d GPUs (RCCL all-reduce + all-gather)' % world
                                                              if sharded else ', folder mode: one scan per GPU, no collective')),
                       'frames_per_gpu': n_local, 'mode': 'single' if world == 1 else args.mode},
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == '__main__':
    main()
