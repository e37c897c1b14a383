#!/usr/bin/env python3
"""bench.py -- SER frames/s of the SHG hot path on MI355X, with the HBM-roofline figure of the dominant
kernel, a decode-inclusive figure and a CPU baseline, as ONE JSON line on stdout (rank 0).

    python bench.py --gpus 1 --steps 50 --warmup 5      # the defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic scan whose frame stack is resident in HBM:
  sum/max over frames -> line detect + cubic fit -> column extraction (raw disks, S = 2: the ellipse-fit
  shift 10 and the requested shift 0) -> limb ellipse fit -> ellipse->circle warp -> transversalium ->
  CLAHE + contrast products (left in HBM).  The K timed steps are K independent scans handed to
  solex_do_work in one call, the way a folder of files is: the reference overlaps the post-processing of up to
  four files (Pool(4), Solex_recon.py:30-42), here `--workers` scan threads (default 4) each drive their own HIP
  stream, so one file's host control plane overlaps the others' kernels.  No file is read or written inside
  that timed region; `value` is frames / wall time of the K steps (max over ranks).

Beside `value` the line carries:
  roofline      pass A (k_accumulate_vec), algorithmic bytes / average launch duration from HIP events recorded on
                the launch streams during the timed region; `uncontended` repeats it from a serial pass (one scan
                at a time, nothing else on the device), which is the figure rocprofv3 of `--workers 1` agrees with
  kernel_ms_per_step / gpu_busy_frac   sum of the event-bracketed durations of every library entry point of one scan
                (serial pass) and its ratio to the timed wall clock per step
  e2e           the same scans read from a SER file in /dev/shm: file -> pinned host -> HBM -> products, frames/s
                and achieved host->device GB/s (PCIe-inclusive; never the headline `value`)
  sharded_c3    BASELINE configs[2]: ONE 4000-frame 2000x200 scan, frames sharded over the ranks, decode-inclusive
                (each rank reads its own byte range of the file), RCCL all-reduce of the integer sum / max frames and
                all-reduce of the zero-filled disk mosaic; strong scaling
  cpu_baseline  the NumPy oracle of the same path on this box's host cores (rank 0, N = 1 only)

N = 1: BASELINE.json configs[1] (2000 frames of 2000x200 16-bit, single H-alpha shift, transversalium + ellipse
fit on).  N > 1: the headline is folder mode (configs[4]'s layout, the reference's own batch parallelism,
SHG_MAIN.handle_folder + Pool over files): K such scans per rank, no data-path collective, weak scaling;
`--mode sharded` makes the HBM-resident sharded scan the headline instead (strong scaling).
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
REFERENCE_CPU_FPS = 310.0      # the reference's own two frame loops on a C2 file, 1 core, read from disk


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
    ap.add_argument('--workers', type=int, default=0, help='scans in flight per process (0 = SHG_WORKERS or 4)')
    ap.add_argument('--repeats', type=int, default=5, help='timed regions of --steps scans each; the median one is quoted')
    ap.add_argument('--stacks', type=int, default=0, help='distinct resident stacks the timed scans cycle over (0 = workers + 1)')
    ap.add_argument('--no-extra', action='store_true', help='skip the C4 (21 disks) and one-C5-file legs')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true', help='skip the decode-inclusive legs (e2e, sharded_c3)')
    ap.add_argument('--e2e-files', type=int, default=8)
    ap.add_argument('--c3-scans', type=int, default=4)
    ap.add_argument('--c3-frames', type=int, default=4000, help='frames of the sharded_c3 scan (BASELINE configs[2]: 4000)')
    ap.add_argument('--cpu-frames', type=int, default=0, help='frames of the CPU-baseline sample (0 = the whole scan)')
    ap.add_argument('--stages', action='store_true', help='print a per-stage wall-clock table to stderr')
    ap.add_argument('--shifts', default='0', help="requested pixel shifts, CLI syntax of -w: '0', 'a,b,c' or 'x:y:w' (C4 = -10:10:1)")
    return ap.parse_args()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def scratch_dir(need_bytes=0):
    """/dev/shm (page cache, no disk in the path) when it has room for the synthetic files, else /tmp."""
    import shutil
    for d in ('/dev/shm', '/tmp'):
        try:
            if os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > need_bytes * 1.25 + (64 << 20):
                return d
        except OSError:
            continue
    raise OSError('no scratch directory with %.1f GB free for the synthetic SER files' % (need_bytes / 1e9))


def shared_path(name, need_bytes, rank, world):
    """Rank 0 picks the scratch directory; every rank learns the path -- or the reason there is none, and raises together."""
    import torch.distributed as td
    box = [None]
    if rank == 0:
        try:
            box = [(os.path.join(scratch_dir(need_bytes), name), None)]
        except OSError as e:
            box = [(None, str(e))]
    if world > 1:
        td.broadcast_object_list(box, src=0)
    path, err = box[0]
    if err:
        raise OSError(err)
    return path


def guarded(leg, *a):
    """A decode-inclusive leg must not take the headline down with it (no room for its files, ...): report the reason."""
    try:
        return leg(*a)
    except Exception as e:      # noqa: BLE001
        log('bench.py: %s skipped: %r' % (leg.__name__, e))
        return {'error': repr(e)}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks here, one child process
    per GPU with the environment torch.distributed.run would give it, and pass rank 0's stdout (the one JSON line) through.  This
    process never touches the GPU -- the children are started before anything here could -- and it does not exec: it waits for them
    and exits with the first non-zero code (the other ranks are then ended by their own PIDs)."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:                              # (MASTER_PORT is still set, for code that reads it; the ranks meet through the file)
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    meet_dir = tempfile.mkdtemp(prefix='shg_bench_')
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SHG_BENCH_SELF_LAUNCHED='1',
                   SHG_BENCH_RENDEZVOUS=os.path.join(meet_dir, 'rendezvous'))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = list(procs)
    try:
        while pending:
            for pr in list(pending):
                code = pr.poll()
                if code is None:
                    continue
                pending.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in pending:                    # a rank that fails leaves the others in a collective for good
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in pending:                                   # (this process is being stopped: its ranks go with it, by their own PIDs)
            pr.terminate()
        import shutil
        shutil.rmtree(meet_dir, ignore_errors=True)
    return rc


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    import numpy as np
    import torch
    import torch.distributed as td

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the SHG hot path has no CPU fallback')
    backend = os.environ.get('SHG_DIST_BACKEND', 'nccl')           # 'gloo' lets two ranks share one GPU (functional tests)
    device_index = local_rank % max(torch.cuda.device_count(), 1) if backend != 'nccl' else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        # ranks this file started itself meet through a FILE (launch_ranks made it): no port that another process could take
        # between its choice and the ranks' bind; under a launcher the environment's MASTER_ADDR / MASTER_PORT are the launcher's
        meet = os.environ.get('SHG_BENCH_RENDEZVOUS')
        kw = dict(init_method='file://' + meet, rank=rank, world_size=world) if meet else {}
        if backend == 'nccl':
            td.init_process_group('nccl', device_id=torch.device('cuda', device_index), **kw)
        else:
            td.init_process_group(backend, **kw)

    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, _lib, synth, timing
    from solex_ser_recon_en_amd import ops as _ops
    from solex_ser_recon_en_amd.CLI_handler import parse_shift
    from solex_ser_recon_en_amd.video_reader import array_reader

    requested_shifts = parse_shift(args.shifts)
    n_disks = len(dict.fromkeys([10, 0] + requested_shifts))
    sharded = world > 1 and args.mode == 'sharded'
    workers = args.workers or Solex_recon._worker_count(None, 1 << 30)
    n_local = args.frames
    n_scan = n_local * world if sharded else n_local
    k0 = rank * n_local if sharded else 0
    # A folder holds a different stack per file: the scans in flight must not all read the same 1.6 GB (what one scan pulls
    # through the Infinity Cache another would find there), so the timed scans cycle over workers + 1 resident stacks.
    n_stacks = 1 if sharded else max(1, args.stacks or workers + 1)
    t0 = time.time()
    stacks = [synth.synth_frames_torch(n_scan, args.width, args.height, args.bits, seed=(rank * 64 + j) if not sharded else 0,
                                       k0=k0, k1=k0 + n_local, n_total=n_scan, padded=True) for j in range(n_stacks)]
    stack = stacks[0]
    torch.cuda.synchronize()
    log('[rank %d] %d synthetic stack(s) %s %s built in %.1f s' % (rank, n_stacks, tuple(stack.shape), stack.dtype, time.time() - t0))

    def options(shifts=None):
        opts = SHG_MAIN.default_options()
        opts['_nolog'] = True
        opts['shift'] = list(requested_shifts if shifts is None else shifts)
        return opts

    def run_scans(n, n_workers, results=False, shifts=None, pool=None, first=0):
        """n scans of resident stacks through the production entry point, scan i on stack (first + i) mod len(pool).
        sharded: collectives per scan (and the disks of a Doppler stack dealt to the ranks), one scan at a time; otherwise the
        scans are independent files."""
        pool = stacks if pool is None else pool
        n_total = n_scan if pool is stacks else pool[0].shape[0]
        tasks = [(array_reader(pool[(first + i) % len(pool)], frame_count=n_total,
                               frame_range=(k0, k0 + n_local) if sharded else None), options(shifts)) for i in range(n)]
        with contextlib.redirect_stdout(io.StringIO()):
            if sharded:                                  # one call: rank 0's post-processing of scan k overlaps everybody's read of scan k + 1
                return Solex_recon.solex_do_work(tasks, True, distribute='frames', return_results=results)
            return Solex_recon.solex_do_work(tasks, True, distribute='none', return_results=results, workers=n_workers)

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    def timed_regions(steps, repeats, **kw):
        """`repeats` timed regions of exactly `steps` scans each, every one bracketed by barrier + synchronize on both sides;
        -> seconds per region, the maximum over ranks of each."""
        times = []
        for r in range(repeats):
            barrier()
            t_start = time.perf_counter()
            run_scans(steps, workers, first=r * steps, **kw)
            barrier()
            times.append(time.perf_counter() - t_start)
        if world > 1:
            t = torch.tensor(times, dtype=torch.float64, device='cuda')
            td.all_reduce(t, op=td.ReduceOp.MAX)
            times = [float(v) for v in t.tolist()]
        return times

    if args.warmup > 0:
        run_scans(args.warmup, workers)
    # What the interpreter holds now (torch, the package, the stack) stays: exempt it from the cyclic collector, as timeit
    # does by switching it off.  Otherwise about one batch in a hundred meets a full collection of those ~10^6 objects --
    # 85 ms with every scan worker stopped (tools/scan_timeline.py with GC=cb), five times a 20-scan timed region.
    import gc
    gc.collect()
    gc.freeze()
    _lib.profile_reset()
    _lib.profile_enable(True, only=('accumulate', 'extract'))       # pass A (on the lane) and pass B (on the scans' own streams)
    region_s = timed_regions(args.steps, max(1, args.repeats))
    _lib.profile_enable(False)
    lane = lane_timeline(_lib, len(region_s))
    elapsed = sorted(region_s)[len(region_s) // 2]                       # the median region is the one quoted

    # ---- roofline of the dominant kernel (pass A: sum/max over the stack), live HIP events --------
    acc_ms, acc_n = _lib.profile_get('accumulate')
    ext_ms, ext_n = _lib.profile_get('extract')
    ih, iw = max(args.width, args.height), min(args.width, args.height)
    bpp = args.bits // 8
    bytes_a = n_local * ih * iw * bpp                                   # algorithmic: every sample read once
    distinct = sorted(set(s + d for s in dict.fromkeys([10, 0] + requested_shifts) for d in (0, 1)))
    u = len(distinct)                                                   # distinct samples per row (4 for S=2: c, c+1, c+10, c+11)
    bytes_b = n_local * ih * (u * bpp + 2 * n_disks)
    ach = bytes_a / (acc_ms / acc_n * 1e-3) / 1e9 if acc_n else 0.0

    # serial pass: one scan at a time, every entry point bracketed by events -> uncontended kernel durations
    serial_steps = max(3, min(10, args.steps))
    run_scans(2, 1, results=True)           # the first scans that hand their images back allocate host arrays and pinned areas: not timed
    torch.cuda.synchronize()
    _lib.profile_reset()
    _lib.profile_enable(True)
    barrier()
    t_serial = time.perf_counter()
    out = []
    for i in range(serial_steps):           # one call per scan: nothing of the next scan (its pass A, launched when it is queued) runs beside this one
        out += run_scans(1, 1, results=True, first=i)
    torch.cuda.synchronize()
    t_serial = time.perf_counter() - t_serial
    _lib.profile_enable(False)
    acc1_ms, acc1_n = _lib.profile_get('accumulate')
    ext1_ms, ext1_n = _lib.profile_get('extract')
    all_ms, all_n = _lib.profile_total()
    _lib.profile_reset()
    kernel_ms_per_step = all_ms / serial_steps
    ach1 = bytes_a / (acc1_ms / acc1_n * 1e-3) / 1e9 if acc1_n else 0.0
    # post-processing bytes of one requested disk: warp 2 + 2, row-pair statistics 2, row scaling 2 + 2,
    # CLAHE histograms 2, CLAHE blend 2 + 2, percentile select 2, contrast products 2 x 2 + 3 x 2 = 28 bytes per output pixel
    out_px = sum(int(np.prod(cc.shape)) for cc, _ in out[-1]) if out and out[-1] else 0
    bytes_post = 28 * out_px

    traffic, traffic_from = None, None
    tpath = os.path.join(REPO, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get('%dx%dx%dx%d' % (n_local, args.width, args.height, args.bits))
            if rec:
                traffic = rec['accumulate_bytes_per_launch']
                traffic_from = 'profiles/traffic.json: %s' % rec.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run '
                                                                    '(PMC counters cannot be read from inside the process)')
        except Exception:      # noqa: BLE001
            traffic = None
    pitch_bytes = _ops.frame_stride(stack) * bpp
    flat = torch.as_strided(stack, (stack.shape[0] * pitch_bytes // bpp,), (1,)) if stack.shape[0] > 1 else stack.reshape(-1)
    ceiling, ceiling_shape = _ops.stream_read_ceiling(flat)
    walk, walk_shape = _ops.stream_read_ceiling(flat, mode=2, vecs_per_frame=pitch_bytes // 16) if pitch_bytes % 16 == 0 else (0.0, None)
    roofline = {'kernel': 'k_accumulate_vec (pass A: sum+max over frames)', 'bound': 'hbm',
                'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                'frac_uncontended': round(ach1 / HBM_PEAK_GBS, 4),      # the kernel by itself (= rocprofv3 of bench.py --workers 1)
                'traffic': traffic, 'traffic_from': traffic_from,
                'algorithmic_bytes_per_launch': bytes_a, 'frame_pitch_bytes': pitch_bytes,
                'avg_launch_ms': round(acc_ms / acc_n, 5) if acc_n else None, 'launches': acc_n,
                'how': 'HIP events around the launch, on the stream it is launched on, over the %d timed regions; with %d scans in '
                       'flight pass A of every scan goes through one stream (the frame-pass lane, csrc/streams.hip): the passes run '
                       'one after the other, each beside the small kernels of the other scans' % (len(region_s), workers),
                'uncontended': {'achieved': round(ach1, 1), 'frac': round(ach1 / HBM_PEAK_GBS, 4),
                                'avg_launch_ms': round(acc1_ms / acc1_n, 5) if acc1_n else None, 'launches': acc1_n,
                                'how': 'serial pass after the timed region: one scan at a time, same events'},
                'measured_read_ceiling': {'value': round(ceiling, 1), 'unit': 'GB/s', 'frac_of_it': round(ach1 / ceiling, 4) if ceiling else None,
                                          'how': 'best of %d launch shapes of a trivial read-only kernel (shg_stream_read_probe) over the same stack, '
                                                 'blocks x unroll = %s; frac_of_it is of the uncontended figure' % (8, ceiling_shape)},
                'frame_walk_ceiling': {'value': round(walk, 1), 'unit': 'GB/s', 'frac_of_it': round(ach1 / walk, 4) if walk else None,
                                       'how': 'the same XOR-only kernel with pass A\'s addresses (a lane walks the frame axis), best of 8 '
                                              '(splits x unroll) = %s' % (walk_shape,)},
                'lane': lane,
                'secondary': {'kernel': 'k_extract_band (pass B)', 'algorithmic_bytes_per_launch': bytes_b,
                              'avg_launch_ms': round(ext_ms / ext_n, 5) if ext_n else None,
                              'achieved': round(bytes_b / (ext_ms / ext_n * 1e-3) / 1e9, 1) if ext_n else None,
                              'uncontended_avg_launch_ms': round(ext1_ms / ext1_n, 5) if ext1_n else None,
                              'uncontended_achieved': round(bytes_b / (ext1_ms / ext1_n * 1e-3) / 1e9, 1) if ext1_n else None}}

    if args.stages and rank == 0:
        timing.enabled = True
        timing.reset()
        run_scans(3, 1)
        timing.enabled = False
        log('per-stage host wall clock (ms / step, each stage fenced by a device sync):')
        for k, v in timing.totals.items():
            log('  %-28s %8.3f' % (k, v / 3 * 1e3))

    # ---- BASELINE configs[3] (Doppler stack, 21 disks) and one file of configs[4] (4000 x 2560x256), stack resident -----
    c4, c5 = None, None
    if not sharded and not args.no_extra and requested_shifts == [0] and (args.frames, args.width, args.height, args.bits) == (2000, 2000, 200, 16):
        c4 = guarded(extra_leg, 'C4: 2000-frame 16-bit SER 2000x200, -w -10:10:1 (21 disks), stack resident', stacks, parse_shift('-10:10:1'),
                     16, 2, workers, run_scans, barrier, _lib, world)          # (16 scans a region: four rounds of the four scans in flight)
        del stacks[1:]                                                   # the legs below use rank 0's first stack only
        torch.cuda.empty_cache()

        def c5_file():
            pool = [synth.synth_frames_torch(4000, 2560, 256, 16, seed=rank * 64 + j, padded=True) for j in range(2)]
            torch.cuda.synchronize()
            try:
                return extra_leg('one file of C5: 4000-frame 16-bit SER 2560x256, single shift, stack resident', pool, [0], 6, 2, workers,
                                 run_scans, barrier, _lib, world)
            finally:
                del pool
                torch.cuda.empty_cache()
        c5 = guarded(c5_file)

    # ---- decode-inclusive legs: SER files in /dev/shm -> pinned host -> HBM -> products ---------------
    e2e, c3 = None, None
    if not args.no_e2e and not sharded:
        e2e = guarded(e2e_leg, args, world, rank, stack, n_local, options, workers)
        c3 = guarded(sharded_c3_leg, args, world, rank, options, backend)

    # ---- CPU baseline: the NumPy oracle of the same path on this box's host cores -------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU baseline is an N=1 figure
        from oracle import pipeline_oracle as po
        n_cpu = args.cpu_frames or n_local
        sample = _ops.stack_to_host(stack[:n_cpu])
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all='ignore'):
            ref = po.run(sample, {'shift': list(requested_shifts)})
        t_cpu = time.perf_counter() - t0
        cpu = {'value': round(n_cpu / t_cpu, 1), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
               'sample': 'the first %d frames of rank 0\'s stack through the whole path (oracle/pipeline_oracle.py, NumPy, single '
                         'thread like the reference\'s frame loops), array already in memory, %.1f s' % (n_cpu, t_cpu),
               'host_cpus': os.cpu_count(),
               'reference_measured': {'value': REFERENCE_CPU_FPS, 'unit': 'frames/s', 'cores': 1,
                                      'what': 'the reference\'s own compute_mean_max + read_video_improved on a C2 file, which it reads '
                                              'from disk twice, 25 frames at a time (SURVEY.md section 6, measured in the build container, '
                                              'not on this box: the reference cannot travel); the oracle works on an in-memory array, hence '
                                              'its ~5x higher figure'}}
        if out and n_cpu == n_local and requested_shifts == [0]:
            cc = np.asarray(out[0][0][0])                                   # the serial pass's first scan read stacks[0], like the oracle
            want = ref['results'][0]['cc']
            d = np.abs(cc.astype(np.int64) - want.astype(np.int64)) if cc.shape == want.shape else None
            cpu['parity_vs_gpu'] = 'shape mismatch' if d is None else 'max |diff| %d LSB, %d of %d px differ' % (
                d.max(), np.count_nonzero(d), d.size)
            # ... and the second scan, whose circularised disk is 2097 px wide: a width CLAHE's 2 x 2 grid does not divide (reflected
            # border in the tile histograms) -- the shape four of the five synthetic scans have
            if len(out) > 1 and n_stacks > 1 and out[1]:
                second = synth.synth_frames_torch(n_scan, args.width, args.height, args.bits, seed=rank * 64 + 1, k0=k0, k1=k0 + n_local,
                                                  n_total=n_scan, padded=True)         # (stacks[1] again: the legs above freed it)
                with contextlib.redirect_stdout(io.StringIO()), np.errstate(all='ignore'):
                    ref1 = po.run(_ops.stack_to_host(second[:n_cpu]), {'shift': list(requested_shifts)})
                del second
                cc1, want1 = np.asarray(out[1][0][0]), ref1['results'][0]['cc']
                d1 = np.abs(cc1.astype(np.int64) - want1.astype(np.int64)) if cc1.shape == want1.shape else None
                cpu['parity_vs_gpu_second_scan'] = 'shape mismatch' if d1 is None else '%d px wide: max |diff| %d LSB, %d of %d px differ' % (
                    cc1.shape[1], d1.max(), np.count_nonzero(d1), d1.size)

    parity_one_rank = None
    if sharded:
        if n_local % 250 == 0:              # (the generator seeds its noise per 250-frame chunk: a rank's block is the whole scan's frames only then)
            parity_one_rank = sharded_parity(
                lambda: [(array_reader(stack, frame_count=n_scan, frame_range=(k0, k0 + n_local)), options()) for _ in range(2)],    # (two: the series route, the one timed)
                lambda: (array_reader(synth.synth_frames_torch(n_scan, args.width, args.height, args.bits, seed=0, padded=True)), options()), world)
        else:
            parity_one_rank = {'skipped': '--frames must be a multiple of 250 for the synthetic blocks to add up to one scan'}

    if rank == 0:
        total_frames = n_scan * args.steps if sharded else n_local * world * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            'metric': 'SER frames/sec end-to-end (decode\u2192clahe) + %HBM roofline, 1/2/4/8 MI355X',
            'value_basis': 'hbm_resident',
            'metric_note': '`value` / `ms_per_step`: the whole hot path (mean/max -> line fit -> extraction -> limb fit -> warp -> '
                           'transversalium -> CLAHE + contrast products) with the frame stack resident in HBM when the timed region starts '
                           '(the bench contract: inputs resident, the PCIe-inclusive rate is never `value`).  The rate from the FILE -- '
                           'file -> pinned host -> PCIe -> HBM -> products, what "end-to-end (decode->clahe)" names for a user -- is '
                           '`value_decode_inclusive` (= e2e.value), bound by the host link: e2e.frac_of_ceiling is its share of a bare '
                           'pinned copy on this box.  The sharded scan (configs[2]) is sharded_c3.value.',
            'value': round(total_frames / elapsed, 1), 'unit': 'frames/s',
            'value_decode_inclusive': e2e.get('value') if isinstance(e2e, dict) else None,
            'n_gpus': world, 'ranks_seen': td.get_world_size() if world > 1 else 1, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'strong' if sharded else 'weak', 'vs_baseline': None, 'dtype': 'u16' if bpp == 2 else 'u8', 'data': 'synthetic',
            'config': {'workload': '%d-frame %d-bit SER, %dx%d frames, %s (S=%d disks), '
                                   'transversalium+ellipse on%s' % (
                                       n_scan, args.bits, args.width, args.height,
                                       'single H-alpha shift' if requested_shifts == [0] else 'shifts -w %s' % args.shifts, n_disks,
                                       '' if world == 1 else (', frames sharded over %d GPUs (RCCL all-reduce of sum / max frames and of the disk mosaic)' % world
                                                              if sharded else ', folder mode: %d scans per GPU, no collective' % args.steps)),
                       'frames_per_gpu': n_local, 'mode': 'single' if world == 1 else args.mode,
                       'scans_in_flight_per_process': 1 if sharded else min(workers, args.steps),
                       'backend': td.get_backend() if world > 1 else None, 'world_size': world,
                       'collectives_per_scan': 2 if sharded else 0,
                       'collectives': 'all_gather of the packed frame statistics (u32 sums | u16 maxima | failure word) after pass A, reduce SUM of the '
                                      'disk mosaic to the scan\'s owner (scan k -> rank k mod G, which post-processes it) after pass B; two scans being read '
                                      'at a time, one failure word at the end of the series' if sharded else None},
            'parity_vs_one_rank': parity_one_rank,
            'repeats': {'n': len(region_s), 'ms_per_step': [round(t / args.steps * 1e3, 4) for t in region_s],
                        'min': round(min(region_s) / args.steps * 1e3, 4), 'median': round(ms_per_step, 4),
                        'max': round(max(region_s) / args.steps * 1e3, 4),
                        'how': '%d timed regions of exactly %d scans, each bracketed by barrier + synchronize; value / ms_per_step are the '
                               'median region; the scans cycle over %d distinct resident stacks' % (len(region_s), args.steps, n_stacks)},
            'whole_step': {'bytes': bytes_a + bytes_b + bytes_post, 'pass_a': bytes_a, 'pass_b': bytes_b, 'post': bytes_post,
                           'achieved': round((bytes_a + bytes_b + bytes_post) / (ms_per_step * 1e-3) / 1e9, 1), 'unit': 'GB/s',
                           'frac': round((bytes_a + bytes_b + bytes_post) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           'how': 'algorithmic bytes of a whole scan (pass A + pass B + 28 B per output pixel of post-processing, '
                                  'DESIGN.md section 3) / ms_per_step / %.0f GB/s' % HBM_PEAK_GBS},
            'kernel_ms_per_step': round(kernel_ms_per_step, 4),
            'gpu_busy_frac': round(kernel_ms_per_step / ms_per_step, 4),
            'kernel_time_how': 'sum of the HIP-event-bracketed durations of all %d library entry points of one scan, serial pass '
                               '(%.3f ms wall per scan there); gpu_busy_frac = that / ms_per_step' % (
                                   all_n // serial_steps, t_serial / serial_steps * 1e3),
            'roofline': roofline, 'cpu_baseline': cpu, 'e2e': e2e, 'sharded_c3': c3, 'c4': c4, 'c5_file': c5,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


def sharded_parity(sharded_tasks, whole_task, world):
    """One frame-sharded scan against the same scan on ONE rank: `sharded_tasks()` -> the task list every rank hands to
    solex_do_work(distribute='frames') (scan 0's owner is rank 0), `whole_task()` -> rank 0's task over all the frames.  The
    reductions are integer, so every product must be identical."""
    import numpy as np
    import torch
    import torch.distributed as td
    from solex_ser_recon_en_amd import Solex_recon
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            got = Solex_recon.solex_do_work(sharded_tasks(), True, distribute='frames', return_results=True)
            torch.cuda.synchronize()
            out = None
            if td.get_rank() == 0:
                want = Solex_recon.solex_do_work([whole_task()], True, distribute='none', return_results=True, workers=1)
                torch.cuda.synchronize()
                images = differ = 0
                for (a1, b1), (a2, b2) in zip(got[0], want[0]):
                    for a, b in ((a1, a2), (b1, b2)):
                        images += 1
                        a, b = np.asarray(a), np.asarray(b)
                        differ += int(a.shape != b.shape or not np.array_equal(a, b))
                out = {'images_compared': images, 'images_that_differ': differ + (len(got[0]) != len(want[0])),
                       'what': '(cc, protus) of one scan sharded over %d ranks vs the same frames on rank 0 alone' % world}
        return out
    except Exception as e:      # noqa: BLE001
        return {'error': repr(e)}
    finally:
        td.barrier()            # (whatever happened on this rank: the others are waiting here, and the next collective must line up)


def lane_timeline(_lib, n_regions):
    """The frame-pass lane over the timed regions, from the same HIP events as roofline.achieved (shg_profile_dump): the idle gap
    between consecutive passes and how busy the lane is from a region's first pass to its last.  The step is the lane's period
    (DESIGN.md section 5): pass A beside the other scans' kernels + this gap + a region's fill and drain."""
    import tempfile
    path = os.path.join(tempfile.gettempdir(), 'shg_bench_lane_%d.csv' % os.getpid())
    try:
        _lib.check(_lib.lib.shg_profile_dump(path.encode()), 'shg_profile_dump')
        spans = sorted((float(r[2]), float(r[3])) for r in (ln.split(',') for ln in open(path).read().splitlines()[1:]) if r[0] == 'accumulate')
    except Exception as e:      # noqa: BLE001
        return {'error': repr(e)}
    finally:
        if os.path.exists(path):
            os.remove(path)
    if len(spans) < 2:
        return None
    gaps = sorted(spans[i + 1][0] - spans[i][1] for i in range(len(spans) - 1))
    inner = gaps[:len(gaps) - (n_regions - 1)] if n_regions > 1 else gaps        # (the n_regions - 1 longest gaps are the region boundaries)
    busy = sum(b - a for a, b in spans)
    span = (spans[-1][1] - spans[0][0]) - sum(gaps[len(inner):])
    return {'passes': len(spans), 'gap_ms_median': round(inner[len(inner) // 2], 4), 'gap_ms_mean': round(sum(inner) / len(inner), 4),
            'busy_frac': round(busy / span, 4) if span > 0 else None,
            'how': 'idle time between consecutive pass A launches on the lane and the share of the time from a region\'s first pass '
                   'to its last that a pass was running (region boundaries taken out)'}


def extra_leg(what, pool, shifts, steps, warmup, workers, run_scans, barrier, _lib, world):
    """Another BASELINE configuration on resident stacks: `steps` scans over `pool` (after `warmup`), timed like the headline;
    then a serial pass with every entry point bracketed by events for the kernel time per scan and the two frame passes."""
    import torch
    import torch.distributed as td
    n, h, w = pool[0].shape
    bpp = pool[0].element_size()
    # every slot of the pool (workers + 2 scans in flight) sizes its arenas and leases its pinned areas on its first scans of a new
    # shape: two rounds through all of them before anything is timed
    run_scans(max(warmup, 2 * (workers + 2), steps), workers, shifts=shifts, pool=pool)      # (and a whole region's worth: the feeder's buffers for that many tasks)
    import gc
    gc.collect()                            # (what the warm-up left behind goes now, not as a full collection inside the first region)
    gc.freeze()
    torch.cuda.synchronize()
    times = []
    for r in range(3):
        barrier()
        t0 = time.perf_counter()
        run_scans(steps, workers, shifts=shifts, pool=pool, first=r * steps)
        barrier()
        times.append(time.perf_counter() - t0)
    if world > 1:
        t = torch.tensor(times, dtype=torch.float64, device='cuda')
        td.all_reduce(t, op=td.ReduceOp.MAX)
        times = [float(v) for v in t.tolist()]
    dt = sorted(times)[1]
    for i in range(2):                      # the one-worker pool's first scans of this shape allocate their arenas: not timed
        run_scans(1, 1, shifts=shifts, pool=pool, first=i)
    torch.cuda.synchronize()
    _lib.profile_reset()
    _lib.profile_enable(True)
    for i in range(3):                      # one call per scan, as in the main serial pass
        run_scans(1, 1, shifts=shifts, pool=pool, first=i)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    acc_ms, acc_n = _lib.profile_get('accumulate')
    ext_ms, ext_n = _lib.profile_get('extract')
    all_ms, _ = _lib.profile_total()
    _lib.profile_reset()
    parity = route_parity(pool, shifts, workers)
    s_all = len(dict.fromkeys([10, 0] + list(shifts)))
    u = len(set(s + d for s in dict.fromkeys([10, 0] + list(shifts)) for d in (0, 1)))
    ih = max(h, w)
    bytes_a, bytes_b = n * h * w * bpp, n * ih * (u * bpp + 2 * s_all)
    return {'workload': what, 'value': round(n * steps * world / dt, 1), 'unit': 'frames/s', 'ms_per_step': round(dt / steps * 1e3, 3),
            'steps': steps, 'regions_ms_per_step': [round(t / steps * 1e3, 3) for t in times], 'disks_per_scan': len(shifts),
            'parity_vs_stage_route': parity,
            'kernel_ms_per_step': round(all_ms / 3, 4),
            'kernel_time_how': 'event-bracketed entry points of one scan, one scan at a time; an entry point that launches once per 24 disks '
                               'has its host work between the launches inside the bracket -- the kernels alone are in '
                               'profiles/*_step_kernel_table_c4.txt (rocprofv3)',
            'pass_a': {'avg_launch_ms': round(acc_ms / acc_n, 5) if acc_n else None, 'algorithmic_bytes': bytes_a,
                       'frac_uncontended': round(bytes_a / (acc_ms / acc_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if acc_n else None},
            'pass_b': {'avg_launch_us': round(ext_ms / ext_n * 1e3, 2) if ext_n else None, 'algorithmic_bytes': bytes_b,
                       'frac_uncontended': round(bytes_b / (ext_ms / ext_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ext_n else None}}


def route_parity(pool, shifts, workers):
    """The route timed above -- shg_scan_file through the native scan pool, `workers` scans in flight, pass A on the lane and
    launched ahead -- against the stage-by-stage route (SHG_SCAN_CALL=0, one scan at a time), which tests/test_fullsize_gpu.py
    holds against the oracle at these sizes: every raw disk and every product of workers + 2 scans, bit for bit."""
    import numpy as np
    import torch
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon
    from solex_ser_recon_en_amd.video_reader import array_reader
    n = workers + 2

    def tasks():
        out = []
        for i in range(n):
            opts = SHG_MAIN.default_options()
            opts.update(_nolog=True, _keep_raw=True, shift=list(shifts))
            out.append((array_reader(pool[i % len(pool)]), opts))
        return out
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            pooled = tasks()
            res_p = Solex_recon.solex_do_work(pooled, True, distribute='none', return_results=True, workers=workers)
            torch.cuda.synchronize()
            staged = tasks()
            res_s = []
            previous = os.environ.get('SHG_SCAN_CALL')
            os.environ['SHG_SCAN_CALL'] = '0'
            try:
                for rdr, opts in staged:
                    disk_list, bounds, hdr = Solex_recon.solex_read(rdr, opts)
                    opts['_raw_disks'] = disk_list
                    res_s.append(Solex_recon.solex_process(opts, disk_list, bounds, hdr))
            finally:
                if previous is None:
                    del os.environ['SHG_SCAN_CALL']
                else:
                    os.environ['SHG_SCAN_CALL'] = previous
            torch.cuda.synchronize()
        images = differ = 0
        for (_, po_), rp, (_, so), rs in zip(pooled, res_p, staged, res_s):
            pairs = list(zip(po_['_raw_disks'], so['_raw_disks']))
            for (a1, b1), (a2, b2) in zip(rp, rs):
                pairs += [(a1, a2), (b1, b2)]
            if len(rp) != len(rs) or len(po_['_raw_disks']) != len(so['_raw_disks']) or po_['ratio_fixe'] != so['ratio_fixe']:
                differ += 1
            for a, b in pairs:
                images += 1
                a, b = np.asarray(a), np.asarray(b)
                if a.shape != b.shape or not np.array_equal(a, b):
                    differ += 1
        return {'scans': n, 'images_compared': images, 'images_that_differ': differ,
                'what': 'raw disks + (cc, protus) of every requested disk, native pool with %d scans in flight vs the stage route' % workers}
    except Exception as e:      # noqa: BLE001
        return {'error': repr(e)}


def _write_near_gpu(write):
    """tmpfs pages live on the NUMA node of the cpu that first touches them.  A writer that happens to run on the far socket leaves
    the whole file there and every read of the decode leg crosses the socket link (measured on the 2 x 64-core box: 35 GB/s into
    the GPU instead of 52).  The file stands in for one the page cache holds next to its reader, so it is written from the
    cpus the decode readers run on."""
    from solex_ser_recon_en_amd import device
    import torch
    old = device.bind_thread('io', torch.device('cuda', torch.cuda.current_device()))
    try:
        write()
    finally:
        if old is not None:
            os.sched_setaffinity(0, old)


def _write_scan(path, n, width, height, bits, rank, world):
    """Rank 0 writes the synthetic SER file (generated on its GPU), everyone waits for it."""
    import torch
    import torch.distributed as td
    from solex_ser_recon_en_amd import ops, synth
    if rank == 0:
        frames = synth.synth_frames_torch(n, width, height, bits, seed=0)
        host = ops.stack_to_host(frames)
        del frames
        _write_near_gpu(lambda: synth.write_ser(path, host))
        del host
    if world > 1:
        td.barrier()


def h2d_ceiling(n_bytes):
    """What a bare pinned-host -> device copy of a file's worth of bytes reaches on THIS box (the figure e2e is up against): the
    same byte count from one pinned buffer, in one piece and in 64 MB pieces from two streams, HIP events around 4 copies each;
    the better of the two."""
    import torch
    host = torch.empty(n_bytes, dtype=torch.uint8).pin_memory()
    host.fill_(7)
    dst = torch.empty(n_bytes, dtype=torch.uint8, device='cuda')
    cur = torch.cuda.current_stream()
    s2 = torch.cuda.Stream()
    piece = 64 << 20

    def one():
        dst.copy_(host, non_blocking=True)

    def two_streams():
        s2.wait_stream(cur)
        for i, o in enumerate(range(0, n_bytes, piece)):
            with torch.cuda.stream(s2 if i & 1 else cur):
                dst[o:o + piece].copy_(host[o:o + piece], non_blocking=True)
        cur.wait_stream(s2)
    best = 0.0
    for fn in (one, two_streams):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = max(best, n_bytes * 4 / (a.elapsed_time(b) * 1e-3) / 1e9)
    del host, dst
    torch.cuda.empty_cache()
    return best


def e2e_leg(args, world, rank, stack, n_local, options, workers):
    """Folder of identical C2 files through solex_do_work: decode (8 reader threads, pinned buffers, async 2-D hipMemcpy) of
    file k+1.. overlaps the scans in flight.  One file on the node, written by rank 0 and read by every rank (each rank
    reads it n_files times from the page cache and uploads over its own PCIe link)."""
    import torch
    import torch.distributed as td
    from solex_ser_recon_en_amd import Solex_recon, ops, synth
    frame_bytes = stack.shape[1] * stack.shape[2] * stack.element_size()
    path = shared_path('shg_bench_e2e.ser', stack.shape[0] * frame_bytes, rank, world)
    try:
        if rank == 0:
            _write_near_gpu(lambda: synth.write_ser(path, ops.stack_to_host(stack)))
        if world > 1:
            td.barrier()
        size = os.path.getsize(path)
        n_files = max(2, args.e2e_files)

        def go(n):
            with contextlib.redirect_stdout(io.StringIO()):
                Solex_recon.solex_do_work([(path, options()) for _ in range(n)], True, distribute='none', workers=workers)
            torch.cuda.synchronize()
        go(2)
        times = []
        for _ in range(3):                                  # three regions of n_files files each; the median one is quoted
            if world > 1:
                td.barrier()
            t0 = time.perf_counter()
            go(n_files)
            times.append(time.perf_counter() - t0)
        if world > 1:
            t = torch.tensor(times, dtype=torch.float64, device='cuda')
            td.all_reduce(t, op=td.ReduceOp.MAX)
            times = [float(v) for v in t.tolist()]
        dt = sorted(times)[1]
        rate = size * n_files / dt / 1e9
        ceiling = h2d_ceiling(size) if rank == 0 else None          # (alone on the link: after the timed regions, rank 0 only)
        if world > 1:
            td.barrier()
        return {'value': round(n_local * n_files * world / dt, 1), 'unit': 'frames/s', 'ms_per_file': round(dt / n_files * 1e3, 2),
                'files_per_gpu': n_files, 'regions': 3, 'regions_ms_per_file': [round(t / n_files * 1e3, 2) for t in times],
                'file_bytes': size, 'host_to_device_GBps_per_gpu': round(rate, 2),
                'pcie_peak_GBps': 63.0,
                'h2d_ceiling_GBps': round(ceiling, 2) if ceiling else None,
                'frac_of_ceiling': round(rate / ceiling, 4) if ceiling else None,
                'h2d_ceiling_how': 'a bare asynchronous copy of the same %d bytes from one pinned buffer to the device on this box, best of '
                                   '(one piece, 64 MB pieces over two streams), HIP events around 4 copies' % size,
                'what': 'SER file in %s -> pread into pinned host buffers -> asynchronous hipMemcpy2D -> the same hot path, products '
                        'left in HBM (no PNG / FITS encode); decode of the next files overlaps the scans in flight' % os.path.dirname(path)}
    finally:
        if world > 1:
            td.barrier()
        if rank == 0 and path and os.path.exists(path):
            os.remove(path)


def sharded_c3_leg(args, world, rank, options, backend):
    """BASELINE configs[2]: one 4000-frame 2000x200 16-bit scan, frames sharded over the ranks, decode-inclusive."""
    import torch
    import torch.distributed as td
    from solex_ser_recon_en_amd import Solex_recon, dist
    n, w, h = args.c3_frames, 2000, 200
    path = shared_path('shg_bench_c3.ser', n * w * h * 2, rank, world)
    try:
        _write_scan(path, n, w, h, 16, rank, world)

        def go(k):
            with contextlib.redirect_stdout(io.StringIO()):
                Solex_recon.solex_do_work([(path, options()) for _ in range(k)], True, distribute='frames' if world > 1 else 'none')
            torch.cuda.synchronize()
        go(args.c3_scans)                               # (as many as are timed: the allocator then holds a block for every stack in flight)
        if world > 1:
            td.barrier()
        t0 = time.perf_counter()
        go(args.c3_scans)
        if world > 1:
            td.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device='cuda')
            td.all_reduce(t, op=td.ReduceOp.MAX)
            dt = float(t.item())
        parity = sharded_parity(lambda: [(path, options()) for _ in range(2)], lambda: (path, options()), world) if world > 1 else None
        return {'value': round(n * args.c3_scans / dt, 1), 'unit': 'frames/s', 'ms_per_scan': round(dt / args.c3_scans * 1e3, 2),
                'scans': args.c3_scans, 'scaling': 'strong', 'world_size': world, 'backend': td.get_backend() if world > 1 else None,
                'collectives_per_scan': 2 if world > 1 else 0,
                'collectives': 'all_gather (packed frame statistics: u32 sums | u16 maxima | failure word), reduce SUM (disk mosaic, disjoint column blocks) to the scan\'s owner' if world > 1 else None,
                'frames_per_rank': dist.frame_block(n, rank, world)[1] - dist.frame_block(n, rank, world)[0],
                'parity_vs_one_rank': parity,
                'what': 'one %d-frame %dx%d 16-bit SER in %s, every rank decodes its own frame block (file -> pinned -> HBM), one all-gather '
                        'after pass A, reduce of the zero-filled disk mosaic after pass B to the scan\'s owner (scan k -> rank k mod G), who '
                        'post-processes it while all ranks read the next scan' % (n, w, h, os.path.dirname(path))}
    finally:
        if world > 1:
            td.barrier()
        if rank == 0 and path and os.path.exists(path):
            os.remove(path)


if __name__ == '__main__':
    main()
