"""BASELINE's configurations at full size through the route bench.py times and the CLI takes by default: solex_do_work with
several files in flight -- one shg_scan_file call per file (csrc/scan.hip) made by the native scan pool (csrc/pool.hip), pass A on
the frame-pass lane and launched when the scan is queued (csrc/streams.hip).  tests/test_fullsize_gpu.py holds the same
configurations against the oracle on the stage-by-stage route; here every raw disk and every product of the pooled route is
held bit for bit against that route, and the oracle (oracle/pipeline_oracle.py) checks the pooled route directly.
Reference loop body: Solex_recon.py:33-42 (Pool), :105-133 (the disks of a file)."""
import numpy as np
import pytest

from tests.conftest import flips

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def pkg():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, ops, outputs, synth
    return SHG_MAIN, Solex_recon, ops, outputs, synth


def run_batch(pkg, stacks, extra, pooled, monkeypatch):
    """-> per file: (options, raw disks as arrays, [(cc, protus), ...] as arrays)"""
    SHG_MAIN, Solex_recon, ops, outputs, synth = pkg
    from solex_ser_recon_en_amd.video_reader import array_reader
    monkeypatch.setenv('SHG_SCAN_CALL', '1' if pooled else '0')
    tasks = []
    for st in stacks:
        opts = SHG_MAIN.default_options()
        opts.update(extra, _nolog=True, _keep_raw=True)
        tasks.append((array_reader(st), opts))
    if pooled:
        res = Solex_recon.solex_do_work(tasks, True, distribute='none', return_results=True, workers=4)
        disks = [[np.asarray(d) for d in opts['_raw_disks']] for _, opts in tasks]
    else:                                                   # one file at a time, one C call per stage
        res, disks = [], []
        for rdr, opts in tasks:
            disk_list, bounds, hdr = Solex_recon.solex_read(rdr, opts)
            disks.append([np.asarray(d) for d in disk_list])
            res.append(Solex_recon.solex_process(opts, disk_list, bounds, hdr))
    outputs.flush()
    torch.cuda.synchronize()
    out = []
    for (_, opts), dk, per_file in zip(tasks, disks, res):
        out.append((opts, dk, [(np.asarray(cc), np.asarray(pr)) for cc, pr in per_file]))
    return out


def same_as_stage_route(pooled, staged):
    assert len(pooled) == len(staged)
    for (po_, pd, pr), (so, sd, sr) in zip(pooled, staged):
        for key in ('shift', 'shift_requested', 'ratio_fixe', 'slant_fix'):
            assert po_[key] == so[key], key
        assert len(pd) == len(sd) and len(pr) == len(sr)
        for a, b in zip(pd, sd):
            np.testing.assert_array_equal(a, b)
        for (cc1, p1), (cc2, p2) in zip(pr, sr):
            np.testing.assert_array_equal(cc1, cc2)
            np.testing.assert_array_equal(p1, p2)


def against_the_oracle(what, host_frames, extra, got, probe):
    from oracle import pipeline_oracle as po
    opts, disks, results = got
    with np.errstate(all='ignore'):
        want = po.run(host_frames, dict(extra, shift=list(probe)))
    for shift, ref in zip(want['read']['shifts'], want['read']['disks']):
        np.testing.assert_array_equal(disks[opts['shift'].index(shift)], ref)                  # raw disks: bit exact
    np.testing.assert_allclose(opts['ratio_fixe'], want['geometry']['ratio'], rtol=1e-9)
    requested = [s for s in opts['shift'] if s in opts['shift_requested']]
    for shift in probe:
        cc, protus = results[requested.index(shift)]
        for name, img, ref in (('cc', cc, want['results'][shift]['cc']), ('protus', protus, want['results'][shift]['protus'])):
            flips('%s shift %d %s' % (what, shift, name), img, ref)


def test_c2_through_the_pool(pkg, monkeypatch):
    """BASELINE configs[1] (2000 frames of 2000 x 200, 16 bit, one shift): six different scans in flight in the native pool.
    All six against the stage route bit for bit, the first and the last against the oracle."""
    SHG_MAIN, Solex_recon, ops, outputs, synth = pkg
    stacks = [synth.synth_frames_torch(2000, 2000, 200, 16, seed=20 + i) for i in range(6)]
    pooled = run_batch(pkg, stacks, {}, True, monkeypatch)
    staged = run_batch(pkg, stacks, {}, False, monkeypatch)
    same_as_stage_route(pooled, staged)
    for i in (0, 5):
        against_the_oracle('C2 pooled, scan %d' % i, ops.stack_to_host(stacks[i]), {}, pooled[i], [0])


def test_c4_through_the_pool(pkg, monkeypatch):
    """BASELINE configs[3] (-w -10:10:1: 21 requested disks, one launch per kernel for all 21 disks): six different
    scans in flight.  All 21 disks and 42 products of every scan against the stage route bit for bit; shifts -10, 0, +10 of
    the first scan against the oracle."""
    SHG_MAIN, Solex_recon, ops, outputs, synth = pkg
    extra = {'shift': list(range(-10, 11))}
    stacks = [synth.synth_frames_torch(2000, 2000, 200, 16, seed=30 + i) for i in range(6)]
    pooled = run_batch(pkg, stacks, extra, True, monkeypatch)
    assert all(len(r) == 21 and len(d) == 21 for _, d, r in pooled)
    staged = run_batch(pkg, stacks, extra, False, monkeypatch)
    same_as_stage_route(pooled, staged)
    against_the_oracle('C4 pooled', ops.stack_to_host(stacks[0]), {}, pooled[0], [-10, 0, 10])


def test_c5_files_through_the_pool(pkg, monkeypatch):
    """BASELINE configs[4]'s file shape (2560 x 256, 16 bit), the first 1000 frames of six different files in flight (the
    whole 4000-frame file runs in test_fullsize_gpu.py): stage route bit for bit, the first file against the oracle."""
    SHG_MAIN, Solex_recon, ops, outputs, synth = pkg
    stacks = [synth.synth_frames_torch(1000, 2560, 256, 16, seed=40 + i) for i in range(6)]
    pooled = run_batch(pkg, stacks, {}, True, monkeypatch)
    staged = run_batch(pkg, stacks, {}, False, monkeypatch)
    same_as_stage_route(pooled, staged)
    against_the_oracle('C5 pooled, first 1000 frames', ops.stack_to_host(stacks[0]), {}, pooled[0], [0])


def test_more_disks_than_a_launch_takes(pkg, monkeypatch):
    """-w -14:14:1 on a small scan: 29 requested disks, more than the 24 a launch of the warp and products kernels carries (their
    pointer tables and bounds travel by value) -- the second launch's disks must come out like the first's: pooled route == stage
    route for every raw disk and product, and three shifts against the oracle."""
    SHG_MAIN, Solex_recon, ops, outputs, synth = pkg
    extra = {'shift': list(range(-14, 15))}
    stacks = [synth.synth_frames_torch(600, 640, 64, 16, seed=70 + i) for i in range(2)]
    pooled = run_batch(pkg, stacks, extra, True, monkeypatch)
    assert all(len(r) == 29 for _, _, r in pooled)
    staged = run_batch(pkg, stacks, extra, False, monkeypatch)
    same_as_stage_route(pooled, staged)
    against_the_oracle('29 disks', ops.stack_to_host(stacks[0]), {}, pooled[0], [-14, 0, 14])


def test_whole_c5_files_through_the_pool_against_the_oracle(pkg, monkeypatch):
    """BASELINE configs[4] at its full size: 4000 frames of 2560 x 256, 16 bit -- 5.2 GB a file.  Three different files in flight in
    the native pool; all three against the stage route bit for bit, and the WHOLE first file against the oracle (a minute of NumPy):
    raw disks bit exact, limb geometry, cc and protus."""
    SHG_MAIN, Solex_recon, ops, outputs, synth = pkg
    stacks = [synth.synth_frames_torch(4000, 2560, 256, 16, seed=60 + i) for i in range(3)]
    pooled = run_batch(pkg, stacks, {}, True, monkeypatch)
    staged = run_batch(pkg, stacks, {}, False, monkeypatch)
    same_as_stage_route(pooled, staged)
    assert pooled[0][1][0].shape == (2560, 4000)
    host = ops.stack_to_host(stacks[0])
    del stacks
    torch.cuda.empty_cache()
    against_the_oracle('C5 pooled, whole 4000-frame file', host, {}, pooled[0], [0])


def test_upload_service_with_more_readers_than_chunks(pkg, tmp_path):
    """Many one-chunk files through eight readers: a reader that found the queue non-empty, lost the last chunk to another
    reader and then slept with a copy in flight never reported its chunk (the wait for the file hung).  300 files, each must land."""
    import threading
    SHG_MAIN, Solex_recon, ops, outputs, synth = pkg
    from solex_ser_recon_en_amd.video_reader import video_reader
    frames = synth.synth_frames_numpy(6, 96, 24, 16, seed=1)
    path = str(tmp_path / 'tiny.ser')
    synth.write_ser(path, frames)
    failed = []

    def work():
        try:
            for _ in range(300):
                rdr = video_reader(path)
                rdr.begin_device_stack(chunk_bytes=3 * 96 * 24 * 2, readers=8)     # two chunks, eight readers
                stack, job = rdr._upload
                if not job.done.wait(timeout=20):
                    failed.append('a chunk was never reported')
                    return
                if job.errors:
                    failed.append(repr(job.errors[0]))
                    return
            got = ops.stack_to_host(rdr.device_stack())
            if not np.array_equal(got, frames):
                failed.append('frames differ')
        except BaseException as e:      # noqa: BLE001
            failed.append(repr(e))
    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(120)
    assert not t.is_alive() and not failed, failed


@pytest.mark.parametrize('mode', ['0', '1', 'auto'])
def test_direct_reads_give_the_same_stack(pkg, tmp_path, monkeypatch, mode):
    """SHG_READ_DIRECT: O_DIRECT reads of the enclosing 4 KiB-aligned span into the pinned buffer (the SER header is 178 bytes:
    no frame starts on a block) against buffered reads: the same frames in HBM (g1's decode), also for chunks that end at the
    end of the file, also where the file system refuses O_DIRECT (the buffered read takes over)."""
    import os
    SHG_MAIN, Solex_recon, ops, outputs, synth = pkg
    from solex_ser_recon_en_amd import video_reader as vr
    frames = synth.synth_frames_numpy(37, 200, 50, 16, seed=2)
    for where in (str(tmp_path), '/dev/shm'):
        path = os.path.join(where, 'direct_%s_%d.ser' % (mode, os.getpid()))
        try:
            synth.write_ser(path, frames)
            fd = os.open(path, os.O_RDONLY)
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)          # cold, where the file system can forget it
            os.close(fd)
            monkeypatch.setenv('SHG_READ_DIRECT', mode)
            got = ops.stack_to_host(vr.video_reader(path).device_stack(chunk_bytes=5 * 200 * 50 * 2))
            np.testing.assert_array_equal(got, frames)
            part = ops.stack_to_host(vr.video_reader(path, frame_range=(30, 37)).device_stack(chunk_bytes=4 * 200 * 50 * 2))
            np.testing.assert_array_equal(part, frames[30:37])
        finally:
            if os.path.exists(path):
                os.unlink(path)


def test_seven_scans_in_flight_give_the_serial_products(pkg):
    """Twelve files of different shapes, depths and options through a native pool of seven workers (every kernel of the chain beside
    other scans' kernels with different grids, LDS sizes and argument structs, some scans on the rarer branches) must come out exactly
    as one scan at a time does."""
    SHG_MAIN, Solex_recon, ops, outputs, synth = pkg
    from solex_ser_recon_en_amd.video_reader import array_reader
    specs = [(600, 640, 48, 16, 3, {}), (520, 600, 40, 16, 4, {'shift': [-2, 0, 3]}), (600, 640, 48, 8, 5, {'flip_x': True}),
             (560, 48, 640, 16, 6, {'crop_width_square': True}), (600, 640, 48, 16, 7, {'transversalium': False}),
             (600, 640, 48, 16, 8, {'fixed_width': 500, 'img_rotate': 90}), (640, 700, 56, 16, 9, {'ratio_fixe': 1.0}),
             (600, 640, 48, 16, 10, {}), (600, 640, 48, 16, 11, {'shift': list(range(-4, 5))}), (600, 640, 48, 16, 12, {}),
             (580, 640, 48, 16, 13, {}), (600, 660, 48, 16, 14, {})]
    stacks = [synth.synth_frames_torch(n, w, h, bits, seed=seed) for n, w, h, bits, seed, _ in specs]

    def run(workers):
        tasks = []
        for st, spec in zip(stacks, specs):
            opts = SHG_MAIN.default_options()
            opts.update(spec[5], _nolog=True, _keep_raw=True)
            tasks.append((array_reader(st), opts))
        res = Solex_recon.solex_do_work(tasks, True, distribute='none', return_results=True, workers=workers)
        outputs.flush()
        torch.cuda.synchronize()
        return [([np.asarray(d) for d in o['_raw_disks']], [(np.asarray(cc), np.asarray(pr)) for cc, pr in per], o['ratio_fixe'])
                for (_, o), per in zip(tasks, res)]
    serial = run(1)
    merged = run(7)
    for (d1, r1, q1), (d2, r2, q2) in zip(serial, merged):
        assert q1 == q2 and len(d1) == len(d2) and len(r1) == len(r2)
        for a, b in zip(d1, d2):
            np.testing.assert_array_equal(a, b)
        for (c1, p1), (c2, p2) in zip(r1, r2):
            np.testing.assert_array_equal(c1, c2)
            np.testing.assert_array_equal(p1, p2)


def test_pool_survives_scans_that_fail(pkg):
    """Scans that fail in the middle of their chains (noise only: the line fit; a tiny disk: the limb fit) among good ones in a pool
    of five workers: the batch raises what the serial order raises, and the pool goes on to give the serial products for the next batch."""
    SHG_MAIN, Solex_recon, ops, outputs, synth = pkg
    from solex_ser_recon_en_amd.video_reader import array_reader
    good = [synth.synth_frames_torch(500, 520, 40, 16, seed=50 + i) for i in range(5)]
    noise = torch.from_numpy(np.random.default_rng(0).integers(0, 3000, (400, 32, 400)).astype(np.int16)).cuda().view(torch.uint16)
    tiny = torch.from_numpy(synth.synth_frames_numpy(400, 400, 32, 16, seed=1, scene=dict(ax=20.0, ay=20.0)).view(np.int16)).cuda().view(torch.uint16)

    def run(stacks, workers):
        tasks = []
        for st in stacks:
            opts = SHG_MAIN.default_options()
            opts.update(_nolog=True)
            tasks.append((array_reader(st), opts))
        try:
            res = Solex_recon.solex_do_work(tasks, True, distribute='none', return_results=True, workers=workers)
            err = None
        except Exception as e:      # noqa: BLE001
            res, err = None, e
        outputs.flush()
        torch.cuda.synchronize()
        return res, err
    mixed = [good[0], good[1], noise, good[2], tiny, good[3], good[4]]
    _, want_err = run(mixed, 1)
    assert want_err is not None
    serial, err = run(good, 1)
    assert err is None
    for _ in range(3):
        _, got_err = run(mixed, 5)
        assert type(got_err) is type(want_err) and str(got_err) == str(want_err), (got_err, want_err)
        merged, err = run(good, 5)
        assert err is None
        for a, b in zip(serial, merged):
            for (c1, p1), (c2, p2) in zip(a, b):
                np.testing.assert_array_equal(np.asarray(c1), np.asarray(c2))
                np.testing.assert_array_equal(np.asarray(p1), np.asarray(p2))
