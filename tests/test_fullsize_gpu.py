"""BASELINE.json's full-size configurations through size-independent properties (the oracle only
runs where it finishes in seconds).  All on the GPU."""
import numpy as np
import pytest

from tests.conftest import flips

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def env():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from solex_ser_recon_en_amd import ops, synth
    return ops, synth


def chunked_sum_max(stack, chunk=100):
    total = torch.zeros(stack.shape[1:], dtype=torch.int64, device=stack.device)
    mx = torch.zeros(stack.shape[1:], dtype=torch.int32, device=stack.device)
    for k in range(0, stack.shape[0], chunk):
        blk = stack[k:k + chunk].to(torch.int32)
        total += blk.sum(0, dtype=torch.int64)
        mx = torch.maximum(mx, blk.amax(0))
    return total.ravel(), mx.ravel()


def integer_fit(ih, iw, amplitude=6.0):
    """A curve with zero fractional part: the bilinear weights are exactly (1, 0), so a disk is a pure gather."""
    curve = np.floor(iw / 2 + amplitude * np.sin(np.arange(ih) / 97.0))
    return np.stack([curve, np.zeros(ih), np.arange(ih, dtype=float), curve], axis=1)


def gather_reference(stack, curve, shift, rotated):
    """disk[y, k] = img_k[y, curve[y] + shift] straight from the file layout (torch indexing as an independent reference)."""
    n, h, w = stack.shape
    ih, iw = (w, h) if rotated else (h, w)
    x = torch.as_tensor(np.clip(curve + shift, 0, iw - 2).astype(np.int64), device=stack.device)
    y = torch.arange(ih, device=stack.device)
    s32 = stack.view(torch.int16) if stack.dtype == torch.uint16 else stack
    if rotated:      # img[y, x] = raw[x, W-1-y]
        out = s32[:, x, w - 1 - y]
    else:
        out = s32[:, y, x]
    return out.t().contiguous()


@pytest.mark.parametrize('n,w,h,bits,shifts', [
    (2000, 2000, 200, 16, [10, 0]),                                   # C2
    (2000, 2000, 200, 16, [10, 0] + [s for s in range(-10, 11) if s not in (10, 0)]),   # C4: -w -10:10:1 -> 21 disks
    (1000, 2560, 256, 16, [10, 0]),                                   # C5 frame shape
    (4000, 2560, 256, 16, [10, 0]),                                   # C5: one whole file (5.2 GB stack)
    (4000, 2000, 200, 16, [10, 0]),                                   # C3: the whole 4000-frame scan on one GPU
    (2000, 200, 2000, 16, [10, 0]),                                   # un-rotated file
    (200, 120, 800, 8, [10, 0]),                                      # C1 (8-bit, Width < Height)
    (200, 800, 120, 8, [10, 0]),                                      # C1 shape stored rotated
])
def test_frame_passes_at_full_size(env, n, w, h, bits, shifts):
    ops, synth = env
    stack = synth.synth_frames_torch(n, w, h, bits, seed=3)
    rotated = w > h
    ih, iw = max(w, h), min(w, h)
    # pass A against an independent chunked torch reduction (exact integers)
    total, mx = ops.accumulate_sum_max(stack)
    want_total, want_max = chunked_sum_max(stack)
    assert torch.equal(total, want_total)
    assert torch.equal(mx.view(torch.int16).to(torch.int32) & 0xffff, want_max)
    mean, mxo = ops.finalize_mean_max(total, mx, n, h, w, bits // 8)
    scale = 256 if bits == 8 else 1
    want_mean = (want_total * scale // n).reshape(h, w)
    if rotated:
        want_mean = torch.rot90(want_mean, 1, dims=(0, 1))
    assert torch.equal(mean.view(torch.int16).to(torch.int64) & 0xffff, want_mean)
    # pass B: integer curve => pure gather, for every shift (incl. the clamped ones)
    fit = integer_fit(ih, iw)
    from solex_ser_recon_en_amd.hostmath import column_plan
    ind_l, lw, rw = column_plan(fit, shifts, ih, iw)
    disks = ops.extract_columns(stack, ind_l, lw, rw)
    assert disks.shape == (len(shifts), ih, n)
    for i, s in enumerate(shifts):
        ref = gather_reference(stack, fit[:, 0], s, rotated)
        got = disks[i].view(torch.int16)
        if bits == 8:
            ref = (ref.to(torch.int32) * 256).to(torch.int16)
        assert torch.equal(got, ref.view(torch.int16) if ref.dtype != torch.int16 else ref), 'shift %d' % s
    # shift linearity: the disk of shift s on curve c equals the disk of shift 0 on curve c + s
    fit2 = fit.copy()
    fit2[:, 0] += 3
    fit2[:, 3] += 3
    ind2, lw2, rw2 = column_plan(fit2, [0], ih, iw)
    ind3, lw3, rw3 = column_plan(fit, [3], ih, iw)
    assert torch.equal(ops.extract_columns(stack, ind2, lw2, rw2).view(torch.int16),
                       ops.extract_columns(stack, ind3, lw3, rw3).view(torch.int16))
    # half-way weights: disk = trunc((L + R) / 2) exactly
    fit_h = fit.copy()
    fit_h[:, 1] = 0.5
    ind_h, lw_h, rw_h = column_plan(fit_h, [0], ih, iw)
    half = ops.extract_columns(stack, ind_h, lw_h, rw_h)[0].view(torch.int16).to(torch.int32) & 0xffff
    left = gather_reference(stack, fit[:, 0], 0, rotated).to(torch.int32) & (0xffff if bits == 16 else 0xff)
    right = gather_reference(stack, fit[:, 0] + 1, 0, rotated).to(torch.int32) & (0xffff if bits == 16 else 0xff)
    assert torch.equal(half, ((left + right) * scale) // 2)


def test_whole_path_c2_against_the_oracle(env):
    """BASELINE configs[1] end to end on the GPU vs the CPU oracle (about 15 s of NumPy)."""
    ops, synth = env
    from oracle import pipeline_oracle as po
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon
    from solex_ser_recon_en_amd.video_reader import array_reader
    stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=0)
    opts = SHG_MAIN.default_options()
    opts['_nolog'] = True
    disk_list, bounds, hdr = Solex_recon.solex_read(array_reader(stack), opts)
    (cc, protus), = Solex_recon.solex_process(opts, disk_list, bounds, hdr)
    with np.errstate(all='ignore'):
        want = po.run(stack.cpu().numpy(), {})
    for got, ref in zip(disk_list, want['read']['disks']):
        np.testing.assert_array_equal(np.asarray(got), ref)                      # raw disks: bit exact
    np.testing.assert_allclose(opts['ratio_fixe'], want['geometry']['ratio'], rtol=1e-9)
    for name, got, ref in (('cc', cc, want['results'][0]['cc']), ('protus', protus, want['results'][0]['protus'])):
        flips('C2 stage route %s' % name, got, ref)


def close_products(what, results, want, shifts, observed=0):
    for (cc, protus), shift in zip(results, shifts):
        for name, got, ref in (('cc', cc, want['results'][shift]['cc']), ('protus', protus, want['results'][shift]['protus'])):
            flips('%s shift %d %s' % (what, shift, name), got, ref, observed)


def test_multishift_c4_against_the_oracle(env):
    """BASELINE configs[3] at its size: 2000 frames of 2000x200, -w -10:10:1 -> 21 requested disks.  Three of them
    (both ends of the Doppler range and the line centre) are held against the oracle, which runs the same scan with
    just those shifts (a disk depends on its own shift, the common fit and the common limb geometry only); the
    shift-0 products also equal the single-shift run bit for bit."""
    ops, synth = env
    from oracle import pipeline_oracle as po
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon
    from solex_ser_recon_en_amd.video_reader import array_reader
    stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=1)

    def run(shifts):
        opts = SHG_MAIN.default_options()
        opts.update(_nolog=True, shift=list(shifts))
        disk_list, bounds, hdr = Solex_recon.solex_read(array_reader(stack), opts)
        return opts, disk_list, Solex_recon.solex_process(opts, disk_list, bounds, hdr)
    o21, d21, r21 = run(range(-10, 11))
    o1, d1, r1 = run([0])
    assert o21['shift'] == [10, 0] + [s for s in range(-10, 11) if s not in (10, 0)]
    assert len(r21) == 21 and len(r1) == 1
    assert o21['ratio_fixe'] == o1['ratio_fixe'] and o21['slant_fix'] == o1['slant_fix']
    requested = [s for s in o21['shift'] if s in o21['shift_requested']]
    idx0 = requested.index(0)
    np.testing.assert_array_equal(np.asarray(r21[idx0][0]), np.asarray(r1[0][0]))
    np.testing.assert_array_equal(np.asarray(r21[idx0][1]), np.asarray(r1[0][1]))
    probe = [-10, 0, 10]
    with np.errstate(all='ignore'):
        want = po.run(stack.cpu().numpy(), {'shift': probe})
    for shift, ref in zip(want['read']['shifts'], want['read']['disks']):
        np.testing.assert_array_equal(np.asarray(d21[o21['shift'].index(shift)]), ref)          # raw disks: bit exact
    np.testing.assert_allclose(o21['ratio_fixe'], want['geometry']['ratio'], rtol=1e-9)
    close_products('C4 stage route', [r21[requested.index(s)] for s in probe], want, probe)


def test_whole_path_c3_size_against_the_oracle(env):
    """BASELINE configs[2]'s scan (4000 frames of 2000x200, 16 bit) whole on one GPU vs the CPU oracle (about 30 s of
    NumPy); tests/test_pipeline_gpu.py shards the same flow over ranks and shows bit-identical products."""
    ops, synth = env
    from oracle import pipeline_oracle as po
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon
    from solex_ser_recon_en_amd.video_reader import array_reader
    stack = synth.synth_frames_torch(4000, 2000, 200, 16, seed=2)
    opts = SHG_MAIN.default_options()
    opts['_nolog'] = True
    disk_list, bounds, hdr = Solex_recon.solex_read(array_reader(stack), opts)
    results = Solex_recon.solex_process(opts, disk_list, bounds, hdr)
    with np.errstate(all='ignore'):
        want = po.run(stack.cpu().numpy(), {})
    assert np.asarray(disk_list[0]).shape == (2000, 4000)
    for got, ref in zip(disk_list, want['read']['disks']):
        np.testing.assert_array_equal(np.asarray(got), ref)
    np.testing.assert_allclose(opts['ratio_fixe'], want['geometry']['ratio'], rtol=1e-9)
    close_products('C3 4000-frame scan', results, want, [0])


def test_c5_file_through_the_whole_path(env):
    """BASELINE configs[4]'s file shape (2560x256, 16 bit): the frame passes run on all 4000 frames in
    test_frame_passes_at_full_size; here the whole per-file flow, CLAHE included, on the first 1000 frames against
    the oracle, and on the full 4000-frame file for shape / determinism (two runs, identical products)."""
    ops, synth = env
    from oracle import pipeline_oracle as po
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon
    from solex_ser_recon_en_amd.video_reader import array_reader
    stack = synth.synth_frames_torch(4000, 2560, 256, 16, seed=5)

    def run(frames):
        opts = SHG_MAIN.default_options()
        opts['_nolog'] = True
        disk_list, bounds, hdr = Solex_recon.solex_read(array_reader(frames), opts)
        return opts, disk_list, Solex_recon.solex_process(opts, disk_list, bounds, hdr)
    head = stack[:1000]
    o, disks, results = run(head)
    with np.errstate(all='ignore'):
        want = po.run(ops.stack_to_host(head), {})
    for got, ref in zip(disks, want['read']['disks']):
        np.testing.assert_array_equal(np.asarray(got), ref)
    np.testing.assert_allclose(o['ratio_fixe'], want['geometry']['ratio'], rtol=1e-9)
    close_products('C5 first 1000 frames', results, want, [0])
    oa, da, ra = run(stack)
    ob, db, rb = run(stack)
    assert np.asarray(da[0]).shape == (2560, 4000) and np.asarray(ra[0][0]).shape[0] == 2560
    assert oa['ratio_fixe'] == ob['ratio_fixe']
    np.testing.assert_array_equal(np.asarray(ra[0][0]), np.asarray(rb[0][0]))
    np.testing.assert_array_equal(np.asarray(ra[0][1]), np.asarray(rb[0][1]))


def test_long_scan_12000_frames_against_the_oracle(env):
    """A slow scan: 12 000 8-bit frames of 800 x 64.  The raw disks are 12 000 px wide and the limb is a 15:1 ellipse
    that the warp squeezes back to a circle (ratio ~ 0.07)."""
    ops, synth = env
    from oracle import pipeline_oracle as po
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon
    from solex_ser_recon_en_amd.video_reader import array_reader
    stack = synth.synth_frames_torch(12000, 800, 64, 8, seed=4)
    opts = SHG_MAIN.default_options()
    opts['_nolog'] = True
    disk_list, bounds, hdr = Solex_recon.solex_read(array_reader(stack), opts)
    (cc, protus), = Solex_recon.solex_process(opts, disk_list, bounds, hdr)
    with np.errstate(all='ignore'):
        want = po.run(stack.cpu().numpy(), {})
    for got, ref in zip(disk_list, want['read']['disks']):
        np.testing.assert_array_equal(np.asarray(got), ref)
    np.testing.assert_allclose(opts['ratio_fixe'], want['geometry']['ratio'], rtol=1e-9)
    assert np.asarray(disk_list[0]).shape == (800, 12000) and opts['ratio_fixe'] < 0.1
    for name, got, ref in (('cc', cc, want['results'][0]['cc']), ('protus', protus, want['results'][0]['protus'])):
        flips('12000-frame scan %s' % name, got, ref)


def test_large_sensor_scan_against_the_oracle(env):
    """A 4096-row slit (large sensor), 3000 frames, 16 bit: 2.4 GB stack, 4096 x ~4300 px products."""
    ops, synth = env
    from oracle import pipeline_oracle as po
    from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon
    from solex_ser_recon_en_amd.video_reader import array_reader
    stack = synth.synth_frames_torch(3000, 4096, 96, 16, seed=6)
    opts = SHG_MAIN.default_options()
    opts['_nolog'] = True
    disk_list, bounds, hdr = Solex_recon.solex_read(array_reader(stack), opts)
    (cc, protus), = Solex_recon.solex_process(opts, disk_list, bounds, hdr)
    with np.errstate(all='ignore'):
        want = po.run(stack.cpu().numpy(), {})
    for got, ref in zip(disk_list, want['read']['disks']):
        np.testing.assert_array_equal(np.asarray(got), ref)
    np.testing.assert_allclose(opts['ratio_fixe'], want['geometry']['ratio'], rtol=1e-9)
    assert np.asarray(cc).shape[0] == 4096 and np.asarray(cc).shape == want['results'][0]['cc'].shape
    for name, got, ref in (('cc', cc, want['results'][0]['cc']), ('protus', protus, want['results'][0]['protus'])):
        flips('4096-row sensor %s' % name, got, ref)
