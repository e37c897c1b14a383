"""HIP kernels (through the C ABI) against the golden vectors and the CPU oracle.
Bit-exact for integer / index work; tolerances stated where floating point is involved."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from solex_ser_recon_en_amd import ops as _ops
    return _ops


@pytest.fixture(scope='module')
def orc():
    from oracle import shg_oracle
    return shg_oracle


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def frames_geometry(frames):
    n, h, w = frames.shape
    return n, h, w, frames.dtype.itemsize


# ---- pass A -----------------------------------------------------------------
@pytest.mark.parametrize('tag', ['u16_rot', 'u16_norot', 'u8_rot', 'u8_norot', 'u16_odd'])
def test_mean_max_golden(ops, golden, tag):
    g = golden('g1_mean_max')
    frames = g[tag + '_frames']
    n, h, w, bpp = frames_geometry(frames)
    total, mx = ops.accumulate_sum_max(dev(frames))
    mean, mxo = ops.finalize_mean_max(total, mx, n, h, w, bpp)
    np.testing.assert_array_equal(host(mean), g[tag + '_mean'])
    np.testing.assert_array_equal(host(mxo), g[tag + '_max'])
    np.testing.assert_array_equal(host(total), frames.astype(np.int64).sum(0).ravel())
    mean1, max1 = ops.accumulate_mean_max(dev(frames))          # the one-GPU form: no 64-bit totals in between
    np.testing.assert_array_equal(host(mean1), g[tag + '_mean'])
    np.testing.assert_array_equal(host(max1), g[tag + '_max'])


@pytest.mark.parametrize('shape,dtype', [((300, 24, 160), np.uint16), ((257, 160, 24), np.uint16),
                                         ((130, 16, 128), np.uint8), ((3, 5, 7), np.uint16), ((1, 8, 16), np.uint8)])
def test_mean_max_oracle_random(ops, orc, shape, dtype):
    rng = np.random.default_rng(42)
    hi = 256 if dtype == np.uint8 else 65536
    frames = rng.integers(0, hi, shape).astype(dtype)
    frames[rng.integers(0, shape[0])] = hi - 1                   # saturated frame: the max path sees full scale
    n, h, w, bpp = frames_geometry(frames)
    total, mx = ops.accumulate_sum_max(dev(frames))
    mean, mxo = ops.finalize_mean_max(total, mx, n, h, w, bpp)
    ref_mean, ref_max = orc.compute_mean_max(orc.SerReader(frames))
    np.testing.assert_array_equal(host(mean), ref_mean)
    np.testing.assert_array_equal(host(mxo), ref_max)
    mean1, max1 = ops.accumulate_mean_max(dev(frames))
    np.testing.assert_array_equal(host(mean1), ref_mean)
    np.testing.assert_array_equal(host(max1), ref_max)


def test_mean_max_sharding_is_bit_identical(ops):
    """Integer sum / max partials of frame shards add up to the unsharded result (what RCCL SUM/MAX does)."""
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 65536, (96, 16, 64)).astype(np.uint16)
    t_all, m_all = ops.accumulate_sum_max(dev(frames))
    parts = [ops.accumulate_sum_max(dev(frames[a:b])) for a, b in [(0, 31), (31, 64), (64, 96)]]
    t_sum = sum(host(p[0]) for p in parts)
    m_max = np.maximum.reduce([host(p[1]) for p in parts])
    np.testing.assert_array_equal(host(t_all), t_sum)
    np.testing.assert_array_equal(host(m_all), m_max)


def test_frame_statistics_of_the_ranks_folded_in_one_launch(ops):
    """shg_reduce_frame_stats: the pieces one all-gather brings -- per rank [sums as 32-bit words | maxima as 16-bit words | failure
    word] -- against NumPy, sums beyond 2^31 per piece and an odd pixel count (the maxima's last word half used) included."""
    rng = np.random.default_rng(77)
    for g, p in ((8, 400001), (2, 64), (3, 1)):
        sums = rng.integers(0, 2 ** 32, (g, p), dtype=np.uint64)
        maxs = rng.integers(0, 65536, (g, p), dtype=np.uint64).astype(np.uint16)
        words = p + (p + 1) // 2 + 1
        pieces = np.zeros((g, words), dtype=np.uint32)
        pieces[:, :p] = sums.astype(np.uint32)
        pieces[:, p:p + (p + 1) // 2].view(np.uint16)[:, :p] = maxs
        pieces[:, -1] = 0xdeadbeef
        total, mx = ops.reduce_frame_stats(torch.from_numpy(pieces.view(np.int32)).cuda(), p)
        np.testing.assert_array_equal(total.cpu().numpy().view(np.uint64), sums.sum(axis=0))
        np.testing.assert_array_equal(host(mx), maxs.max(axis=0))


# ---- line detection helpers ---------------------------------------------------
@pytest.mark.parametrize('h,w,kw,kh', [(180, 48, 25, 1), (180, 48, 5, 5), (64, 203, 25, 7), (33, 31, 4, 6), (12, 9, 25, 3)])
def test_box_blur_matches_oracle(ops, orc, h, w, kw, kh):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    np.testing.assert_array_equal(host(ops.box_blur_u16(dev(img), kw, kh)), orc.box_blur_u16(img, kw, kh))


def test_row_argmin_and_mean(ops):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 300, (211, 157)).astype(np.uint16)     # many ties: first occurrence matters
    np.testing.assert_array_equal(host(ops.row_argmin_u16(dev(img), 12, 157 - 13)), np.argmin(img[:, 12:-13], axis=1))
    np.testing.assert_array_equal(host(ops.row_argmin_u16(dev(img), 0, 157)), np.argmin(img, axis=1))
    np.testing.assert_array_equal(host(ops.row_mean_u16(dev(img))), np.mean(img, axis=1))


@pytest.mark.parametrize('h,w,kw,kh', [(180, 48, 25, 1), (180, 48, 5, 5), (203, 64, 25, 7), (33, 31, 4, 6), (12, 9, 25, 3),
                                      (2000, 200, 25, 17), (1001, 257, 25, 36), (7, 300, 5, 5)])
def test_fused_blur_reductions_match_the_separate_kernels_and_the_oracle(ops, orc, h, w, kw, kh):
    """shg_blur_row_mean_u16 / shg_blur_argmin_u16 (the blurred image stays in LDS) against cv2.blur's restatement
    followed by np.mean / np.argmin -- ties included (first occurrence)."""
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 400 if h % 2 else 65536, (h, w)).astype(np.uint16)
    blurred = orc.box_blur_u16(img, kw, kh)
    np.testing.assert_array_equal(host(ops.blur_row_mean_u16(dev(img), kw, kh)), np.mean(blurred, axis=1))
    x0, x1 = (12, w - 13) if w > 30 else (1, w - 1)
    a_blur, a_sharp = ops.blur_argmin_u16(dev(img), kw, kh, x0, x1)
    np.testing.assert_array_equal(host(a_blur), np.argmin(blurred[:, x0:x1], axis=1))
    np.testing.assert_array_equal(host(a_sharp), np.argmin(img, axis=1))
    np.testing.assert_array_equal(host(a_blur), host(ops.row_argmin_u16(ops.box_blur_u16(dev(img), kw, kh), x0, x1)))


# ---- pass B -----------------------------------------------------------------
@pytest.mark.parametrize('n,h,w,bits,shifts,slope,flip', [
    (150, 48, 600, 16, [10, 0], 0.01, False),              # rotated, two shifts
    (97, 40, 1028, 16, list(range(-10, 11)), 0.02, True),  # 21 shifts, ragged frame count, flip, clamps at both spectral edges
    (130, 36, 516, 8, [10, 0, -3], 0.03, False),           # 8-bit file: 4-byte requests
    (64, 48, 600, 16, [10, 0], 0.9, False),                # a steep line
    (64, 48, 602, 16, [10, 0], 0.01, False),               # slit length not a multiple of the tile
    # consecutive shifts in any order take the load-each-sample-once kernel (shg_extract_columns_dense):
    (75, 40, 700, 16, [2, 0, 1, 3, -1], 0.05, False),      # five of them, unordered, the line runs past both edges
    (61, 30, 333, 8, [1, 0, -1], 0.03, True),              # three, 8-bit, flip, ragged everything
    (50, 64, 520, 16, list(range(-12, 12)), 0.02, False),  # twenty-four: the most the dense path takes
    (40, 500, 36, 16, list(range(-3, 4)), 0.004, False),   # un-rotated file (Height > Width)
    (33, 26, 300, 16, list(range(-12, 12)), 0.0, False),   # frames barely wider than the shift range
])
def test_extract_stage_equals_the_kernel_entry_point_and_the_oracle(ops, orc, n, h, w, bits, shifts, slope, flip):
    """shg_stage_extract (host column plan + upload + kernel + extrema) against shg_extract_columns fed with the same plan and
    against the oracle's read_video_improved."""
    from solex_ser_recon_en_amd import hostmath, stages
    rng = np.random.default_rng(n + w)
    frames = rng.integers(0, 256 if bits == 8 else 65536, (n, h, w)).astype(np.uint8 if bits == 8 else np.uint16)
    ih, iw = (w, h) if w > h else (h, w)
    y = np.arange(ih)
    curve = iw / 2 + slope * (y - ih / 2) + 3 * np.sin(y / 40.0)          # runs past both spectral edges for the larger slopes: clamps
    fit = np.stack([np.floor(curve), curve - np.floor(curve), y.astype(float), curve], axis=1)
    got = host(stages.extract(dev(frames), fit, shifts, flip_x=flip))
    ind_l, lw, rw = hostmath.column_plan(fit, shifts, ih, iw)
    direct = host(ops.extract_columns(dev(frames), ind_l, lw, rw, flip_x=flip))
    np.testing.assert_array_equal(got, direct)
    if 3 <= len(shifts) <= 24 and max(shifts) - min(shifts) == len(shifts) - 1 and iw > len(shifts):
        # (the stage takes this kernel for un-rotated files only; the entry point serves both layouts)
        dense, mm = ops.extract_columns_dense(dev(frames), fit, shifts, flip_x=flip, want_minmax=True)
        np.testing.assert_array_equal(host(dense), direct)
        ext = mm.cpu().numpy().astype(np.int64)
        assert [tuple(e) for e in ext] == [(int(p.min()), int(p.max())) for p in direct]
    want = np.stack(orc.extract_columns(orc.SerReader(frames), fit, shifts))
    np.testing.assert_array_equal(got, want[:, :, ::-1] if flip else want)


@pytest.mark.parametrize('n,h,w,bits,shifts,case', [
    (200, 40, 1000, 16, [10, 0], 'plain'),                           # two shifts, one group of two
    (200, 40, 1000, 16, [10, 0, -7, 3, 25], 'plain'),                # any list: groups of four in plane order, a short last group
    (137, 40, 1000, 16, list(range(-10, 11)), 'plain'),              # 21 consecutive: three groups of seven values
    (137, 40, 1000, 16, [3, 0, 1, 2, -1, -2, -3, 4], 'plain'),       # eight consecutive, unordered: two groups of four
    (137, 40, 1000, 8, list(range(-4, 5)), 'plain'),                 # nine, 8-bit: two groups of five (the second short)
    (137, 40, 1000, 16, list(range(-10, 11)), 'shard'),              # a rank's frames inside a wider mosaic, not on a 16-byte boundary
    (137, 40, 1000, 16, [10, 0], 'shard'),
    (137, 40, 1000, 16, list(range(-10, 11)), 'wild'),               # weights the single-rounding product cannot take
    (64, 40, 1000, 8, [10, 0], 'wild'),
    (90, 30, 333, 16, list(range(-12, 12)), 'edge'),                 # the line within the shift range of both edges: every clamp
])
def test_band_kernel_equals_the_general_kernel(ops, monkeypatch, n, h, w, bits, shifts, case):
    """Rotated files go through k_extract_band (round 6); k_extract, which served them before and still serves un-rotated files,
    is the independent check: same disks bit for bit -- flipped, sharded into a wider mosaic, with weights outside [0, 1] (the
    reference's uint16 wrap) and beyond 2^900 (the plain-product path), and with every slit row on the clamps."""
    from solex_ser_recon_en_amd import hostmath
    rng = np.random.default_rng(n * 7 + w + len(shifts))
    frames = rng.integers(0, 256 if bits == 8 else 65536, (n, h, w)).astype(np.uint8 if bits == 8 else np.uint16)
    ih, iw = w, h
    y = np.arange(ih)
    slope = 0.06 if case == 'edge' else 0.01
    curve = iw / 2 + slope * (y - ih / 2) + 3 * np.sin(y / 40.0)
    fit = np.stack([np.floor(curve), curve - np.floor(curve), y.astype(float), curve], axis=1)
    ind_l, lw, rw = hostmath.column_plan(fit, shifts, ih, iw)
    if case == 'wild':
        lw = lw.copy()
        rw = rw.copy()
        lw[::7] = 1.75                                                # outside [0, 1]: the sum wraps as the uint16 cast does
        rw[5::11] = -0.5
        lw[100:164] = 1e280                                           # a whole wave beyond 2^900: plain products (saturating conversion)
    kw = dict(n_cols=n + 37, k_offset=21) if case == 'shard' else {}
    stack = dev(frames)
    consecutive = 3 <= len(shifts) <= 24 and max(shifts) - min(shifts) == len(shifts) - 1
    for flip in (False, True):
        monkeypatch.setenv('SHG_EXT_GENERAL', '1')
        want = host(ops.extract_columns(stack, ind_l, lw, rw, flip_x=flip, **kw))
        monkeypatch.delenv('SHG_EXT_GENERAL')
        got = host(ops.extract_columns(stack, ind_l, lw, rw, flip_x=flip, **kw))
        np.testing.assert_array_equal(got, want)
        if consecutive and case in ('plain', 'edge'):
            dense, mm = ops.extract_columns_dense(stack, fit, shifts, flip_x=flip, want_minmax=True)
            np.testing.assert_array_equal(host(dense), want)
            assert [tuple(e) for e in mm.cpu().numpy().astype(np.int64)] == [(int(p.min()), int(p.max())) for p in want]


@pytest.mark.parametrize('shifts', [[10, 0], list(range(-10, 11))])
def test_band_kernel_into_rows_that_are_not_16_byte_aligned(ops, monkeypatch, shifts):
    """The band kernel's 16-byte stores need rows on 16-byte boundaries; a caller's buffer whose row pitch is odd takes its pixel-by-pixel
    write-out (and the extrema from there): same disks as k_extract, the buffer's padding columns untouched."""
    from solex_ser_recon_en_amd import hostmath
    rng = np.random.default_rng(len(shifts))
    n, h, w = 150, 40, 700
    frames = rng.integers(0, 65536, (n, h, w)).astype(np.uint16)
    y = np.arange(w)
    curve = h / 2 + 0.004 * (y - w / 2)
    fit = np.stack([np.floor(curve), curve - np.floor(curve), y.astype(float), curve], axis=1)
    ind_l, lw, rw = hostmath.column_plan(fit, shifts, w, h)
    stack = dev(frames)
    outs = []
    for general in (True, False):
        if general:
            monkeypatch.setenv('SHG_EXT_GENERAL', '1')
        else:
            monkeypatch.delenv('SHG_EXT_GENERAL')
        buf = torch.full((len(shifts), w, n + 3), 0x5a5a, dtype=torch.int32, device='cuda').to(torch.uint16)
        ops.extract_columns(stack, ind_l, lw, rw, out=buf[:, :, :n])
        outs.append(host(buf))
    np.testing.assert_array_equal(outs[1], outs[0])
    assert (outs[1][:, :, n:] == 0x5a5a).all()


def test_pass_a_in_the_lane_waits_for_the_stack_its_caller_is_still_writing(ops):
    """With a frame-pass lane set, pass A runs on another stream than its caller's: it must still see a stack that the caller's
    stream has only queued the writing of (no synchronisation in between)."""
    from solex_ser_recon_en_amd import Solex_recon
    Solex_recon._ensure_lane(torch.device('cuda', torch.cuda.current_device()))
    for trial in range(4):
        base = torch.randint(0, 60000, (1500, 64, 512), dtype=torch.int32, device='cuda')
        torch.cuda.synchronize()
        work = base
        for _ in range(6):                                                # a queue of elementwise kernels the stack depends on
            work = (work * 3 + 7) % 60001
        stack = work.to(torch.int16).view(torch.uint16)
        mean, mx = ops.accumulate_mean_max(stack)                         # queued right behind them
        torch.cuda.synchronize()
        mean2, mx2 = ops.accumulate_mean_max(stack)
        torch.cuda.synchronize()
        assert torch.equal(mean.view(torch.int16), mean2.view(torch.int16)) and torch.equal(mx.view(torch.int16), mx2.view(torch.int16))


def test_pass_a_launched_ahead_is_found_by_its_scan_and_by_nobody_else(ops):
    """shg_pass_a_prelaunch starts the pass on the lane; accumulate_mean_max with the same stack and workspace only finalises
    it, one with another stack waits for the stray pass and runs its own; shg_pass_a_forget drops a pass nobody came for."""
    from solex_ser_recon_en_amd import Solex_recon, _lib
    Solex_recon._ensure_lane(torch.device('cuda', torch.cuda.current_device()))
    g = torch.Generator(device='cuda').manual_seed(11)
    a = torch.randint(0, 65536, (700, 48, 640), dtype=torch.int32, device='cuda', generator=g).to(torch.uint16)
    b = torch.randint(0, 65536, (700, 48, 640), dtype=torch.int32, device='cuda', generator=g).to(torch.uint16)
    want_a, want_b = ops.accumulate_mean_max(a), ops.accumulate_mean_max(b)
    ws = torch.empty(_lib.lib.shg_accumulate_workspace_bytes(700, 48, 640, 2), dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()

    def same(x, y):
        return all(torch.equal(p.view(torch.int16), q.view(torch.int16)) for p, q in zip(x, y))
    _lib.profile_enable(True, only=['accumulate'])
    _lib.profile_reset()
    try:
        assert ops.pass_a_prelaunch(a, ws)
        got = ops.accumulate_mean_max(a, ws)                              # the pass launched ahead: no second launch
        torch.cuda.synchronize()
        assert same(got, want_a)
        assert ops.pass_a_prelaunch(a, ws)
        got = ops.accumulate_mean_max(b, ws)                              # another stack: waits for the stray pass, runs its own
        torch.cuda.synchronize()
        assert same(got, want_b)
        launches = _lib.profile_get('accumulate')[1]
    finally:
        _lib.profile_enable(False)
    assert launches == 3
    assert ops.pass_a_prelaunch(b, ws)
    ws.zero_()                                                            # (queued behind nothing: the pass may still be writing)
    assert _lib.lib.shg_pass_a_forget(ws.data_ptr()) == 0
    torch.cuda.synchronize()
    got = ops.accumulate_mean_max(a, ws)
    torch.cuda.synchronize()
    assert same(got, want_a)


def test_back_to_back_extract_stages_do_not_share_a_staging_area(ops, orc):
    """shg_stage_extract returns while its copy kernel has yet to read the pinned staging area: two calls in a row with
    DIFFERENT fits (nothing in between that waits for the stream) must each sample their own columns."""
    from solex_ser_recon_en_amd import stages
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 65536, (600, 40, 800)).astype(np.uint16)
    stack = dev(frames)
    ih, iw = 800, 40
    y = np.arange(ih)
    fits = []
    for off in (9.3, 21.7, 14.1, 27.9, 11.2, 18.8):
        curve = off + 0.004 * y
        fits.append(np.stack([np.floor(curve), curve - np.floor(curve), y.astype(float), curve], axis=1))
    torch.cuda.synchronize()
    outs = [stages.extract(stack, f, [3, 0]) for f in fits]              # queued back to back
    for f, out in zip(fits, outs):
        want = np.stack(orc.extract_columns(orc.SerReader(frames), f, [3, 0]))
        np.testing.assert_array_equal(host(out), want)


def test_stage_refuses_a_staging_area_the_gpu_cannot_address(ops):
    """The stage composites store host-bound results straight into the staging area and read their plans from it: it has to
    be page-locked, GPU-mapped memory.  Pageable memory is refused with the reason, nothing is launched."""
    import ctypes
    from solex_ser_recon_en_amd import _lib
    lib = _lib.lib
    n, h, w = 8, 16, 64
    frames = dev(np.zeros((n, h, w), dtype=np.uint16))
    ih = w
    fit = np.zeros((ih, 4))
    shifts = np.zeros(1, dtype=np.int32)
    need = lib.shg_stage_extract_workspace_bytes(h, w, 1)
    ws = torch.empty(need, dtype=torch.uint8, device='cuda')
    pageable = np.zeros(need, dtype=np.uint8)
    out = torch.empty((1, ih, 64), dtype=torch.uint16, device='cuda')
    rc = lib.shg_stage_extract(frames.data_ptr(), n, h, w, 2, ops.frame_stride(frames), fit.ctypes.data, shifts.ctypes.data, 1,
                               out.data_ptr(), out.stride(1), out.stride(0), n, 0, 0, None, ws.data_ptr(), need,
                               pageable.ctypes.data, need, ops._stream())
    assert rc == -1                                                        # SHG_E_ARG
    assert b'page-locked' in lib.shg_last_error_string()
    pinned = torch.empty(need, dtype=torch.uint8).pin_memory()
    rc = lib.shg_stage_extract(frames.data_ptr(), n, h, w, 2, ops.frame_stride(frames), fit.ctypes.data, shifts.ctypes.data, 1,
                               out.data_ptr(), out.stride(1), out.stride(0), n, 0, 0, None, ws.data_ptr(), need,
                               pinned.data_ptr(), need, ops._stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert int(host(out)[0, :, :n].max()) == 0


@pytest.mark.parametrize('shape,flip', [((70, 40, 300), False), ((33, 130, 24), True), ((257, 24, 200), False)])
def test_extract_gathers_the_extrema_the_warp_clips_to(ops, orc, shape, flip):
    """shg_extract_columns_minmax leaves every plane's min / max, and the warp that takes them equals the warp
    that looks for them itself (ragged tiles, both orientations, flip)."""
    from solex_ser_recon_en_amd import stages
    rng = np.random.default_rng(shape[0])
    frames = rng.integers(100, 60000, shape).astype(np.uint16)
    n, h, w = shape
    ih, iw = max(h, w), min(h, w)
    curve = np.linspace(2.3, iw - 3.4, ih)
    fit = np.stack([np.floor(curve), curve - np.floor(curve), np.arange(ih), curve], axis=1)
    shifts = [10, 0, -2]
    disks, mm = stages.extract(dev(frames), fit, shifts, flip_x=flip, want_minmax=True)
    plain = stages.extract(dev(frames), fit, shifts, flip_x=flip)
    np.testing.assert_array_equal(host(disks), host(plain))
    extrema = mm.cpu().numpy().astype(np.int64)
    d = host(disks)
    for s in range(len(shifts)):
        assert tuple(extrema[s]) == (d[s].min(), d[s].max())
        a = ops.warp_rows_u16(disks[s], 0.97, 0.013, 1.7, ih, n + 9, minmax=mm[s])
        b = ops.warp_rows_u16(disks[s], 0.97, 0.013, 1.7, ih, n + 9)
        np.testing.assert_array_equal(host(a), host(b))


@pytest.mark.parametrize('tag', ['u16_rot', 'u16_norot', 'u8_rot', 'u16_odd'])
@pytest.mark.parametrize('stag', ['s2', 's21', 's3'])
def test_extract_golden(ops, orc, golden, tag, stag):
    g = golden('g2_extract')
    frames = g[tag + '_frames']
    shifts = [int(s) for s in g[tag + '_' + stag + '_shifts']]
    iw = min(frames.shape[1:])
    cols, lw, rw = orc.column_indices(g[tag + '_fit'], shifts, iw)
    ind_l = np.stack([c[0] for c in cols]).astype(np.int32)
    disks = ops.extract_columns(dev(frames), ind_l, lw, rw)
    np.testing.assert_array_equal(host(disks), g[tag + '_' + stag + '_disks'])


def test_extract_flip_and_column_blocks(ops, orc, golden):
    """flip_x (Solex_recon.py:74-76) and per-rank column blocks assemble to the unsharded disks."""
    g = golden('g2_extract')
    frames = g['u16_rot_frames']
    shifts = [10, 0]
    cols, lw, rw = orc.column_indices(g['u16_rot_fit'], shifts, min(frames.shape[1:]))
    ind_l = np.stack([c[0] for c in cols]).astype(np.int32)
    want = g['u16_rot_s2_disks']
    n = frames.shape[0]
    flipped = ops.extract_columns(dev(frames), ind_l, lw, rw, flip_x=True)
    np.testing.assert_array_equal(host(flipped), want[:, :, ::-1])
    for flip in (False, True):
        out = None
        for a, b in [(0, 7), (7, 13), (13, n)]:
            out = ops.extract_columns(dev(frames[a:b]), ind_l, lw, rw, n_cols=n, k_offset=a, flip_x=flip, out=out)
        np.testing.assert_array_equal(host(out), want[:, :, ::-1] if flip else want)


@pytest.mark.parametrize('n,h,w,dtype', [(150, 24, 200, np.uint16), (70, 200, 24, np.uint16), (130, 24, 200, np.uint8)])
def test_extract_oracle_random(ops, orc, n, h, w, dtype):
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256 if dtype == np.uint8 else 65536, (n, h, w)).astype(dtype)
    ih, iw = max(h, w), min(h, w)
    curve = iw / 2 + 3 * np.sin(np.arange(ih) / 17.0) + rng.random(ih)
    fit = np.stack([np.floor(curve), curve - np.floor(curve), np.arange(ih), curve], axis=1)
    shifts = orc.shift_list(10, list(range(-4, 5, 2)))
    cols, lw, rw = orc.column_indices(fit, shifts, iw)
    disks = ops.extract_columns(dev(frames), np.stack([c[0] for c in cols]).astype(np.int32), lw, rw)
    want = orc.extract_columns(orc.SerReader(frames), fit, shifts)
    np.testing.assert_array_equal(host(disks), np.stack(want))


@pytest.mark.parametrize('h,w,dtype', [(24, 200, np.uint16), (200, 24, np.uint16), (24, 200, np.uint8)])
def test_extract_with_weights_no_fit_would_give(ops, h, w, dtype):
    """k_extract forms sample * weight as fma(2^52 + sample, w, -2^52 w) where that is exact (every fit's weights, which lie in [0, 1])
    and as the plain product where 2^52 w is not finite -- wave by wave.  Weights through the C ABI may be anything: rows with
    NaN, +-inf, 1e300, negative and > 1 weights, mixed with ordinary rows inside one wave and in waves of their own, must give
    what NumPy's float64 arithmetic and its uint16 cast give (solex_util.py:128-133: left * lw + right * rw, astype('uint16'))."""
    rng = np.random.default_rng(21)
    n = 70
    full = 256 if dtype == np.uint8 else 65536
    frames = rng.integers(0, full, (n, h, w)).astype(dtype)
    frames[:, :3, :3] = 0
    ih, iw = max(h, w), min(h, w)
    col = rng.integers(0, iw - 1, (2, ih)).astype(np.int32)
    lw = rng.random(ih)
    rw = 1.0 - lw
    odd = [np.nan, np.inf, -np.inf, 1e300, -1e300, 2.5, -0.75, 1e-320, 0.0, 65536.0]
    for i, v in enumerate(odd):                           # single odd rows among ordinary ones (one wave holds both kinds)
        lw[3 + 7 * i] = v
        rw[5 + 7 * i] = v
    lw[128:192] = np.resize(odd, 64)                      # a whole wave of them
    rw[128:192] = np.resize(odd[::-1], 64)
    disks = host(ops.extract_columns(dev(frames), col, lw, rw))
    img = np.transpose(frames, (0, 2, 1))[:, ::-1, :] if w > h else frames          # video_reader.py:119-120: rot90 of every frame
    img = img.astype(np.int64) * (256 if dtype == np.uint8 else 1)
    rows = np.arange(ih)
    with np.errstate(all='ignore'):
        for s in range(2):
            left = img[:, rows, col[s]].astype(np.float64)
            right = img[:, rows, col[s] + 1].astype(np.float64)
            val = left * lw[None, :] + right * rw[None, :]
            ordinary = np.isfinite(val) & (np.abs(val) < 2.0 ** 31)
            want = np.where(ordinary, val, 0).astype(np.int64).astype(np.uint16)
            got = disks[s].T                               # [n, ih]
            np.testing.assert_array_equal(got[ordinary], want[ordinary])
            # what a cast of NaN / out-of-range values gives is the hardware's business (v_cvt_i32_f64 saturates, NaN -> 0), but it must
            # be the same whichever form the wave took: the odd rows inside ordinary waves against the all-odd wave
            sat = np.where(np.isnan(val), 0, np.clip(np.where(np.isnan(val), 0, val), -2.0 ** 31, 2.0 ** 31 - 1)).astype(np.int64)
            np.testing.assert_array_equal(got, (sat & 0xffff).astype(np.uint16))


# ---- warp ---------------------------------------------------------------------
def test_warp_golden_bit_exact(ops, golden):
    g = golden('g3_warp')
    img = g['image_u16']
    for i in range(6):
        mat3 = g['c%d_mat3' % i]
        want = g['c%d_out' % i]
        out = ops.warp_rows_u16(dev(img), mat3[0, 0], mat3[0, 1], mat3[0, 2], want.shape[0], want.shape[1])
        np.testing.assert_array_equal(host(out), want)


@pytest.mark.parametrize('h,w,oh,ow,h00,h01,h02', [
    (300, 517, 300, 517, 1.0, 0.0, 0.0),            # the identity: every position whole (weight 0 on a neighbour that may lie outside)
    (300, 517, 300, 540, 0.957, 0.013, -9.4),       # a typical correction: off the image on the left, shear
    (300, 517, 310, 400, 1.31, -0.02, 3.0),         # more output rows than the source has
    (257, 1003, 257, 1003, 1.0, 0.0, 5.0),          # whole positions, shifted: runs off the right edge
    (64, 200, 64, 131, 1.75, 0.1, -20.25),          # the widest column step the 24-pixel window takes
    (64, 200, 64, 131, 1.76, 0.0, 0.0),             # beyond it: the narrow kernel serves
    (64, 200, 64, 300, 0.31, 0.0, 190.0),           # neighbouring outputs share their samples; most of the row beyond the image
    (33, 9, 33, 40, 1.0, 0.5, -3.0),                # an image narrower than a piece's stride
    (70, 4200, 70, 4200, 0.98, 0.004, 17.3),        # wide: most waves lie inside their rows (the copy of the blend without border tests)
    (50, 300, 50, 300, 1.0, 0.0, -1e-13),           # a hair left of whole positions: weights 1e-13 and 1 - 1e-13, the first pixel at -1e-13
    (40, 64, 40, 16, 1.0, 0.0, 3.0),                # two vectors a row: the smallest divisor of the multiply-high
    (40, 5000, 40, 4096, 1.2, 0.0, -7.5),           # 512 vectors a row: a power of two
    (2100, 8200, 2100, 8192, 0.999, 0.0001, 2.0),   # 2.15 M vectors: four of them a thread
])
def test_warp_eight_pixels_a_lane_equals_one_pixel_a_lane(ops, monkeypatch, h, w, oh, ow, h00, h01, h02):
    """k_warp_rows8 (round 6: three 16-byte pieces of the source row per lane, LDS window, one 16-byte store) against k_warp_rows
    (two 2-byte gathers per pixel), which g3 pins to scikit-image: same pixels on transforms that run off either edge, land on
    whole positions, squeeze and stretch."""
    rng = np.random.default_rng(h * 31 + w + ow)
    img = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    img[:, :3] = 65535                                                # saturated columns at the edge: the clip's upper bound in play
    monkeypatch.setenv('SHG_WARP_WIDE', '0')
    want = host(ops.warp_rows_u16(dev(img), h00, h01, h02, oh, ow))
    monkeypatch.delenv('SHG_WARP_WIDE')
    got = host(ops.warp_rows_u16(dev(img), h00, h01, h02, oh, ow))
    np.testing.assert_array_equal(got, want)


# ---- transversalium -------------------------------------------------------------
def test_transversalium_stats_and_scale(ops, orc, golden):
    import math
    g = golden('g4_transversalium')
    img = g['image']
    circle, borders = tuple(g['circle']), list(g['borders'])
    y1, y2, want = orc.transversalium_row_stats(img, circle, borders)
    xa = np.zeros(y2 - y1, np.int32)
    xb = np.zeros(y2 - y1, np.int32)
    for y in range(y1 + 1, y2):
        dx = math.floor((circle[2] ** 2 - (y - circle[1]) ** 2) ** 0.5)
        a, b, _ = slice(math.ceil(max(circle[0] - dx, borders[0])), math.floor(min(circle[0] + dx, borders[2]))).indices(img.shape[1])
        xa[y - y1], xb[y - y1] = a, max(a, b)
    got = host(ops.rowpair_logratio_stats(dev(img), y1, y2, xa, xb))
    # float64 log + a different summation order: a few ulp of values ~1e-2
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)
    for tag in ('a', 'b'):
        out = host(ops.scale_rows_u16(dev(img), g[tag + '_c']))
        np.testing.assert_array_equal(out, g[tag + '_out'])


# ---- crop, rescale, disc, downscale, hist ---------------------------------------
def test_crop_pad(ops):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 65536, (37, 91)).astype(np.uint16)
    out = host(ops.crop_pad_u16(dev(img), 50, 20, 5, 40, 777))
    want = np.full((37, 50), 777, np.uint16)
    want[:, 5:45] = img[:, 20:60]
    np.testing.assert_array_equal(out, want)
    want[:, :5] = img[0, 0]; want[:, 45:] = img[0, 0]
    np.testing.assert_array_equal(host(ops.crop_pad_u16(dev(img), 50, 20, 5, 40, None)), want)     # fill = img[0, 0] on the device


def test_rescale_golden(ops, golden):
    g = golden('g5_rescale')
    img = g['image']
    bright = float(g['bright'])
    np.testing.assert_array_equal(host(ops.rescale_u16(dev(img), bright * 0.25, bright)), g['hc'])
    np.testing.assert_array_equal(host(ops.rescale_u16(dev(img), 0, bright * 0.18)), g['protus'])
    np.testing.assert_array_equal(host(ops.rescale_u16(dev(img), 1000.0, 50000.0, 0.8)), g['alpha'])
    with pytest.raises(RuntimeError):
        ops.rescale_u16(dev(img), 10.0, 10.0)


@pytest.mark.parametrize('x0,y0,r', [(40, 30, 17), (3, 4, 9), (79, 59, 30), (40, 30, 1), (200, 30, 5)])
def test_fill_disc_matches_oracle(ops, orc, x0, y0, r):
    img = np.full((60, 80), 1000, np.uint16)
    want = orc.filled_circle(img.copy(), x0, y0, r, 80)
    np.testing.assert_array_equal(host(ops.fill_disc_u16(dev(img), x0, y0, r, 80)), want)


def test_downscale_and_hist(ops):
    rng = np.random.default_rng(6)
    img = rng.integers(0, 65536, (51, 70)).astype(np.uint16)
    got = host(ops.downscale_mean_u16(dev(img), 4))
    pad = np.zeros((52, 72))
    pad[:51, :70] = img / 65536
    want = pad.reshape(13, 4, 18, 4).mean(axis=(1, 3))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(host(ops.histogram(dev(img))), np.bincount(img.ravel(), minlength=65536))
    img8 = (img >> 8).astype(np.uint8)
    np.testing.assert_array_equal(host(ops.histogram(dev(img8))), np.bincount(img8.ravel(), minlength=256))


# ---- CLAHE ------------------------------------------------------------------------
@pytest.mark.parametrize('h,w,tiles,dtype', [(64, 64, 2, np.uint16), (150, 161, 2, np.uint16), (90, 120, 3, np.uint16),
                                             (77, 50, 4, np.uint8), (40, 40, 1, np.uint16), (300, 340, 2, np.uint16)])
def test_clahe_matches_oracle(ops, orc, h, w, tiles, dtype):
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:h, 0:w]
    full = 255 if dtype == np.uint8 else 65535
    smooth = 0.5 + 0.4 * np.sin(xx / 13.0) * np.cos(yy / 11.0)
    img = np.clip((smooth + 0.05 * rng.standard_normal((h, w))) * full, 0, full).astype(dtype)
    got = host(ops.clahe(dev(img), 0.8, tiles))
    want = orc.clahe(img, 0.8, tiles)
    # float32 arithmetic restated operation by operation: identical bits expected
    np.testing.assert_array_equal(got, want)
    # the other histogram path (global atomics, one workgroup per tile LUT) gives the same image
    np.testing.assert_array_equal(host(ops.clahe(dev(img), 0.8, tiles, small_workspace=True)), want)


@pytest.mark.parametrize('h,w,tiles,clip_limit', [(1100, 1000, 2, 0.8), (1100, 1000, 2, 40.0), (901, 1203, 2, 3.0), (640, 960, 4, 8.0),
                                                  (1100, 1000, 2, 0.0), (1100, 1008, 2, 0.8), (700, 1056, 3, 2.0)])
def test_clahe_paths_agree_with_the_oracle_on_several_slices_per_tile(ops, orc, h, w, tiles, clip_limit):
    """Tiles of more than one 32768-pixel slice, clip limits from 1 count to hundreds (batch and residual redistribution
    both at work), a padded (reflected) tile grid, no clipping at all, tile rows that are whole 16-byte vectors (the slice
    histogram's vector loads: widths 1008 and 1056) and ones that are not: slice histograms + block-wise LUT, the atomics path
    and the oracle agree bit for bit.  A solar-like image: a bright disc with limb darkening on a near-constant sky."""
    rng = np.random.default_rng(17)
    yy, xx = np.mgrid[0:h, 0:w]
    r = np.hypot(yy - h / 2, xx - w / 2) / (0.42 * min(h, w))
    disc = np.where(r < 1, 0.35 + 0.55 * np.sqrt(np.clip(1 - r * r, 0, 1)), 0.01)
    img = np.clip((disc + 0.01 * rng.standard_normal((h, w))) * 65535, 0, 65535).astype(np.uint16)
    want = orc.clahe(img, clip_limit, tiles)
    np.testing.assert_array_equal(host(ops.clahe(dev(img), clip_limit, tiles)), want)
    np.testing.assert_array_equal(host(ops.clahe(dev(img), clip_limit, tiles, small_workspace=True)), want)


# ---- limb detection kernels -----------------------------------------------------------
@pytest.mark.parametrize('h,w,k', [(103, 115, 1), (103, 115, 5), (64, 50, 4), (31, 77, 6)])
def test_box_blur_f64_matches_oracle(ops, orc, h, w, k):
    rng = np.random.default_rng(11)
    img = rng.random((h, w))
    np.testing.assert_array_equal(host(ops.box_blur_f64(dev(img), k)), orc.box_blur_f64(img, k, k))


def host_hysteresis(low_mask, high_mask):
    """skimage canny's last step with scipy.ndimage.label (reference for the GPU labelling)."""
    from scipy import ndimage as ndi
    labels, count = ndi.label(low_mask, np.ones((3, 3), bool))
    good = np.zeros(count + 1, dtype=bool)
    good[np.unique(labels[high_mask])] = True
    good[0] = False
    return good[labels]


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_edge_components_match_scipy_label(ops, seed):
    """Random blobs, spirals and long thin chains: components, hysteresis and raster numbering vs scipy."""
    from scipy import ndimage as ndi
    from tests import numpy_ref
    rng = np.random.default_rng(seed)
    h, w = [(97, 131), (256, 256), (500, 500), (33, 700)][seed]
    low = rng.random((h, w)) < [0.3, 0.45, 0.05, 0.5][seed]
    t = np.linspace(0, 12 * np.pi, 20000)
    yy = np.clip((h / 2 + (h / 2.2) * t / t.max() * np.sin(t)).astype(int), 0, h - 1)
    xx = np.clip((w / 2 + (w / 2.2) * t / t.max() * np.cos(t)).astype(int), 0, w - 1)
    low[yy, xx] = True                                       # a long spiral: deep union-find chains
    high = low & (rng.random((h, w)) < 0.02)
    idx, root = ops.edge_components(dev(low.astype(np.uint8)), dev(high.astype(np.uint8)), prefetch=1000)
    want = host_hysteresis(low, high)
    np.testing.assert_array_equal(idx, np.flatnonzero(want))
    labelled, nf = ndi.label(want, np.ones((3, 3), int))
    lab, n = numpy_ref.labels_from_roots(root)
    assert n == nf
    np.testing.assert_array_equal(lab, labelled.ravel()[idx])
    # nothing survives without a high pixel
    idx0, _ = ops.edge_components(dev(low.astype(np.uint8)), dev(np.zeros((h, w), np.uint8)))
    assert idx0.size == 0


def test_canny_masks_match_skimage_0_18_3(ops, orc, golden):
    """GPU canny (Gaussian, Sobel, hypot, NMS, thresholds) + host hysteresis == the real scikit-image output."""
    g = golden('g13_limb')
    small = g['small']
    k = int(small.shape[0] * 0.01)
    blurred = orc.box_blur_f64(small, k, k)
    from tests.test_host_cpu import flood_threshold_numpy
    thresh3 = flood_threshold_numpy(small, blurred)
    for i in range(3):
        sigma, lo, hi = g['canny%d_params' % i]
        low_m, high_m = ops.canny_masks(dev(blurred), thresh3, sigma, lo, hi)
        edges = host_hysteresis(host(low_m).astype(bool), host(high_m).astype(bool))
        np.testing.assert_array_equal(edges, g['canny%d' % i])
        idx, root = ops.edge_components(low_m, high_m)                 # the same on the GPU
        np.testing.assert_array_equal(idx, np.flatnonzero(edges))
    # a noisy gray-level image (no flooding): thresholds that bite, real hysteresis
    noisy = g['noisy']
    low_m, high_m = ops.canny_masks(dev(noisy * 65000.0 + 1.0), 0.5, 1.0, 0.05 * 65000, 0.12 * 65000)
    assert host(low_m).shape == noisy.shape


def test_limb_points_stage_matches_oracle(ops, golden):
    """shg_stage_limb_points (block mean -> flood image -> canny -> labelling -> region / hull / row selection) on a
    uint16 disk against the oracle's get_edge_list on the same block mean."""
    from oracle import limb_oracle as limb
    from solex_ser_recon_en_amd import stages, synth
    rng = np.random.default_rng(3)
    frames = synth.synth_frames_numpy(900, 820, 24, 16, seed=2)
    disk = np.ascontiguousarray(frames[:, 12, :].T)                      # [820 rows, 900 columns]: a limb-darkened disk on sky
    disk = np.clip(disk.astype(np.int64) + rng.integers(-40, 40, disk.shape), 0, 65535).astype(np.uint16)
    X, raw = stages.limb_points(dev(disk))
    small = limb.downscale_local_mean(disk / 65536, 4)
    Xo, rawo = limb.get_edge_list(small.copy())
    np.testing.assert_array_equal(X, Xo * 4)
    np.testing.assert_array_equal(raw, rawo * 4)
    with pytest.raises(RuntimeError, match='at least 400 slit rows'):
        stages.limb_points(dev(disk[:300]))


@pytest.mark.parametrize('h,w,k', [(203, 230, 2), (500, 525, 5), (97, 131, 7), (640, 1000, 6), (30, 40, 63)])
def test_integer_key_select_equals_the_float64_select(ops, orc, h, w, k):
    """Order statistics of cv2.blur(block mean) through the integer window sums (shg_box_blur_key_f64 + shg_select_keys_u32,
    three 11-bit passes) against the float64 radix select on the blurred image and against np.sort."""
    rng = np.random.default_rng(h + k)
    small = rng.integers(0, 1 << 20, (h, w)).astype(np.float64) / (1 << 20)
    small[: h // 3] = small[0, 0]                                          # heavy ties
    blurred, keys = ops.box_blur_key_f64(dev(small), k)
    plain = ops.box_blur_f64(dev(small), k)
    np.testing.assert_array_equal(host(blurred), host(plain))
    np.testing.assert_array_equal(host(blurred), orc.box_blur_f64(small, k, k))
    n = h * w
    ranks = [0, n // 2 - 1, n // 2, int(0.99 * (n - 1)), int(0.99 * (n - 1)) + 1, n - 1]
    got = host(ops.select_keys_u32([keys] * len(ranks), ranks, [k] * len(ranks)))
    np.testing.assert_array_equal(got, host(ops.select_f64(plain, ranks)))
    np.testing.assert_array_equal(got, np.sort(host(plain).ravel())[ranks])


@pytest.mark.parametrize('n', [1, 2, 5, 1000, 250000])
def test_select_f64_is_exact(ops, n):
    rng = np.random.default_rng(n)
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)
    if n > 10:
        v[: n // 4] = v[0]                       # heavy ties
        v[-3:] = [0.0, -0.0, 1e-300]
    srt = np.sort(v)
    ranks = sorted(set([0, n // 2, max(n // 2 - 1, 0), n - 1, int(0.99 * (n - 1))]))
    got = host(ops.select_f64(dev(v), ranks))
    np.testing.assert_array_equal(got, srt[ranks])


def test_flood_stats_match_numpy(ops, golden):
    g = golden('g13_limb')
    small = g['small']                           # block means of uint16/65536: multiples of 2^-20
    from oracle import shg_oracle
    blurred = shg_oracle.box_blur_f64(small, 2, 2)
    vb = np.percentile(blurred, 99)
    stats, counts = ops.flood_stats(dev(small), dev(blurred), vb)
    data = blurred.ravel()[blurred.ravel() < vb]
    want, _ = np.histogram(data, bins=20)
    np.testing.assert_array_equal(host(counts), want)
    np.testing.assert_array_equal(host(stats), [np.sum(small), data.min(), data.max()])
    # very_bright interpolated on the device from the two order statistics np.percentile uses (q = 99 and others)
    from solex_ser_recon_en_amd.order_stats import lerp_gamma, lerp_order_stats
    srt = np.sort(blurred.ravel())
    for q in (99, 50, 12.5, 99.9999, 0, 100):
        lo, hi, mix = lerp_order_stats(srt.size, q)
        vbq = np.percentile(blurred, q)
        assert mix(srt[lo], srt[hi]) == vbq
        s2, c2 = ops.flood_stats_lerp(dev(small), dev(blurred), dev(np.array([srt[lo], srt[hi]])), lerp_gamma(srt.size, q))
        dq = blurred.ravel()[blurred.ravel() < vbq]
        if dq.size:
            np.testing.assert_array_equal(host(c2), np.histogram(dq, bins=20)[0])
            np.testing.assert_array_equal(host(s2), [np.sum(small), dq.min(), dq.max()])


@pytest.mark.parametrize('h,w', [(300, 330), (57, 1025), (1, 9), (260, 3)])
def test_line_order_stats_and_percentile(ops, h, w):
    from solex_ser_recon_en_amd.order_stats import lerp_order_stats
    rng = np.random.default_rng(h * w)
    img = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    img[:, : w // 3] = 513                                  # ties, and one high byte shared by both ranks
    for axis in (0, 1):
        n = img.shape[axis]
        lo, hi, mix = lerp_order_stats(n, 85)
        a, b = ops.line_order_stats_u16(dev(img), axis, lo, hi)
        srt = np.sort(img, axis=axis)
        np.testing.assert_array_equal(host(a), np.take(srt, lo, axis=axis))
        np.testing.assert_array_equal(host(b), np.take(srt, hi, axis=axis))
        got = mix(host(a).astype(np.float64), host(b).astype(np.float64))
        np.testing.assert_array_equal(got, np.percentile(img, 85, axis=axis))


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_remove_vignette_golden(ops, golden, tag):
    from solex_ser_recon_en_amd import solex_util
    g = golden('g6_vignette')
    img = g[tag + '_image']
    out = solex_util.removeVignette(dev(img), tuple(g[tag + '_circle']))
    assert out.dtype == np.float64 and out.shape == img.shape
    np.testing.assert_allclose(out.row_factor.cpu().numpy(), g[tag + '_factor'], rtol=1e-12)
    np.testing.assert_allclose(np.asarray(out)[::17, ::13], g[tag + '_row_sample'], rtol=1e-12)
    small = dev(np.full((90, 90), 1000, np.uint16))
    assert solex_util.removeVignette(small, (45.0, 45.0, 40.0)) is small


def test_row_factor_paths_match_float_image(ops, orc, golden):
    """rowpair statistics / row scaling on the factored float64 frame == the oracle on the materialised one."""
    g = golden('g4_transversalium')
    img = g['image']
    rng = np.random.default_rng(3)
    rf = 1.0 + 0.05 * rng.standard_normal(img.shape[0])
    fimg = img * rf.reshape((-1, 1))
    circle, borders = tuple(g['circle']), list(g['borders'])
    y1, y2, want = orc.transversalium_row_stats(fimg, circle, borders)
    from solex_ser_recon_en_amd.hostmath import chord_bounds
    xa, xb = chord_bounds(circle, borders, y1, y2, img.shape[1])
    got = host(ops.rowpair_logratio_stats(dev(img), y1, y2, xa, xb, rf))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)
    c = g['a_c']
    ret = (fimg.T * c).T
    ret[ret > 65535] = 65535
    np.testing.assert_array_equal(host(ops.scale_rows_u16(dev(img), c, rf)), ret.astype(np.uint16))
    np.testing.assert_array_equal(host(ops.scale_rows_u16(dev(img), np.ones(img.shape[0]), rf)), np.minimum(fimg, 65535).astype(np.uint16))


@pytest.mark.parametrize('h,w', [(7, 5), (9, 8), (33, 17), (64, 2049), (50, 2096), (5, 4100), (130, 1023)])
def test_row_kernels_on_widths_around_their_vector_and_workgroup_sizes(ops, h, w):
    """k_scale_rows8, k_products8 and the CLAHE blend deal their lanes from one flat (row, vector) sequence: widths below a vector,
    just past a multiple of a workgroup's span (2049, 2096, 4100), one short of it (1023), row counts that are not multiples of
    the rows a lane takes -- against plain NumPy, pixel for pixel, padding columns untouched."""
    rng = np.random.default_rng(h * 10007 + w)
    img = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    c = 0.5 + rng.random(h)
    want = np.minimum(img.astype(np.float64) * c[:, None], 65535.0).astype(np.uint16)       # (img.T * c).T, saturate, truncate
    np.testing.assert_array_equal(host(ops.scale_rows_u16(dev(img), c)), want)

    cl1 = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    lo_hi6 = [9000.0, 61000.0, 0.0, 11000.0, 3000.5, 64000.0]

    def rescale(a, lo, hi):                                      # rescale_brightness, solex_util.py:519-524
        v = 65535.0 * (a.astype(np.float64) - lo) / (hi - lo)
        return np.clip(v, 0, 65535).astype(np.uint16)
    disc = (w // 2, h // 2, max(1, min(h, w) // 3))
    hc, protus, cc = (host(t) for t in ops.contrast_products_u16(dev(img), dev(cl1), lo_hi6, disc))
    np.testing.assert_array_equal(hc, rescale(img, lo_hi6[0], lo_hi6[1]))
    np.testing.assert_array_equal(cc, rescale(cl1, lo_hi6[4], lo_hi6[5]))
    want_p = rescale(img, lo_hi6[2], lo_hi6[3])
    yy, xx = np.mgrid[0:h, 0:w]
    ady = np.abs(yy - disc[1])
    half = np.floor(np.sqrt(np.maximum(disc[2] ** 2 - ady.astype(np.float64) ** 2, 0)) + 1e-9).astype(np.int64)
    inside = (ady <= disc[2]) & (np.abs(xx - disc[0]) <= half)
    want_p[inside] = 80                                          # cv2.circle(frame_protus, centre, r, 80, -1)
    np.testing.assert_array_equal(protus, want_p)


@pytest.mark.parametrize('bounds', [
    [0.0, 65535.0, 0.0, 65535.0, 0.0, 65535.0],              # q = px: every quotient whole -- every lane takes the exact division
    [1000.0, 14107.0, 0.0, 13107.0, 2000.0, 15107.0],        # span 13107 = 65535 / 5: q = 5 (px - lo), whole again
    [100.0, 103.0, 0.0, 3.0, 65000.0, 65535.0],              # tiny spans: quotients far beyond 65535, some beyond int32's reach of a fraction
    [16383.75, 65535.0, 0.0, 11796.3, 12345.0, 65535.0],     # the shape image_process gives them (0.25 / 0.18 of the brightest, an integral dark level)
    [0.5, 65534.5, 0.0, 1e-3, 7.0, 8.0],                     # a span of 1e-3: every quotient but 0 saturates the conversion
])
@pytest.mark.parametrize('disc', [None, (40, 30, 25), (-500, 20, 600), (90, 400, 380), (50, 30, 1), (50, 30, 32000)])
def test_products_where_the_short_way_must_not_be_trusted(ops, bounds, disc):
    """k_products8 computes 65535 (px - lo) / span as a product with 1 / span and lets the exact division decide for a lane any of whose
    24 quotients lies within 1e-7 of a whole number (round 6: one decision per lane, not one branch per quotient).  Bounds that make
    EVERY quotient whole, black pixels under lo = 0, pixels equal to lo, quotients beyond 65535 and beyond int32, and discs that
    cover the image, miss it, start left of it or are a single pixel: the three products equal rescale_brightness
    (solex_util.py:519-524) and cv2.circle's fill, pixel for pixel."""
    h, w = 61, 104
    rng = np.random.default_rng(int(bounds[0] * 7 + bounds[3]) & 0xffff)
    img = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    img[::3, ::5] = 0
    img[1::4, 2::7] = 65535
    for k, v in enumerate(bounds):
        img[5 + k, 10:40] = np.uint16(min(max(int(v), 0), 65535))
    cl1 = np.ascontiguousarray(img[::-1, ::-1])

    def rescale(a, lo, hi):
        v = 65535.0 * (a.astype(np.float64) - lo) / (hi - lo)
        return np.clip(v, 0, 65535).astype(np.uint16)
    hc, protus, cc = (host(t) for t in ops.contrast_products_u16(dev(img), dev(cl1), bounds, disc))
    np.testing.assert_array_equal(hc, rescale(img, bounds[0], bounds[1]))
    np.testing.assert_array_equal(cc, rescale(cl1, bounds[4], bounds[5]))
    want_p = rescale(img, bounds[2], bounds[3])
    if disc is not None:
        import math
        yy, xx = np.mgrid[0:h, 0:w]
        ady = np.abs(yy - disc[1])
        half = np.vectorize(lambda d: math.isqrt(max(disc[2] ** 2 - int(d) ** 2, 0)))(ady)
        want_p[(ady <= disc[2]) & (np.abs(xx - disc[0]) <= half)] = 80
    np.testing.assert_array_equal(protus, want_p)


@pytest.mark.parametrize('h,w,tiles', [(96, 104, 2), (130, 200, 2), (257, 2096, 2), (1000, 1048, 2), (90, 120, 3), (77, 50, 2)])
def test_contrast_stats_is_clahe_plus_the_order_statistics(ops, orc, h, w, tiles):
    """shg_contrast_stats_u16 (the first half of image_process in one call): the blend that lays its lanes out in 16 x 16 tiles and counts the
    first select pass on the way, the select histograms zeroed by CLAHE's histogram reduction, the percentiles read off the tile
    histograms or selected -- cl1 equals the oracle's CLAHE and the five statistics are NumPy's order statistics, on shapes that do
    and do not divide into tiles / vectors / workgroup spans."""
    from solex_ser_recon_en_amd.order_stats import lerp_order_stats
    rng = np.random.default_rng(h * 31 + w)
    yy, xx = np.mgrid[0:h, 0:w]
    r = np.hypot(yy - h / 2, xx - w / 2) / (0.45 * min(h, w))
    disc = np.where(r < 1, 0.3 + 0.6 * np.sqrt(np.clip(1 - r * r, 0, 1)), 0.02)
    img = np.clip((disc + 0.01 * rng.standard_normal((h, w))) * 65535, 0, 65535).astype(np.uint16)
    n = h * w
    f_lo, f_hi, _ = lerp_order_stats(n, 99.9999)
    c_lo, c_hi, _ = lerp_order_stats(n, 10)
    out5 = torch.zeros(5, dtype=torch.float64, device='cuda')
    for small in (False, True):
        out5.zero_()
        cl1 = host(ops.contrast_stats_u16(dev(img), [f_lo, f_hi], [c_lo, c_hi, n - 1], out5, 0.8, tiles, small_workspace=small))
        want = orc.clahe(img, 0.8, tiles)
        np.testing.assert_array_equal(cl1, want)
        fs, cs = np.sort(img.ravel()), np.sort(want.ravel())
        np.testing.assert_array_equal(host(out5), np.array([fs[f_lo], fs[f_hi], cs[c_lo], cs[c_hi], cs[n - 1]], dtype=np.float64))


@pytest.mark.parametrize('h,w', [(70, 90), (1, 1), (333, 1027), (2000, 64)])
def test_select_u16_is_exact(ops, h, w):
    rng = np.random.default_rng(h + w)
    img = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    if h * w > 100:
        img[: h // 2] = 40000
    n = h * w
    srt = np.sort(img.ravel())
    ranks = sorted(set([0, n // 10, n // 2, int(0.999999 * (n - 1)), n - 1]))
    for pad in (5, (-w) % 8 + 8):                         # odd pitch: scalar reads; pitch % 8 == 0: 16-byte reads + row tail
        padded = torch.full((h, w + pad), 7, dtype=torch.int16, device='cuda').view(torch.uint16)
        padded[:, :w] = dev(img)                          # a pitched view: the padding must not be counted
        got = host(ops.select_u16(padded[:, :w], ranks))
        np.testing.assert_array_equal(got, srt[ranks].astype(np.float64))
    from solex_ser_recon_en_amd.order_stats import lerp_order_stats
    for q in (10, 99.9999):
        lo, hi, mix = lerp_order_stats(n, q)
        a, b = host(ops.select_u16(dev(img), [lo, hi]))
        assert mix(a, b) == np.percentile(img, q)


@pytest.mark.parametrize('n,h,w,dtype', [(40, 24, 160, np.uint16), (33, 160, 24, np.uint16), (21, 16, 128, np.uint8), (9, 7, 5, np.uint16)])
def test_frame_passes_on_a_pitched_stack(ops, orc, n, h, w, dtype):
    """The stack video_reader.device_stack() builds has a padded frame pitch: both frame passes must ignore the padding."""
    rng = np.random.default_rng(n)
    frames = rng.integers(0, 256 if dtype == np.uint8 else 65536, (n, h, w)).astype(dtype)
    tdt = torch.uint8 if dtype == np.uint8 else torch.uint16
    stack = ops.padded_stack(n, h, w, tdt, 'cuda')
    assert stack.stride(0) * frames.itemsize % 8192 == 0 and stack.stride(0) >= h * w
    torch.as_strided(stack, (n * stack.stride(0),), (1,)).view(torch.uint8).fill_(255)      # poison the padding
    stack.view(torch.int16 if dtype == np.uint16 else torch.uint8)[:] = dev(frames).view(torch.int16 if dtype == np.uint16 else torch.uint8)
    np.testing.assert_array_equal(ops.stack_to_host(stack), frames)
    total, mx = ops.accumulate_sum_max(stack)
    mean, mxo = ops.finalize_mean_max(total, mx, n, h, w, frames.itemsize)
    ref_mean, ref_max = orc.compute_mean_max(orc.SerReader(frames))
    np.testing.assert_array_equal(host(mean), ref_mean)
    np.testing.assert_array_equal(host(mxo), ref_max)
    ih, iw = max(h, w), min(h, w)
    curve = np.clip(iw / 2 + 2 * np.sin(np.arange(ih) / 5.0) + rng.random(ih), 0, iw - 1.5)
    fit = np.stack([np.floor(curve), curve - np.floor(curve), np.arange(ih), curve], axis=1)
    shifts = [1, 0, -1]
    cols, lw, rw = orc.column_indices(fit, shifts, iw)
    disks = ops.extract_columns(stack, np.stack([c[0] for c in cols]).astype(np.int32), lw, rw)
    np.testing.assert_array_equal(host(disks), np.stack(orc.extract_columns(orc.SerReader(frames), fit, shifts)))


@pytest.mark.parametrize('h,w', [(2, 8), (3, 5)])
def test_sum_does_not_overflow_at_70000_full_scale_frames(ops, h, w):
    """More than 65 537 saturated 16-bit frames overflow a u32 sum: the frame axis must be split so that every
    partial fits and the total is exact in u64 (the reference accumulates in uint64, solex_util.py:182)."""
    n = 70000
    stack = torch.full((n, h, w), -1, dtype=torch.int16, device='cuda').view(torch.uint16)      # 65535 everywhere
    total, mx = ops.accumulate_sum_max(stack)
    assert int(total.min()) == int(total.max()) == n * 65535 > 2 ** 32
    mean, mxo = ops.finalize_mean_max(total, mx, n, h, w, 2)
    assert int(mean.view(torch.int16).to(torch.int32).min()) & 0xffff == 65535
    assert int(mxo.view(torch.int16).to(torch.int32).min()) & 0xffff == 65535


def test_single_frame_and_single_pixel_edge_cases(ops, orc):
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 65536, (1, 2, 16)).astype(np.uint16)        # one frame; iw = 2: the only legal column pair is (0, 1)
    total, mx = ops.accumulate_sum_max(dev(frames))
    mean, mxo = ops.finalize_mean_max(total, mx, 1, 2, 16, 2)
    ref_mean, ref_max = orc.compute_mean_max(orc.SerReader(frames))
    np.testing.assert_array_equal(host(mean), ref_mean)
    fit = np.stack([np.full(16, 5.0), np.full(16, 0.25), np.arange(16.0), np.full(16, 5.25)], axis=1)   # far off the 2-px axis: clamps to 0
    cols, lw, rw = orc.column_indices(fit, [10, 0], 2)
    disks = ops.extract_columns(dev(frames), np.stack([c[0] for c in cols]).astype(np.int32), lw, rw)
    np.testing.assert_array_equal(host(disks), np.stack(orc.extract_columns(orc.SerReader(frames), fit, [10, 0])))
    with pytest.raises(RuntimeError):
        ops.extract_columns(dev(rng.integers(0, 9, (2, 1, 8)).astype(np.uint16)), np.zeros((1, 8), np.int32), np.ones(8), np.zeros(8))   # 1-px spectral axis


# ---- stubborn transversalium (line filter) ------------------------------------------------------------
def _flips(got, want):
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    return int(d.max()), int(np.count_nonzero(d))


@pytest.mark.parametrize('case', ['u16', 'f64', 'bb', 'edge_rows', 'zeros'])
def test_lin_filter_vs_oracle_and_reference_shim(ops, orc, golden, case):
    """shg_lin_filter_row_sums + shg_lin_filter_apply through the product's correct_transversalium2 (stubborn
    branch) against the oracle and against the reference's own run (G15, shim mode).  Sums are float64 in one
    fixed order on both sides; what can differ is the device's exp (and log for float64 frames) in the last
    bit, i.e. a truncation flip: <= 1 LSB on <= 8 pixels."""
    from solex_ser_recon_en_amd import solex_util as su
    from solex_ser_recon_en_amd.device import DeviceImage
    g = golden('g15_stubborn')
    img, circle, borders = g['image'].copy(), tuple(g['circle']), list(g['borders'])
    opts = {'stubborn_transversalium': True, 'trans_strength': 301, '_nolog': True, 'clahe_only': True, 'protus_only': False}
    ref_key, rf = 'u16_out', None
    if case == 'f64':
        rf, ref_key = g['row_factor'], 'f64_out'
    elif case == 'bb':
        circle, borders, ref_key = (0, 0, 99999), list(g['bb_borders']), 'bb_out'
        opts['trans_strength'] = 41
    elif case == 'edge_rows':
        ref_key = None
        img[0:3] = (img[0:3] * 1.4).clip(1, 65535).astype(np.uint16)        # does not matter: outside the circle
        circle, borders = (165.0, 149.0, 149.0), [0, 0, 329, 299]            # circle touches the top and bottom rows
    elif case == 'zeros':
        ref_key = None
        img[150, 100:103] = 0                                                # log(0) = -inf inside the windows
    frame = DeviceImage(dev(img), row_factor=dev(rf)) if rf is not None else dev(img)
    got = np.asarray(su.correct_transversalium2(frame, circle, borders, opts, 0, 'x'))
    cpu_img = img * rf[:, None] if rf is not None else img
    with np.errstate(all='ignore'):
        want, flag = orc.correct_transversalium2_stubborn(cpu_img, circle, borders, opts['trans_strength'])
    assert got.dtype == np.uint16 and got.shape == want.shape
    mx, n = _flips(got, want)
    assert mx <= 1 and n <= 8, (mx, n)
    if ref_key:
        mx, n = _flips(got, g[ref_key])
        assert mx <= 1 and n <= 8, (mx, n)
    if case in ('u16', 'bb'):
        assert flag.sum() >= 3 and '_transversalium_cache' not in opts


@pytest.mark.parametrize('n', [255, 256, 257, 513, 1300])
@pytest.mark.parametrize('nsplit', ['1', '2', ''])
def test_u8_sums_across_the_packed_u16_flush(ops, monkeypatch, n, nsplit):
    """8-bit frames are summed in packed u16 pairs that are folded into u32 every 256 frames: saturated and random
    stacks on either side of the fold, with the frame axis in one piece, two, or split by the heuristic."""
    if nsplit:
        monkeypatch.setenv('SHG_ACC_NSPLIT', nsplit)          # the library reads its tuning overrides per call
    rng = np.random.default_rng(n)
    h, w = 3, 32                                                        # 96 B frames: the 16-byte vector path
    sat = torch.full((n, h, w), 255, dtype=torch.uint8, device='cuda')
    total, mx = ops.accumulate_sum_max(sat)
    assert int(total.min()) == int(total.max()) == 255 * n and int(mx.view(torch.int16).min()) == 255
    frames = rng.integers(0, 256, (n, h, w)).astype(np.uint8)
    frames[:, 0, :5] = 255
    total, mx = ops.accumulate_sum_max(dev(frames))
    np.testing.assert_array_equal(host(total).reshape(h, w), frames.sum(axis=0, dtype=np.uint64).astype(np.int64))
    np.testing.assert_array_equal(host(mx.view(torch.int16)).reshape(h, w).astype(np.uint16), frames.max(axis=0))


def test_rowpair_stats_on_a_12000_column_disk(ops, orc):
    """A slow scan (12 000 frames) gives disk rows far longer than 8192 px: the row's keys then take 94 KiB of LDS
    (dynamic allocation above the 64 KiB default)."""
    import math
    rng = np.random.default_rng(12)
    h, w = 12, 12000
    img = rng.integers(20000, 40000, (h, w)).astype(np.uint16)
    circle, borders = (6000.2, 5.5, 5990.0), [3.0, 0, 11990.0, 11]
    y1, y2, want = orc.transversalium_row_stats(img, circle, borders)
    xa = np.zeros(y2 - y1, np.int32)
    xb = np.zeros(y2 - y1, np.int32)
    for y in range(y1 + 1, y2):
        dx = math.floor((circle[2] ** 2 - (y - circle[1]) ** 2) ** 0.5)
        a, b, _ = slice(math.ceil(max(circle[0] - dx, borders[0])), math.floor(min(circle[0] + dx, borders[2]))).indices(w)
        xa[y - y1], xb[y - y1] = a, max(a, b)
    assert (xb - xa).max() > 11000
    got = host(ops.rowpair_logratio_stats(dev(img), y1, y2, dev(xa), dev(xb)))
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)
    with pytest.raises(RuntimeError, match='width'):
        ops.rowpair_logratio_stats(torch.zeros((4, 20000), dtype=torch.int16, device='cuda').view(torch.uint16), 0, 4,
                                   dev(np.zeros(4, np.int32)), dev(np.zeros(4, np.int32)))


@pytest.mark.parametrize('w', [5, 8, 13, 64, 77])
def test_image_passes_respect_view_boundaries(ops, orc, w):
    """rescale / scale_rows / warp on views inside larger images whose neighbouring columns hold sentinels: results
    must equal those on dense copies and the neighbours must stay untouched (source and destination pitch > width)."""
    rng = np.random.default_rng(w)
    h, pitch = 9, 96                                                      # 192-byte rows: aligned, wider than every w
    img = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    big = torch.full((h, pitch), 7, dtype=torch.int16, device='cuda').view(torch.uint16)
    big[:, :w] = dev(img)
    src = big[:, :w]
    dense = dev(np.ascontiguousarray(img[:, :w]))                         # pitch = w
    lo, hi = 1000.0, 60000.0
    want = host(ops.rescale_u16(dense, lo, hi, 1.0))
    np.testing.assert_array_equal(want, orc.rescale_brightness(img, lo, hi))
    np.testing.assert_array_equal(host(ops.rescale_u16(src, lo, hi, 1.0)), want)
    c = 1 + 0.3 * rng.standard_normal(h)
    np.testing.assert_array_equal(host(ops.scale_rows_u16(src, c)), host(ops.scale_rows_u16(dense, c)))
    rf = 1 + 0.1 * rng.standard_normal(h)
    np.testing.assert_array_equal(host(ops.scale_rows_u16(src, c, rf)), host(ops.scale_rows_u16(dense, c, rf)))
    a = host(ops.warp_rows_u16(src, 1.0, 0.0, 0.25, h, w))
    b = host(ops.warp_rows_u16(dense, 1.0, 0.0, 0.25, h, w))
    np.testing.assert_array_equal(a, b)
    # destination views through the C ABI itself (ops always allocates a fresh pitched image)
    from solex_ser_recon_en_amd import _lib
    out_big = torch.full((h, pitch), 9, dtype=torch.int16, device='cuda').view(torch.uint16)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib.shg_rescale_u16(src.data_ptr(), h, w, pitch, lo, hi, 1.0, out_big.data_ptr(), pitch, st), 'shg_rescale_u16')
    np.testing.assert_array_equal(host(out_big[:, :w]), want)
    assert int((out_big[:, w:].view(torch.int16) != 9).sum()) == 0
    cd = dev(c)
    _lib.check(_lib.lib.shg_scale_rows_u16(src.data_ptr(), h, w, pitch, cd.data_ptr(), None, out_big.data_ptr(), pitch, st), 'shg_scale_rows_u16')
    np.testing.assert_array_equal(host(out_big[:, :w]), host(ops.scale_rows_u16(dense, c)))
    assert int((out_big[:, w:].view(torch.int16) != 9).sum()) == 0
    assert int((big[:, w:].view(torch.int16) != 7).sum()) == 0            # nothing leaked out of the source view either


@pytest.mark.parametrize('k,n,window', [(1, 1800, 301), (21, 1800, 301), (3, 280, 279), (2, 40, 21), (1, 5, 5), (4, 302, 301)])
def test_correlate1d_rows_is_scipys(ops, k, n, window):
    """shg_correlate1d_rows_f64 == scipy.ndimage.convolve1d(rows, savgol taps, mode='constant') bit for bit (SciPy's
    symmetric-filter order of operations), also for a non-symmetric filter and with NaN / inf samples."""
    from scipy.ndimage import convolve1d, correlate1d
    from scipy.signal import savgol_coeffs
    rng = np.random.default_rng(n + window)
    rows = rng.standard_normal((k, n)) * 0.01
    taps = savgol_coeffs(window, 3)
    got = host(ops.correlate1d_rows_f64(dev(rows), taps[::-1]))
    np.testing.assert_array_equal(got, convolve1d(rows, taps, axis=-1, mode='constant'))
    anti = savgol_coeffs(window, 3, deriv=1) if window > 3 else np.array([-0.5, 0.0, 0.5, 0.25, -0.25])[:window]
    np.testing.assert_array_equal(host(ops.correlate1d_rows_f64(dev(rows), anti)), correlate1d(rows, anti, axis=-1, mode='constant'))
    skew = rng.standard_normal(window)                                   # neither symmetric nor antisymmetric
    np.testing.assert_array_equal(host(ops.correlate1d_rows_f64(dev(rows), skew)), correlate1d(rows, skew, axis=-1, mode='constant'))
    if n > 20:
        rows[0, 7] = np.nan
        rows[-1, n - 3] = np.inf
        with np.errstate(all='ignore'):
            want = convolve1d(rows, taps, axis=-1, mode='constant')
        np.testing.assert_array_equal(host(ops.correlate1d_rows_f64(dev(rows), taps[::-1])), want)


def test_c_caller_runs_without_python_or_torch(tmp_path):
    """tests/c_abi/abi_smoke.cpp: HIP runtime + libshg_hip.so only (hipMalloc'd buffers, plain pointers and sizes).
    Built with hipcc on the box and run as a child process; exit code 0 = pass A, pass B and a row pass equal CPU loops."""
    import os
    import shutil
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc on this box')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, 'solex_ser_recon_en_amd', 'csrc')
    exe = str(tmp_path / 'abi_smoke')
    b = subprocess.run([hipcc, '--offload-arch=gfx950', '-O1', '-I', os.path.join(root, 'include'),
                        os.path.join(root, 'tests', 'c_abi', 'abi_smoke.cpp'), '-L', lib_dir, '-lshg_hip',
                        '-Wl,-rpath,' + lib_dir, '-o', exe], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    assert 'C ABI smoke OK' in r.stdout


@pytest.mark.parametrize('levels', [1, 2, 3, 17])
def test_rowpair_stats_with_heavily_tied_ratios(ops, orc, levels):
    """Rows quantised to a few grey levels: the log-ratios repeat, so the radix select cannot stop early on a
    singleton bin and the two middle order statistics often share every digit (median / MAD of tied data,
    MAD = 0 -> every sample is an inlier, solex_util.py:81-86)."""
    import math
    rng = np.random.default_rng(levels)
    h, w = 40, 300
    img = (1000 + rng.integers(0, levels, (h, w)) * 7).astype(np.uint16)
    circle, borders = (150.2, 19.6, 148.0), [2.0, 0, 297.0, 39]
    y1, y2, want = orc.transversalium_row_stats(img, circle, borders)
    xa = np.zeros(y2 - y1, np.int32)
    xb = np.zeros(y2 - y1, np.int32)
    for y in range(y1 + 1, y2):
        dx = math.floor((circle[2] ** 2 - (y - circle[1]) ** 2) ** 0.5)
        a, b, _ = slice(math.ceil(max(circle[0] - dx, borders[0])), math.floor(min(circle[0] + dx, borders[2]))).indices(w)
        xa[y - y1], xb[y - y1] = a, max(a, b)
    got = host(ops.rowpair_logratio_stats(dev(img), y1, y2, dev(xa), dev(xb)))
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)


def test_rowpair_stats_row_by_row_over_the_shapes_the_bucket_select_has_to_survive(ops, orc):
    """One row pair per shape, every one against np.mean(reject_outliers(np.log(row / row_before))) (solex_util.py:81-86, 384-395):
    chords of 1 to 2600 pixels on either side of a workgroup's 256 threads and of a wave's eight buckets a lane; rows whose ratios
    are all equal (sigma 0: the radix select), nearly all equal (every value in one bucket: the bucket select's second level),
    bimodal (the middle ranks' bucket far from the mean), with one wild pixel (sigma set by an outlier), with zero pixels (-inf, +inf
    and NaN ratios) and all black; round 6's rough float sums and one-wave bucket scan must steer the select, never change it."""
    rng = np.random.default_rng(66)
    w = 2700
    shapes = []
    for n in (1, 2, 3, 7, 8, 9, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1025, 2047, 2049, 2600):
        shapes.append(('noise', n))
    for kind in ('flat', 'nearly_flat', 'bimodal', 'wild', 'zeros', 'black', 'ramp', 'two_values'):
        for n in (5, 300, 1800):
            shapes.append((kind, n))
    h = len(shapes) + 1
    img = np.empty((h, w), np.uint16)
    img[0] = rng.integers(9000, 11000, w)
    xa = np.zeros(h, np.int32)
    xb = np.zeros(h, np.int32)
    for t, (kind, n) in enumerate(shapes, start=1):
        prev = img[t - 1].astype(np.int64)
        prev = np.where(prev == 0, 10000, prev)
        a = int(rng.integers(0, w - n + 1))
        xa[t], xb[t] = a, a + n
        row = (prev * (1 + 0.01 * rng.standard_normal(w))).clip(1, 65535)
        if kind == 'flat':
            row = prev * 2
        elif kind == 'nearly_flat':
            row = prev * 2
            row[a + n // 2] += 1
        elif kind == 'bimodal':
            row = np.where(rng.random(w) < 0.5, prev * 0.5, prev * 1.9)
        elif kind == 'wild':
            row[a + n // 3] = 65535 if prev[a + n // 3] < 30000 else 1
        elif kind == 'zeros':
            row[a:a + n:4] = 0
        elif kind == 'black':
            row[:] = 0
        elif kind == 'ramp':
            row = prev + np.arange(w) % 7
        elif kind == 'two_values':
            row = np.where(np.arange(w) % 3 == 0, prev, prev + 1)
        img[t] = np.clip(row, 0, 65535).astype(np.uint16)
    with np.errstate(all='ignore'):
        want = np.array([0.0] + [np.mean(orc.reject_outliers(np.log(img[t, xa[t]:xb[t]] / img[t - 1, xa[t]:xb[t]]))) for t in range(1, h)])
    got = host(ops.rowpair_logratio_stats(dev(img), 0, h, xa, xb))
    assert np.array_equal(np.isnan(got), np.isnan(want)), (np.flatnonzero(np.isnan(got) != np.isnan(want)), [shapes[i - 1] for i in np.flatnonzero(np.isnan(got) != np.isnan(want))])
    ok = ~np.isnan(want)
    bad = np.flatnonzero(ok & ~np.isclose(got, want, rtol=1e-12, atol=1e-15, equal_nan=True) & ~(np.isinf(want) & (got == want)))
    assert bad.size == 0, [(shapes[i - 1], got[i], want[i]) for i in bad[:8]]


# ---- the limb stage's fused kernels against the one-kernel-per-call chain (csrc/limb_fused.hip vs csrc/limb.hip) ----------
def _limb_disk(h, w, seed):
    """A disk-like uint16 image [h, w]: a bright ellipse with limb darkening, sky, noise, a few saturated pixels."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    r2 = ((xx - 0.52 * w) / (0.40 * w)) ** 2 + ((yy - 0.49 * h) / (0.43 * h)) ** 2
    img = np.where(r2 < 1, 9000 + 30000 * np.sqrt(np.clip(1 - r2, 0, 1)), 900.0)
    img = img + rng.normal(0, 250, img.shape)
    if seed >= 100:                                          # a flat top: the 99th percentile of the blurred image is a tie
        img = np.minimum(img, 30000.0)
    else:
        img[rng.integers(0, h, 40), rng.integers(0, w, 40)] = 65535
    return np.clip(img, 0, 65535).astype(np.uint16)


@pytest.mark.parametrize('fence', ['0', '1'])
@pytest.mark.parametrize('h,w,seed', [(2000, 2000, 1), (1203, 997, 2), (3204, 1601, 3), (6400, 800, 4), (420, 640, 5), (2000, 2000, 101), (1203, 997, 102)])
def test_fused_limb_kernels_equal_the_separate_ones(ops, h, w, seed, fence, monkeypatch):
    """shg_limb_prepare == downscale + cv2.blur (k and 5) + the two selects + the flood statistics, and shg_limb_edges ==
    canny masks + hysteresis labelling, value for value: same window sums, same order statistics, same sum / min / max /
    histogram, same edge pixels with the same roots -- for blur windows 5, 3, 8, 16 and 1, images whose sides are not
    multiples of 4 or of the tile, and every rung of canny's retry ladder.  Seeds >= 100: a disk with a flat top, so that the two
    order statistics the 99th percentile is interpolated between coincide -- the case in which the fused path has to look the image
    over for the largest value below very_bright instead of taking the lower order statistic.
    fence = 1: SHG_LIMB_FENCE=1, the last-workgroup counters bumped with an acq_rel agent-scope read-modify-write (the memory model's
    own way) instead of the relaxed one behind published(): the fallback must give the same values."""
    monkeypatch.setenv('SHG_LIMB_FENCE', fence)
    disk = torch.from_numpy(_limb_disk(h, w, seed)).cuda()
    sh, sw = -(-h // 4), -(-w // 4)
    n = sh * sw
    k = int(sh * 0.01)
    assert k >= 1
    from solex_ser_recon_en_amd import hostmath
    ranks = [n // 2 if n & 1 else n // 2 - 1, n // 2]
    lo, hi, gamma = hostmath.percentile_plan(n, 99.0)
    ranks += [lo, hi]
    # the separate kernels
    small = ops.downscale_mean_u16(disk, 4)
    blurred, keys_k = ops.box_blur_key_f64(small, k)
    _, keys_5 = (blurred, keys_k) if k == 5 else ops.box_blur_key_f64(small, 5)
    want_os = ops.select_keys_u32([keys_5, keys_5, keys_k, keys_k], ranks, [5, 5, k, k]).cpu().numpy()
    stats, counts = ops.flood_stats_lerp(small, blurred, torch.from_numpy(want_os[2:4].copy()).cuda(), gamma)
    # the fused ones
    packed, keys, ws = ops.limb_prepare(disk, k, ranks, gamma)
    torch.cuda.synchronize()
    got = packed.cpu().numpy()
    np.testing.assert_array_equal(keys.cpu().numpy(), keys_k.cpu().numpy())
    np.testing.assert_array_equal(got[0:4], want_os)
    np.testing.assert_array_equal(got[4:7], stats.cpu().numpy())
    np.testing.assert_array_equal(got[8:18].view(np.uint32), counts.cpu().numpy().view(np.uint32))
    # canny + labelling, every rung of the ladder
    counts64 = counts.cpu().numpy().astype(np.int64)
    thresh = hostmath.flood_threshold(float(got[4]), (sh, sw), float(got[5]), float(got[6]), counts64)
    median5 = got[0] if n & 1 else (got[0] + got[1]) / 2
    low, high = median5 / 10, median5 / 10 * 1.5
    for sigma in (2.0, 1.5, 1.0, 0.5):
        lm, hm = ops.canny_masks(blurred, thresh, sigma, low, high)
        want_idx, want_root = ops.edge_components(lm, hm, prefetch=n)
        idx, root, strong = ops.limb_edges(keys, k, thresh, sigma, low, high)
        flat_low = np.flatnonzero(lm.cpu().numpy().reshape(-1))
        np.testing.assert_array_equal(idx, flat_low)                              # every low pixel, raster order
        np.testing.assert_array_equal(strong, hm.cpu().numpy().reshape(-1)[idx].astype(bool))
        keep = np.isin(root, np.unique(root[strong]))                              # the hysteresis the stage does on the host
        np.testing.assert_array_equal(idx[keep], want_idx)
        np.testing.assert_array_equal(root[keep], want_root)
        assert len(want_idx) > 0


@pytest.mark.parametrize('shape', [(200, 304), (202, 310), (64, 2096)])
def test_blend_with_eight_pixel_lanes_equals_four_pixel_lanes(shape, monkeypatch):
    """From four disks up the CLAHE blend (k_clahe_interp_vm, clahe_apply.py:243-256's interpolation) gives a lane eight pixels --
    16-byte loads and stores -- instead of four.  Same products bit for bit as with four-pixel lanes (SHG_INTERP_SHAPE=34: the
    single-disk shape), on a width that is a multiple of eight, one that is not (a row's last vector goes pixel by pixel), and a
    full-width one."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from solex_ser_recon_en_amd import stages
    h, w = shape
    k = 5
    rng = np.random.default_rng(h * 7 + w)
    yy, xx = np.mgrid[0:h, 0:w]
    pitch = (w + 63) // 64 * 64
    store = torch.zeros((k, h, pitch), dtype=torch.uint16, device='cuda')
    views = []
    for i in range(k):
        r = np.hypot((yy - h / 2) / (0.45 * h), (xx - w / 2) / (0.45 * w))
        img = np.where(r < 1, 50000.0 * (0.4 + 0.6 * np.sqrt(np.clip(1 - r * r, 0, 1))), 800.0 + 40 * i)
        img = np.clip(img + rng.normal(0, 400, img.shape), 1, 65535).astype(np.uint16)
        store[i, :, :w] = torch.from_numpy(img.view(np.int16)).cuda().view(torch.uint16)
        views.append(store[i, :, :w])

    def run(shape_env):
        if shape_env is None:
            monkeypatch.delenv('SHG_INTERP_SHAPE', raising=False)
        else:
            monkeypatch.setenv('SHG_INTERP_SHAPE', shape_env)
        res = stages.process_frames(views, None, None, (w // 2, h // 2, int(0.3 * h)))
        torch.cuda.synchronize()
        return {name: [np.asarray(t.cpu().view(torch.int16).numpy()).view(np.uint16).copy() for t in res[name]] for name in ('cl1', 'hc', 'protus', 'cc')}
    wide, narrow = run(None), run('34')
    for name in wide:
        for x, y in zip(wide[name], narrow[name]):
            np.testing.assert_array_equal(x, y, err_msg=name)


@pytest.mark.parametrize('shape,crop,trans,k', [((200, 304), None, True, 1), ((200, 304), (256, 24, 0, 256), True, 3),
                                                ((202, 310), (400, 0, 45, 310), True, 2), ((200, 304), None, False, 2),
                                                ((198, 306), (198, 54, 0, 198), False, 1), ((2000, 2096), None, True, 2),
                                                ((2000, 2096), (2000, 48, 0, 2000), True, 1), ((201, 304), None, True, 1)])
def test_scaling_and_crop_fused_into_the_histogram_kernel(shape, crop, trans, k, monkeypatch):
    """shg_stage_process_frames forms the image CLAHE works on inside the CLAHE histogram kernel (k_tile_hist16_slices<true>: frame x
    row factor, saturate, truncate, crop / pad; Solex_recon.py:149-171, solex_util.py:515-516).  Every product equals the separate
    kernels' (SHG_FUSE_SCALE=0: k_scale_rows8, k_crop_pad, then the histograms) bit for bit: vector and pixel paths, crops that pad
    on either side, several disks, no transversalium (a plain copy), and a shape the grid does not divide (201 rows: the border's
    pixels formed a second time for the count)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from solex_ser_recon_en_amd import solex_util, stages
    h, w = shape
    rng = np.random.default_rng(h * 31 + w)
    yy, xx = np.mgrid[0:h, 0:w]
    frames = []
    for i in range(k):
        r = np.hypot((yy - h / 2) / (0.45 * h), (xx - w / 2) / (0.45 * w))
        img = np.where(r < 1, 52000.0 * (0.4 + 0.6 * np.sqrt(np.clip(1 - r * r, 0, 1))), 900.0) * (1 + 0.02 * np.sin(yy / 3.0 + i))
        img = np.clip(img + rng.normal(0, 300, img.shape), 1, 65535).astype(np.uint16)
        img[0, 0] = 777 + i                                  # the padding value is img[0, 0] (times its row's factor)
        frames.append(torch.from_numpy(img.view(np.int16)).cuda().view(torch.uint16))
    pitch = (w + 63) // 64 * 64
    store = torch.zeros((k, h, pitch), dtype=torch.uint16, device='cuda')
    views = []
    for i, f in enumerate(frames):
        store[i, :, :w] = f
        views.append(store[i, :, :w])
    transv = None
    if trans:
        window = min(61, (int(0.8 * h) // 2) * 2 - 1)
        transv = dict(circle=(w / 2.0, h / 2.0, 0.4 * h), borders=[0.0, 0.05 * h, w - 1.0, 0.95 * h], taps=solex_util.savgol_taps(window), window=window)

    def run(fused):
        monkeypatch.setenv('SHG_FUSE_SCALE', '1' if fused else '0')
        res = stages.process_frames(views, transv, crop, (w // 2, h // 2, int(0.3 * h)))
        torch.cuda.synchronize()
        return {name: [np.asarray(t.cpu().view(torch.int16).numpy()).view(np.uint16).copy() for t in res[name]]
                for name in ('final', 'cl1', 'hc', 'protus', 'cc')}, None if res['factors'] is None else res['factors'].copy()
    a, fa = run(True)
    b, fb = run(False)
    if trans:
        np.testing.assert_array_equal(fa, fb)
    for name in a:
        for x, y in zip(a[name], b[name]):
            assert x.shape == (h, crop[0] if crop else w)
            np.testing.assert_array_equal(x, y, err_msg=name)


def test_limb_stage_under_load_equals_the_separate_kernels(ops, monkeypatch):
    """The fused limb kernels find their last workgroup with a RELAXED agent-scope counter behind `s_waitcnt vmcnt(0)` + barrier
    (limb_fused.hip, published()): correct because gfx950 performs agent-scope atomics at the memory side and acknowledges them
    afterwards -- an assumption about this part, not a promise of the memory model, so it is held here under the load it has to
    survive: four streams run 2000 limb stages between them on four different disks while a fifth keeps pass A (1.6 GB of
    non-temporal reads per launch) going; every one of the 2000 results -- geometry, limb points, kept points -- equals what the
    one-kernel-per-call chain of limb.hip (SHG_LIMB_FUSED=0, acq_rel-free as well but without any last-workgroup logic) gives
    for that disk, bit for bit."""
    import threading
    from solex_ser_recon_en_amd import stages, synth
    shapes = [(2000, 2000, 11), (2000, 1777, 12), (1603, 2100, 13), (2404, 1500, 14)]
    disks = [torch.from_numpy(_limb_disk(h, w, seed)).cuda() for h, w, seed in shapes]
    monkeypatch.setenv('SHG_LIMB_FUSED', '0')
    want = [stages.limb_fit(d, want_points=True) for d in disks]
    monkeypatch.setenv('SHG_LIMB_FUSED', '1')
    stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=3, padded=True)
    torch.cuda.synchronize()
    stop = threading.Event()
    failures = []

    def frame_pass():
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                ws = None
                while not stop.is_set():
                    for _ in range(4):
                        ops.accumulate_sum_max(stack, ws)
                    st.synchronize()
        except BaseException as e:      # noqa: BLE001
            failures.append('pass A: %r' % (e,))

    def same(a, b):
        if isinstance(a, np.ndarray):
            return a.shape == b.shape and np.array_equal(a, b)
        if isinstance(a, (list, tuple)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return a == b

    def limb_worker(t):
        try:
            torch.cuda.set_device(0)
            stages.use_buffers({})
            with torch.cuda.stream(torch.cuda.Stream()):
                for i in range(500):
                    j = (t + i) % len(disks)
                    got = stages.limb_fit(disks[j], want_points=True)
                    for key, ref in want[j].items():
                        if not same(got[key], ref):
                            failures.append('stream %d, stage %d, disk %d: %s differs' % (t, i, j, key))
                            return
        except BaseException as e:      # noqa: BLE001
            failures.append('stream %d: %r' % (t, e))

    bg = threading.Thread(target=frame_pass)
    bg.start()
    workers = [threading.Thread(target=limb_worker, args=(t,)) for t in range(4)]
    for w_ in workers:
        w_.start()
    for w_ in workers:
        w_.join()
    stop.set()
    bg.join()
    torch.cuda.synchronize()
    assert not failures, failures[:5]


def _solar_image(h, w, seed, flat_top=False, sky=0.01):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    r = np.hypot(yy - h / 2, xx - w / 2) / (0.42 * min(h, w))
    disc = np.where(r < 1, 0.35 + 0.55 * np.sqrt(np.clip(1 - r * r, 0, 1)), sky)
    img = np.clip((disc + 0.01 * rng.standard_normal((h, w))) * 65535, 0, 65535).astype(np.uint16)
    img[r >= 1] = int(sky * 65535)                      # a perfectly flat sky: one bin holds most of every slice (clamped at every level)
    if flat_top:
        img[img > 55000] = 65535                         # a burnt-out core: the top order statistics sit in ONE bin with thousands of pixels
    return img


@pytest.mark.parametrize('h,w,tiles,clip_limit', [(2000, 2096, 2, 0.8), (2000, 2098, 2, 0.8), (2560, 2676, 2, 0.8), (1100, 1000, 2, 0.8),
                                                  (901, 1203, 2, 3.0), (640, 960, 4, 8.0), (700, 1056, 3, 2.0), (1100, 1008, 2, 15.9),
                                                  (1100, 1008, 2, 16.0), (1100, 1008, 2, 59.5), (1100, 1008, 2, 61.0), (64, 64, 2, 0.8)])
def test_clahe_saturated_slice_histograms_equal_the_u16_ones(ops, orc, h, w, tiles, clip_limit, monkeypatch):
    """csrc/clahe.hip, k_tile_hist16_slices<., 8 / 4>: a slice's counters clamped to the clip limit and stored as bytes or nibbles
    (min(a + b, c) = min(min(a, c) + b, c)) must give the image the whole u16 counters give (SHG_CLAHE_SAT=0), bit for bit, and the
    oracle's where that is quick: clip limits 12 (nibbles), 20, 15 | 16 (the nibble / byte border), 255 | 261 (byte / u16 border), a grid
    that does not divide the image (reflected tiles), one-chunk slices and slices of several chunks clamped in between
    (SHG_CLAHE_SAT_PX = 300 000: four or five chunks of < 65 280 pixels), bytes forced where nibbles would do (SHG_CLAHE_SAT=8).
    The sky is perfectly flat: one bin takes most of a slice and is clamped at every level."""
    img = _solar_image(h, w, seed=h + w)
    d = dev(img)
    monkeypatch.setenv('SHG_CLAHE_SAT', '0')
    want = host(ops.clahe(d, clip_limit, tiles))
    if h * w <= 1_300_000:
        np.testing.assert_array_equal(want, orc.clahe(img, clip_limit, tiles))
    for mode, px in (('1', None), ('8', None), ('1', '300000'), ('8', '300000'), ('1', '5000')):
        monkeypatch.setenv('SHG_CLAHE_SAT', mode)
        if px is None:
            monkeypatch.delenv('SHG_CLAHE_SAT_PX', raising=False)
        else:
            monkeypatch.setenv('SHG_CLAHE_SAT_PX', px)
        np.testing.assert_array_equal(host(ops.clahe(d, clip_limit, tiles)), want, err_msg='SHG_CLAHE_SAT=%s px=%s' % (mode, px))


@pytest.mark.parametrize('shape,k,flat_top', [((2000, 2096), 1, False), ((2000, 2096), 3, True), ((2560, 2676), 2, False), ((2000, 2098), 5, True),
                                              ((200, 304), 2, False), ((1000, 1048), 4, True)])
def test_frame_percentile_off_the_saturated_histograms(shape, k, flat_top, monkeypatch):
    """shg_stage_process_frames reads np.percentile(frame, 99.9999)'s two order statistics off CLAHE's tile histograms; with clamped
    counters (k_hist_reduce_sat, hist_rank_top_job) they are found from the TOP, which is exact while they lie within `clip` pixels of
    it.  Every product -- the contrast bounds depend on those statistics -- must equal the u16-counter route's (SHG_CLAHE_SAT=0):
    1 to 5 disks (5: the stack's larger slices, several chunks each), a burnt-out core (the statistics inside one bin of thousands of
    pixels), a 6.8 Mpx image (clip 20, the 7th and 8th largest), a small image (clip 1: the statistics are out of reach, the u16 route
    must be taken by itself) and a 1 Mpx one (clip 3, the 2nd and 3rd largest)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from solex_ser_recon_en_amd import stages
    h, w = shape
    pitch = (w + 63) // 64 * 64
    store = torch.zeros((k, h, pitch), dtype=torch.uint16, device='cuda')
    views = []
    for i in range(k):
        store[i, :, :w] = torch.from_numpy(_solar_image(h, w, seed=100 + i, flat_top=flat_top and i % 2 == 0).view(np.int16)).cuda().view(torch.uint16)
        views.append(store[i, :, :w])

    def run(mode):
        monkeypatch.setenv('SHG_CLAHE_SAT', mode)
        res = stages.process_frames(views, None, None, (w // 2, h // 2, int(0.3 * h)))
        torch.cuda.synchronize()
        return {name: [np.asarray(t.cpu().view(torch.int16).numpy()).view(np.uint16).copy() for t in res[name]] for name in ('final', 'cl1', 'hc', 'protus', 'cc')}
    want = run('0')
    for mode in ('1', '8'):
        got = run(mode)
        for name in want:
            for i, (x, y) in enumerate(zip(got[name], want[name])):
                np.testing.assert_array_equal(x, y, err_msg='%s[%d] SHG_CLAHE_SAT=%s' % (name, i, mode))


@pytest.mark.parametrize('shape,k,crop,trans,streak', [((2000, 2097), 1, None, True, None), ((2000, 2097), 3, None, True, 'row'), ((2001, 2096), 2, None, False, None),
                                                       ((2001, 2097), 1, None, True, 'column'), ((200, 305), 2, None, True, None),
                                                       ((1000, 1049), 4, None, False, 'row'), ((2000, 2096), 2, (1999, 48, 0, 1999), True, None),
                                                       ((600, 1001), 2, (1101, 0, 50, 1001), True, 'column'), ((999, 1203), 5, None, True, None)])
def test_contrast_stage_on_a_tile_grid_that_does_not_divide_the_image(shape, k, crop, trans, streak, monkeypatch):
    """cv2.createCLAHE pads an image its tile grid does not divide below and to the right with its mirror image (REFLECT_101; by a whole
    `tiles` along an axis that does divide when the other does not) and the tiles count those pixels (solex_util.py:532-533).  The
    batched contrast stage takes such images too (half of all scans: the width of a circularised disk is any number): the histogram
    kernel counts the border pixels without storing them, and np.percentile(frame, 99.9999)'s order statistics, read off the tile
    histograms, leave them out again (BorderPx) -- with saturated counters only while that can be told, else the stage selects over
    the image (a streak of equal, brightest pixels in the row / column the border mirrors forces that).  Every product must equal
    what the disk-by-disk route gives (SHG_CONTRAST_BATCH=0: shg_contrast_stats_u16, held against the oracle elsewhere), with u16,
    byte and nibble counters: odd widths, odd heights, both, a crop to an odd width, a small image (clip 1), five disks."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from solex_ser_recon_en_amd import solex_util, stages
    h, w = shape
    pitch = (w + 63) // 64 * 64
    store = torch.zeros((k, h, pitch), dtype=torch.uint16, device='cuda')
    views = []
    out_w = crop[0] if crop else w
    for i in range(k):
        img = _solar_image(h, w, seed=300 + i)
        if streak == 'row' and i % 2 == 0:
            img[h - 2, w // 3:w // 3 + 40] = 65000          # (row h - 2 is the first row of the border)
        if streak == 'column' and i % 2 == 0:
            c = (crop[1] + min(crop[3], crop[0] - crop[2]) - 2) if crop else w - 2         # the image's last column but one
            img[h // 3:h // 3 + 40, c] = 64000
        store[i, :, :w] = torch.from_numpy(img.view(np.int16)).cuda().view(torch.uint16)
        views.append(store[i, :, :w])
    transv = None
    if trans:
        window = min(61, (int(0.8 * h) // 2) * 2 - 1)
        transv = dict(circle=(w / 2.0, h / 2.0, 0.4 * h), borders=[0.0, 0.05 * h, w - 1.0, 0.95 * h], taps=solex_util.savgol_taps(window), window=window)

    def run(batch, sat):
        monkeypatch.setenv('SHG_CONTRAST_BATCH', batch)
        monkeypatch.setenv('SHG_CLAHE_SAT', sat)
        res = stages.process_frames(views, transv, crop, (out_w // 2, h // 2, int(0.3 * h)))
        torch.cuda.synchronize()
        return {name: [np.asarray(t.cpu().view(torch.int16).numpy()).view(np.uint16).copy() for t in res[name]] for name in ('final', 'cl1', 'hc', 'protus', 'cc')}
    def reselected():                                        # how often the stage fell back on the select over the image since enable(1)
        import ctypes
        from solex_ser_recon_en_amd import _lib
        buf = ctypes.create_string_buffer(1 << 14)
        _lib.lib.shg_host_timing_report(buf, len(buf))
        return sum(int(line.rsplit(' ', 1)[1]) for line in buf.value.decode().splitlines() if line.startswith('frame percentile selected'))
    from solex_ser_recon_en_amd import _lib
    want = run('0', '1')
    for sat in ('1', '0', '8'):
        _lib.lib.shg_host_timing_enable(1)
        try:
            got = run('1', sat)
            fell_back = reselected()
        finally:
            _lib.lib.shg_host_timing_enable(0)
        for name in want:
            for i, (x, y) in enumerate(zip(got[name], want[name])):
                assert x.shape == (h, out_w)
                np.testing.assert_array_equal(x, y, err_msg='%s[%d] SHG_CLAHE_SAT=%s' % (name, i, sat))
        # the streak's 40 equal pixels and their 40 mirror images clamp the top bin: saturated counters cannot tell -- every other disk asks again
        he, we = (h, out_w) if h % 2 == 0 and out_w % 2 == 0 else (h + 2 - h % 2, out_w + 2 - out_w % 2)
        clip = max(int(0.8 * (he // 2) * (we // 2) / 65536), 1)
        k_top = h * out_w - int((h * out_w - 1) * 0.999999)         # the lower of np.percentile's two order statistics, counted from the top
        if streak == 'row' and sat != '0' and k_top <= clip <= 255:
            assert fell_back == (k + 1) // 2, (fell_back, clip, k_top)
        if streak is None:
            assert fell_back == 0


def test_percentile_window_hits_and_misses_give_the_second_pass_results(monkeypatch):
    """np.percentile(cl1, 10) and np.max(cl1) without a second pass over cl1 (csrc/clahe.hip, SelWin): the blend kernel counts the low
    bytes of the eight high bytes around the PREVIOUS scan's answer -- kept in the stage's workspace -- and an atomic maximum; when
    this scan's ranks fall inside, k_select16_pass returns at once.  A run of scans through one workspace -- the same image again
    (a hit), a much brighter and a much darker one (misses: the window sits where the last image had its 10th percentile), back
    again, disks of a stack with different levels in one launch -- must give exactly what SHG_SELECT_WINDOW=0 gives."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from solex_ser_recon_en_amd import stages
    h, w = 1000, 1048
    pitch = (w + 63) // 64 * 64

    def images(levels):
        store = torch.zeros((len(levels), h, pitch), dtype=torch.uint16, device='cuda')
        views = []
        for i, (seed, sky, gain) in enumerate(levels):
            img = _solar_image(h, w, seed=seed, sky=sky).astype(np.float64) * gain
            store[i, :, :w] = torch.from_numpy(np.clip(img, 0, 65535).astype(np.uint16).view(np.int16)).cuda().view(torch.uint16)
            views.append(store[i, :, :w])
        return views
    series = [images([(1, 0.01, 1.0)]), images([(1, 0.01, 1.0)]), images([(2, 0.30, 1.0)]), images([(3, 0.001, 0.2)]), images([(1, 0.01, 1.0)]),
              images([(4, 0.01, 1.0), (5, 0.25, 0.9), (6, 0.002, 0.3)]), images([(4, 0.01, 1.0), (5, 0.25, 0.9), (6, 0.002, 0.3)])]

    def run(mode):
        monkeypatch.setenv('SHG_SELECT_WINDOW', mode)          # (1: also for a single disk)
        out = []
        for views in series:
            res = stages.process_frames(views, None, None, (w // 2, h // 2, int(0.3 * h)))
            torch.cuda.synchronize()
            out.append({name: [np.asarray(t.cpu().view(torch.int16).numpy()).view(np.uint16).copy() for t in res[name]] for name in ('cl1', 'hc', 'protus', 'cc')})
        return out
    want = run('0')
    got = run('1')
    for k, (a, b) in enumerate(zip(got, want)):
        for name in b:
            for i, (x, y) in enumerate(zip(a[name], b[name])):
                np.testing.assert_array_equal(x, y, err_msg='scan %d %s[%d]' % (k, name, i))
