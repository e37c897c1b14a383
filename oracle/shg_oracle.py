"""CPU oracle for the SHG reconstruction hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference algorithm
(thelondonsmiths/Solex_ser_recon_EN, snapshot 2025-10-03).  Each function cites
the reference file:line it follows.  It exists to *check* the HIP product path:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it.  Nothing under solex_ser_recon_en_amd/ imports it, and the product raises if
its HIP library is missing -- there is no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * SER decode, mean/max, column extraction, correction matrix, warp,
    transversalium, rescale, crop: PINNED -- tests/golden/*.npz were produced by
    running the reference's own unmodified functions (oracle/capture_goldens.py,
    /opt/conda python3.9, NumPy 1.26, SciPy 1.7, scikit-image 0.18.3).
  * line fit (compute_mean_return_fit): pinned in SHIM MODE -- the reference
    function ran unmodified with cv2.blur replaced by box_blur() below, so all
    logic around the blur is pinned, the blur primitive itself is not.
  * cv2.blur, cv2.createCLAHE, cv2.circle, lsq-ellipse: PARITY UNPINNED -- those
    third-party packages are absent from /root/reference and from this image and
    are unpinned upstream (requirements.txt:2-10).  The functions below restate
    their published algorithms (OpenCV 4.x box_filter / clahe.cpp / drawing.cpp,
    Halir & Flusser 1998) and are checked by known-answer tests only.

Written to run under both NumPy 1.26 (capture, shim mode) and NumPy 2.x.
"""
import math

import numpy as np
from numpy.polynomial.polynomial import polyval

SER_HEADER_BYTES = 178


# ----------------------------------------------------------------------------
# a1  SER decode                                         video_reader.py:10-126
# ----------------------------------------------------------------------------
class SerReader:
    """Frame iterator with the attributes the reference's video_reader exposes.

    Source is either a .ser path or an in-memory array [N, Height, Width] in file
    layout (the same thing the file holds after its 178-byte header)."""

    def __init__(self, source, k0=0, k1=None):
        if isinstance(source, str):
            with open(source, 'rb') as f:
                head = f.read(SER_HEADER_BYTES)
            # Width @26, Height @30, PixelDepthPerPlane @34, FrameCount @38 (video_reader.py:43-54)
            self.Width, self.Height, depth, self.FrameCount = (int(v) for v in np.frombuffer(head, '<u4', 4, 26))
            self.infilebytes = 1 if depth == 8 else 2          # video_reader.py:56-63
            dt = np.uint8 if depth == 8 else np.dtype('<u2')
            self._frames = np.memmap(source, dtype=dt, mode='r', offset=SER_HEADER_BYTES,
                                     shape=(self.FrameCount, self.Height, self.Width))
        else:
            self._frames = source
            self.FrameCount, self.Height, self.Width = (int(v) for v in source.shape)
            self.infilebytes = source.dtype.itemsize
        self.infiledatatype = 'uint8' if self.infilebytes == 1 else 'uint16'
        self.count = self.Width * self.Height
        self.flag_rotate = self.Width > self.Height             # video_reader.py:84-91
        self.ih, self.iw = (self.Width, self.Height) if self.flag_rotate else (self.Height, self.Width)
        self._k0 = k0
        self._k1 = self.FrameCount if k1 is None else k1
        self.FrameIndex = k0 - 1                                # video_reader.py:64

    def has_frames(self):                                       # video_reader.py:125-126
        return self.FrameIndex + 1 < self._k1

    def next_frame(self):                                       # video_reader.py:94-123
        self.FrameIndex += 1
        img = np.asarray(self._frames[self.FrameIndex])
        if self.flag_rotate:
            img = np.rot90(img)                                 # img[y, x] = raw[x, Width-1-y]
        if self.infilebytes == 1:
            img = img.astype(np.uint16) * 256                   # video_reader.py:121-122
        return img


class AviReader:
    """The AVI branch of the reference's video_reader (video_reader.py:20-23, 68-80, 111-113):
    cv2.VideoCapture(file).read() -> BGR frame -> cv2.cvtColor(COLOR_BGR2GRAY) -> rot90 / x256 as for SER.
    UNPINNED: cv2 is absent.  Restated for UNCOMPRESSED streams only, from the AVI RIFF layout: the frames are
    located through the file's 'idx1' index (the product walks the 'movi' chunks instead), decoded to the BGR
    image VideoCapture would hand back, and converted with OpenCV 4's 8-bit BGR2GRAY
    (B*3735 + G*19235 + R*9798 + 2^14) >> 15."""

    def __init__(self, path):
        import struct
        data = open(path, 'rb').read()
        assert data[:4] == b'RIFF' and data[8:12] == b'AVI '
        # stream headers in file order; the first 'vids' stream is the one VideoCapture decodes
        kinds, pos = [], 0
        while True:
            pos = data.find(b'strh', pos)
            if pos < 0:
                break
            kinds.append(data[pos + 8:pos + 12])
            pos += 4
        vid = kinds.index(b'vids')
        strf = data.find(b'strf', data.find(b'vids'))
        strf_size = struct.unpack_from('<I', data, strf + 4)[0]
        bi = strf + 8
        bi_size, w, h, _, bits, comp = struct.unpack_from('<IiiHH4s', data, bi)
        self.Width, self.Height = w, abs(h)
        raw_rgb = comp == b'\0\0\0\0'
        if not (raw_rgb and bits in (8, 24)) and not (comp in (b'Y800', b'Y8  ', b'GREY') and bits == 8):
            raise Exception('compressed AVI')
        flip = raw_rgb and h > 0                              # a DIB with positive height stores the bottom row first
        row = (w * bits + 31) // 32 * 4 if raw_rgb else w
        palette = None
        if raw_rgb and bits == 8 and strf_size > bi_size:
            palette = np.frombuffer(data, np.uint8, (strf_size - bi_size) // 4 * 4, bi + bi_size).reshape(-1, 4)[:, :3]
        movi = data.find(b'movi')
        idx = data.rfind(b'idx1')
        n_idx = struct.unpack_from('<I', data, idx + 4)[0] // 16
        self._bgr = []
        for e in range(n_idx):
            ckid, _flags, off, size = struct.unpack_from('<4sIII', data, idx + 8 + 16 * e)
            if ckid[2:4] not in (b'db', b'dc') or int(ckid[:2]) != vid or size == 0:
                continue
            at = movi + off if data[movi + off:movi + off + 4] == ckid else off       # offsets from 'movi' (usual) or absolute
            rows = np.frombuffer(data, np.uint8, row * self.Height, at + 8).reshape(self.Height, row)
            if flip:
                rows = rows[::-1]
            if bits == 24:
                bgr = rows[:, :3 * w].reshape(self.Height, w, 3)
            else:
                index = rows[:, :w]
                bgr = palette[index] if palette is not None else np.stack([index] * 3, axis=-1)
            self._bgr.append(bgr)
        self.FrameCount = len(self._bgr)
        self.infilebytes, self.infiledatatype = 1, 'uint8'      # video_reader.py:23, 77
        self.count = self.Width * self.Height
        self.flag_rotate = self.Width > self.Height
        self.ih, self.iw = (self.Width, self.Height) if self.flag_rotate else (self.Height, self.Width)
        self.FrameIndex = -1

    def has_frames(self):
        return self.FrameIndex + 1 < self.FrameCount

    @staticmethod
    def bgr2gray(bgr):
        b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
        return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)

    def next_frame(self):
        self.FrameIndex += 1
        img = self.bgr2gray(self._bgr[self.FrameIndex])         # video_reader.py:112-113
        img = np.reshape(img, (self.Height, self.Width))
        if self.flag_rotate:
            img = np.rot90(img)
        return np.asarray(img, dtype='uint16') * 256            # video_reader.py:121-122

    def raw_frames(self):
        """[N, Height, Width] uint8 grey frames in file orientation (what the GPU stack holds)."""
        return np.stack([self.bgr2gray(f) for f in self._bgr])


# ----------------------------------------------------------------------------
# a2  mean and max frames                                  solex_util.py:174-188
# ----------------------------------------------------------------------------
def compute_mean_max(rdr):
    acc = np.zeros((rdr.ih, rdr.iw), dtype=np.uint64)
    mx = np.zeros((rdr.ih, rdr.iw), dtype=np.uint16)
    while rdr.has_frames():
        img = rdr.next_frame()
        acc += img
        np.maximum(mx, img, out=mx)
    # float64 true division then truncation (solex_util.py:188)
    return (acc / np.uint32(rdr.FrameCount)).astype(np.uint16), mx


# ----------------------------------------------------------------------------
# cv2.blur restatement (UNPINNED)     call sites solex_util.py:166, 230
# ----------------------------------------------------------------------------
def _window_sums(img, kw, kh, dtype):
    """Box sums with anchor (kw//2, kh//2) and BORDER_REFLECT_101 (OpenCV default)."""
    left, top = kw // 2, kh // 2
    pad = np.pad(img.astype(dtype), ((top, kh - 1 - top), (left, kw - 1 - left)), mode='reflect')
    c = np.cumsum(np.cumsum(pad, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    h, w = img.shape
    return c[kh:kh + h, kw:kw + w] - c[0:h, kw:kw + w] - c[kh:kh + h, 0:w] + c[0:h, 0:w]


def box_blur_u16(img, kw, kh, simd_lanes=8):
    """cv2.blur(img_u16, ksize=(kw, kh)).

    OpenCV's ColumnSum<int, ushort> (box_filter.simd.hpp) scales the exact integer
    window sum by 1/(kw*kh): in float32 (`v_round(v_cvt_f32(sum) * (float)scale)`)
    for the SIMD lanes, i.e. all columns below the last multiple of the vector width, and in
    double (`saturate_cast<ushort>(sum * scale)`) for the scalar tail columns.  Both round
    half to even.  simd_lanes: uint16 lanes of the vector unit OpenCV was dispatched to --
    8 for the 128-bit baseline (SSE2 / NEON), 16 for an AVX2 build, 32 for AVX-512; 0 = no SIMD
    (every column scaled in double).  The HIP kernel uses 8; tests/test_oracle_golden.py shows
    that everything derived from the blurred images on this path (sunlit rows, both line traces,
    the fit) is the same for every width."""
    assert img.dtype == np.uint16 and kw >= 1 and kh >= 1
    s = _window_sums(img, kw, kh, np.int64)
    scale = 1.0 / (kw * kh)
    out = np.rint(s.astype(np.float32) * np.float32(scale)).astype(np.int64)
    w = img.shape[1]
    tail = (w // simd_lanes) * simd_lanes if simd_lanes else 0
    if tail < w:
        out[:, tail:] = np.rint(s[:, tail:].astype(np.float64) * scale).astype(np.int64)
    return np.clip(out, 0, 65535).astype(np.uint16)


def box_blur_f64(img, kw, kh):
    """cv2.blur on a float64 image (ellipse_to_circle.py:163, 241): the window is summed along the row
    (left to right), the row sums along the column (top to bottom), then scaled by 1/(kw*kh).
    OpenCV's RowSum/ColumnSum keep sliding running sums instead, equal up to float64 rounding;
    the primitive is unpinned either way and this order is the one the HIP kernel uses."""
    img = np.asarray(img, dtype=np.float64)
    h, w = img.shape
    left, top = kw // 2, kh // 2
    pad = np.pad(img, ((top, kh - 1 - top), (left, kw - 1 - left)), mode='reflect')
    rows = np.zeros((pad.shape[0], w))
    for j in range(kw):
        rows = rows + pad[:, j:j + w]
    out = np.zeros((h, w))
    for j in range(kh):
        out = out + rows[j:j + h]
    return out * (1.0 / (kw * kh))


# ----------------------------------------------------------------------------
# a3  sunlit row range                                     solex_util.py:165-172
# ----------------------------------------------------------------------------
def detect_bord(img, axis=1, blur=box_blur_u16):
    b = blur(img, 5, 5)
    ymean = np.mean(b, axis)
    where_sun = ymean > np.median(ymean) / 5
    lb = int(np.argmax(where_sun))
    ub = int(img.shape[int(not axis)] - 1 - np.argmax(np.flip(where_sun)))
    return lb, ub


# ----------------------------------------------------------------------------
# a4  spectral-line detection + cubic fit                  solex_util.py:191-274
# ----------------------------------------------------------------------------
def line_fit(mean_img, max_img, blur=box_blur_u16):
    """Numerics of compute_mean_return_fit after the mean/max pass.
    Returns (fit[ih,4], y1, y2, p, aux) with aux = intermediate row traces."""
    ih, iw = mean_img.shape
    y1, y2 = detect_bord(max_img, axis=1, blur=blur)            # :223
    clip = int((y2 - y1) * 0.05)
    y1 = min(max_img.shape[0] - 1, y1 + clip)
    y2 = max(0, y2 - clip)
    bwx = 25
    bwy = int((y2 - y1) * 0.01)                                 # :229
    blurred = blur(mean_img, bwx, bwy)
    # columns 12 .. iw-14, note -25//2 == -13                   :231
    min_intensity = bwx // 2 + np.argmin(blurred[:, bwx // 2:-bwx // 2], axis=1)
    rows = np.arange(y1, y2)
    rows_d = np.asarray(rows, dtype='d')
    p = np.flip(np.asarray(np.polyfit(rows, min_intensity[y1:y2], 3), dtype='d'))        # :233
    delta = polyval(rows_d, p) - min_intensity[y1:y2]
    keep = np.abs(delta / np.std(delta)) < 3
    p = np.flip(np.asarray(np.polyfit(rows[keep], min_intensity[y1:y2][keep], 3), dtype='d'))   # :238
    sharp = np.argmin(mean_img, axis=1)                         # :242
    delta_sharp = polyval(rows_d, p) - sharp[y1:y2]
    values, counts = np.unique(np.around(delta_sharp, 1), return_counts=True)
    ind = np.argpartition(-counts, kth=2)[:2]                   # :246 (needs >= 3 unique values)
    shift = values[ind[0]]
    mask_good = np.abs(delta_sharp - shift) < 5                 # :253-254
    p = np.flip(np.asarray(np.polyfit(rows[mask_good], sharp[y1:y2][mask_good], 3), dtype='d'))
    curve = polyval(np.asarray(np.arange(ih), dtype='d'), p)
    fit = np.array([[math.floor(curve[y]), curve[y] - math.floor(curve[y]), y, curve[y]] for y in range(ih)])
    aux = dict(min_intensity=min_intensity, sharp=sharp, mask_good=mask_good, blurred=blurred)
    return fit, y1, y2, p, aux


# ----------------------------------------------------------------------------
# a5  per-frame column extraction                          solex_util.py:93-144
# ----------------------------------------------------------------------------
def shift_list(ellipse_fit_shift, requested):
    """options['shift'] after Solex_recon.py:55."""
    return list(dict.fromkeys([ellipse_fit_shift, 0] + list(requested)))


def column_indices(fit, shifts, iw):
    """Clamped left/right sample columns per shift and the unclamped weights (:113-123)."""
    ih = fit.shape[0]
    cols = []
    for shift in shifts:
        ind_l = (np.asarray(fit)[:, 0] + np.ones(ih) * shift).astype(int)
        ind_l[ind_l < 0] = 0
        ind_l[ind_l > iw - 2] = iw - 2
        ind_r = (ind_l + np.ones(ih)).astype(int)
        cols.append((ind_l, ind_r))
    left_w = np.ones(ih) - np.asarray(fit)[:, 1]
    right_w = np.ones(ih) - left_w
    return cols, left_w, right_w


def extract_columns(rdr, fit, shifts):
    ih, iw = rdr.ih, rdr.iw
    cols, left_w, right_w = column_indices(fit, shifts, iw)
    disks = [np.zeros((ih, rdr.FrameCount), dtype=np.uint16) for _ in shifts]
    rows = np.arange(ih)
    while rdr.has_frames():
        img = rdr.next_frame()
        for i, (ind_l, ind_r) in enumerate(cols):
            # float64: two rounded products, one rounded add, truncating store (:131-134)
            disks[i][:, rdr.FrameIndex] = img[rows, ind_l] * left_w + img[rows, ind_r] * right_w
    return disks


# ----------------------------------------------------------------------------
# a6  correction matrix                               ellipse_to_circle.py:35-50
# ----------------------------------------------------------------------------
def _rot(x):
    return np.array([[np.cos(x), np.sin(x)], [-np.sin(x), np.cos(x)]])


def correction_matrix(phi, r):
    stretch = _rot(phi) @ np.array([[r, 0], [0, 1]]) @ _rot(-phi)
    theta = np.arctan(stretch[1, 0] / stretch[0, 0])
    corr = _rot(theta) @ stretch
    corr[1, 0] = 0
    corr /= corr[1, 1]
    return np.linalg.inv(corr), theta


# ----------------------------------------------------------------------------
# a7  the warp                                       ellipse_to_circle.py:94-145
# ----------------------------------------------------------------------------
def warp_rows(image, mat3, out_shape, cval):
    """skimage.transform.warp(image, ProjectiveTransform(mat3), output_shape, cval=cval)
    for order 1 / mode 'constant' / clip=True, restated from scikit-image 0.18.3
    (_warp_fast + bilinear_interpolation + _clip_warp_output).  Requires the last
    two rows of mat3 to be [0,1,0] and [0,0,1] (the correction matrix never moves
    rows, ellipse_to_circle.py:48-49), so each output row is a 1-D linear resample
    of the same input row."""
    assert mat3[1, 0] == 0 and mat3[1, 1] == 1 and mat3[1, 2] == 0
    assert mat3[2, 0] == 0 and mat3[2, 1] == 0 and mat3[2, 2] == 1
    rows_in, cols_in = image.shape
    out_r, out_c = int(out_shape[0]), int(out_shape[1])
    image = np.asarray(image, dtype=np.float64)
    out = np.empty((out_r, out_c), dtype=np.float64)
    c = np.arange(out_c, dtype=np.float64)
    for r in range(out_r):
        x = mat3[0, 0] * c + mat3[0, 1] * float(r) + mat3[0, 2]
        x0 = np.floor(x)
        x1 = np.ceil(x)
        dc = x - x0
        # rows past the input sample cval on both sides; the blend below is still evaluated
        row = image[r] if r < rows_in else np.full(cols_in, cval)
        i0 = x0.astype(np.int64)
        i1 = x1.astype(np.int64)
        left = np.where((i0 >= 0) & (i0 < cols_in), row[np.clip(i0, 0, cols_in - 1)], cval)
        right = np.where((i1 >= 0) & (i1 < cols_in), row[np.clip(i1, 0, cols_in - 1)], cval)
        out[r] = (1 - dc) * left + dc * right
    np.clip(out, image.min(), image.max(), out=out)
    return out


def correct_image(image, phi, ratio, center, height):
    """Returns (uint16 image, (cx, cy, r), mat3)."""
    mat, theta = correction_matrix(phi, ratio)
    mat3 = np.zeros((3, 3))
    mat3[:2, :2] = mat
    mat3[2, 2] = 1
    h, w = image.shape
    corners = np.array([[0, 0], [0, h], [w, 0], [w, h]])
    new_corners = (np.linalg.inv(mat) @ corners.T).T
    new_h = np.max(new_corners[:, 1]) - np.min(new_corners[:, 1])
    new_w = np.max(new_corners[:, 0]) - np.min(new_corners[:, 0])
    mat3 = mat3 @ np.array([[1, 0, np.min(new_corners[:, 0])], [0, 1, np.min(new_corners[:, 1])], [0, 0, 1]])
    warped = warp_rows(image, mat3, (np.ceil(new_h), np.ceil(new_w)), image[0, 0])
    out = (2 ** 16 * warped).astype(np.uint16)
    new_center = (np.linalg.inv(mat) @ np.asarray(center).T).T - np.array(
        [np.min(new_corners[:, 0]), np.min(new_corners[:, 1])])
    new_radius = height * np.sqrt(np.abs(ratio / np.linalg.det(mat)))
    return out, (new_center[0], new_center[1], new_radius), mat3


# ----------------------------------------------------------------------------
# a9  transversalium                               solex_util.py:76-86, 383-516
# ----------------------------------------------------------------------------
def reject_outliers(data, m=2):
    med = np.median(data)
    d = np.abs(data - med)
    mdev = np.median(d)
    s = d / mdev if mdev else np.zeros(len(d))
    return data[s < m]


def tukey_taper(n, a=0.05):                                     # solex_util.py:456-470
    def t(x):
        if 0 <= x < a * n / 2:
            return 1 / 2 * (1 - math.cos(2 * math.pi * x / (a * n)))
        elif a * n / 2 <= x <= n / 2:
            return 1
        elif n / 2 <= x <= n:
            return t(n - x)
        return 1
    return np.array([t(x) for x in range(n)])


def transversalium_row_stats(img, circle, borders):
    """Per-row-pair robust mean of log(row[y]/row[y-1]) inside circle & borders (:384-395)."""
    y1 = math.ceil(max(circle[1] - circle[2], borders[1]))
    y2 = math.floor(min(circle[1] + circle[2], borders[3]))
    ratios = [0]
    for y in range(y1 + 1, y2):
        dx = math.floor((circle[2] ** 2 - (y - circle[1]) ** 2) ** 0.5)
        a = math.ceil(max(circle[0] - dx, borders[0]))
        b = math.floor(min(circle[0] + dx, borders[2]))
        with np.errstate(divide='ignore', invalid='ignore'):
            rat = np.log(img[y, a:b] / img[y - 1, a:b])
        ratios.append(np.mean(reject_outliers(rat)))
    return y1, y2, np.array(ratios)


def transversalium_factors(ratios, n_rows, y1, y2, trans_strength=301):
    from scipy.signal import savgol_filter
    trend = savgol_filter(ratios, min(trans_strength, len(ratios) // 2 * 2 - 1), 3)     # :400
    detrended = ratios - trend
    detrended -= np.mean(detrended)
    correction = np.exp(-np.cumsum(detrended))
    n = correction.shape[0]
    corr_t = np.ones(n) + (correction - np.ones(n)) * tukey_taper(n)
    c = np.ones(n_rows)
    c[y1:y2] = corr_t
    return c


def correct_transversalium2(img, circle, borders, trans_strength=301):
    y1, y2, ratios = transversalium_row_stats(img, circle, borders)
    c = transversalium_factors(ratios, img.shape[0], y1, y2, trans_strength)
    ret = (img.T * c).T
    ret[ret > 65535] = 65535
    return np.array(ret, dtype='uint16'), c


# ----------------------------------------------------------------------------
# f4  stubborn transversalium                     solex_util.py:277-375, 415-423
#     cv2.filter2D is restated (UNPINNED: opencv-python is absent here and unpinned upstream).  OpenCV runs
#     kernels this large (11x101, 1x101) through a DFT whose rounding cannot be reproduced; the restatement
#     is the exact correlation with BORDER_REFLECT_101, summed in float64 and rounded once to the image's
#     float type -- the value the DFT approximates.
# ----------------------------------------------------------------------------
def row_box_sums_reflect101(a, k):
    """S[y, x] = sum_{d = -(k//2) .. k//2} a[y, reflect101(x + d)], float64, added left to right."""
    half = k // 2
    w = a.shape[1]
    p = np.pad(np.asarray(a, dtype=np.float64), ((0, 0), (half, half)), mode='reflect')
    acc = p[:, 0:w].copy()
    for d in range(1, k):
        acc += p[:, d:d + w]
    return acc


def spurious_rows(correction, n_rows, y1, y2):
    """Rows whose accumulated correction is an outlier, dilated by one row with wrap-around (:416-421)."""
    lc = np.log(correction)
    c = np.zeros(n_rows)
    c[y1:y2] = lc
    flag = np.abs(c) > np.std(lc) * 2.5
    return np.logical_or(flag, np.logical_or(np.roll(flag, -1), np.roll(flag, 1)))


def neighbour_rows(flag):
    """For every row the nearest unflagged row above and below (-1: none), the two passes of :306-317."""
    n = len(flag)
    up = np.full(n, -1, dtype=np.int32)
    dn = np.full(n, -1, dtype=np.int32)
    last = -1
    for i in range(n):
        if flag[i]:
            up[i] = last
        else:
            last = i
    last = -1
    for i in range(n - 1, -1, -1):
        if flag[i]:
            dn[i] = last
        else:
            last = i
    return up, dn


def edge_plan(circle, h, w, linlen):
    """fix_edge_effect (:356-375) as per-row data: delta is kept on [xa, xb), copied from column xa+half into
    [xa, xa+half) when left[y], from column xb-half-1 into [xb-half, xb) when right[y], zero elsewhere."""
    y1 = math.ceil(max(circle[1] - circle[2], 0))
    y2 = math.floor(min(circle[1] + circle[2], h - 1))
    half = linlen // 2
    xa = np.zeros(h, dtype=np.int32)
    xb = np.zeros(h, dtype=np.int32)
    left = np.zeros(h, dtype=bool)
    right = np.zeros(h, dtype=bool)
    for y in range(max(y1, 0), y2):
        dx = math.floor((circle[2] ** 2 - (y - circle[1]) ** 2) ** 0.5)
        x2 = math.floor(min(circle[0] + dx, w - 1))
        x1 = math.ceil(max(circle[0] - dx, 0))
        xa[y], xb[y] = x1, max(x2, x1)
        if x2 - x1 < linlen:
            continue
        left[y] = x1 > 0
        right[y] = x2 < w - 1
    if 0 <= y2 < h and y2 >= y1:
        xa[y2], xb[y2] = 0, w                     # the loop stops before row y2 and only rows > y2 are cleared
    return xa, xb, left, right, half


def apply_lin_filter(img, flag, y1, y2, circle, linlen=101, half_width=5):
    """img * exp(-delta * taper): delta = (1 x linlen mean of log img) - (mean over the 2*half_width neighbouring
    rows x linlen columns of log img with the flagged rows replaced by the mean of their nearest unflagged
    neighbours), cleaned at the limb by fix_edge_effect with linlen + 20 (:277-353).
    np.log of a uint16 image is float32, of a float64 image float64: the filters keep that type."""
    h, w = img.shape
    with np.errstate(divide='ignore', invalid='ignore'):
        L = np.log(img)
    ftype = L.dtype
    up, dn = neighbour_rows(flag)
    filt2 = L.copy()
    zero = np.zeros(w, dtype=ftype)
    with np.errstate(invalid='ignore'):
        for i in np.nonzero(flag)[0]:
            a = L[up[i]] if up[i] >= 0 else zero
            b = L[dn[i]] if dn[i] >= 0 else zero
            filt2[i] = a / 2
            filt2[i] += b / 2
        hf = row_box_sums_reflect101(filt2, linlen)
        hl = row_box_sums_reflect101(L, linlen)
        pf = np.pad(hf, ((half_width, half_width), (0, 0)), mode='reflect')
        acc = np.zeros((h, w))
        for d in range(2 * half_width + 1):
            if d != half_width:
                acc += pf[d:d + h]
        result3 = (acc * (1.0 / (2 * half_width * linlen))).astype(ftype)
        result4 = (hl * (1.0 / linlen)).astype(ftype)
        delta = result4 - result3
        xa, xb, left, right, half = edge_plan(circle, h, w, linlen + 20)
        fixed = np.zeros_like(delta)
        for y in range(h):
            a, b = int(xa[y]), int(xb[y])
            fixed[y, a:b] = delta[y, a:b]
            if left[y]:
                fixed[y, a:a + half] = delta[y, a + half]
            if right[y]:
                fixed[y, b - half:b] = delta[y, b - half - 1]
        n = y2 - y1
        c = np.zeros(h)
        c[y1:y2] = tukey_taper(n)
        return img * np.exp(-fixed * c.reshape(-1, 1))


def correct_transversalium2_stubborn(img, circle, borders, trans_strength=301):
    """The `stubborn_transversalium` branch of correct_transversalium2 (:415-423)."""
    from scipy.signal import savgol_filter
    y1, y2, ratios = transversalium_row_stats(img, circle, borders)
    trend = savgol_filter(ratios, min(trans_strength, len(ratios) // 2 * 2 - 1), 3)
    detrended = ratios - trend
    detrended -= np.mean(detrended)
    correction = np.exp(-np.cumsum(detrended))
    flag = spurious_rows(correction, img.shape[0], y1, y2)
    with np.errstate(invalid='ignore'):
        out = apply_lin_filter(img, flag, y1, y2, circle)
        return np.minimum(out, 65535).astype('uint16'), flag


# ----------------------------------------------------------------------------
# removeVignette                                             solex_util.py:590-654
# ----------------------------------------------------------------------------
def remove_vignette(frame, cercle0):
    """Returns the float64 image frame * correction_factor[:, None] (or `frame` itself when there is
    not enough data), exactly as the reference: 85th percentiles per column / row, Savitzky-Golay
    trends inside the disk, ratio of the two axes as the row correction, NaN fill, Gaussian smoothing."""
    from scipy.ndimage import gaussian_filter1d
    from scipy.signal import savgol_filter
    y_arr = np.percentile(frame, 85, axis=0)
    y_arr2 = np.percentile(frame, 85, axis=1)
    shrink = 65
    start1 = max(0, int(cercle0[0] - cercle0[2] + shrink))
    end1 = min(y_arr.shape[0], int(cercle0[0] + cercle0[2] + 1 - shrink))
    start2 = max(0, int(cercle0[1] - cercle0[2] + shrink))
    end2 = min(y_arr2.shape[0], int(cercle0[1] + cercle0[2] + 1 - shrink))
    y1 = y_arr[start1:end1]
    y2 = y_arr2[start2:end2]
    x1 = np.arange(y1.shape[0]) + start1 - int(cercle0[0])
    x2 = np.arange(y2.shape[0]) + start2 - int(cercle0[1])
    if y1.shape[0] < 20 or y2.shape[0] < 20:
        return frame
    scale_pix = int(min(y1.shape[0] // 2.75, y2.shape[0] // 2.75)) // 2 * 2 - 1
    trend1 = savgol_filter(y1, min(801, scale_pix), 3)
    trend2 = savgol_filter(y2, min(801, scale_pix), 3)
    mm = min(np.min(x1), np.min(x2))
    dest = np.zeros((3, int(max(np.max(x1), np.max(x2)) - mm + 1)))
    dest.fill(np.nan)
    dest[0, :] = np.arange(dest.shape[1]) + mm
    dest[1, int(x1[0] - mm): int(x1[-1] - mm + 1)] = trend1
    dest[2, int(x2[0] - mm): int(x2[-1] - mm + 1)] = trend2
    with np.errstate(divide='ignore', invalid='ignore'):
        ratio_axes = dest[1, :] / dest[2, :]
    ratio_axes[dest[1, :] == 0] = np.nan
    ratio_axes[dest[2, :] == 0] = np.nan
    correction_factor = np.zeros(frame.shape[0])
    correction_factor.fill(np.nan)
    correction_factor[dest[0, :].astype(int) + int(cercle0[1])] = ratio_axes
    for i in range(1, len(correction_factor)):
        if np.isnan(correction_factor[i]):
            correction_factor[i] = correction_factor[i - 1]
    for i in range(len(correction_factor) - 2, -1, -1):
        if np.isnan(correction_factor[i]):
            correction_factor[i] = correction_factor[i + 1]
    correction_factor = gaussian_filter1d(correction_factor, max(2, min(150, scale_pix // 4)))
    return frame * correction_factor.reshape((-1, 1))


# ----------------------------------------------------------------------------
# a10  crop / pad                                        Solex_recon.py:155-171
# ----------------------------------------------------------------------------
def crop_center(img, cercle, fixed_width, crop_width_square):
    if fixed_width is None and not crop_width_square:
        return img, cercle
    h, w = img.shape
    nw = h if fixed_width is None else fixed_width
    nw2 = nw // 2
    cx = w // 2 if cercle == (-1, -1, -1) else int(cercle[0])
    tx = nw2 - cx
    new_img = np.full((h, nw), img[0, 0], dtype=img.dtype)
    lo, hi = max(0, cx - nw2), min(cx + nw2, w)
    new_img[:, :hi - lo] = img[:, lo:hi]
    if tx > 0:
        new_img = np.roll(new_img, tx, axis=1)
        new_img[:, :tx] = img[0, 0]
    if not cercle == (-1, -1, -1):
        cercle = (nw2, cercle[1], cercle[2])
    return new_img, cercle


# ----------------------------------------------------------------------------
# a11/a13  CLAHE (cv2.createCLAHE restatement, UNPINNED)  solex_util.py:532-533
# ----------------------------------------------------------------------------
def clahe(img, clip_limit=0.8, tiles=2):
    """cv2.createCLAHE(clipLimit, (tiles, tiles)).apply(img) for uint8 / uint16,
    restated from OpenCV 4.x imgproc/clahe.cpp (CLAHE_CalcLut_Body,
    CLAHE_Interpolation_Body).  All interpolation arithmetic is float32, unfused."""
    assert img.dtype in (np.uint8, np.uint16) and img.ndim == 2
    hist_size = 256 if img.dtype == np.uint8 else 65536
    h, w = img.shape
    if w % tiles == 0 and h % tiles == 0:
        src = img
    else:
        # copyMakeBorder(0, tiles - h%tiles, 0, tiles - w%tiles, BORDER_REFLECT_101)
        src = np.pad(img, ((0, tiles - h % tiles), (0, tiles - w % tiles)), mode='reflect')
    th, tw = src.shape[0] // tiles, src.shape[1] // tiles
    area = th * tw
    lut_scale = np.float32(hist_size - 1) / np.float32(area)
    clip = 0
    if clip_limit > 0.0:
        clip = max(int(clip_limit * area / hist_size), 1)
    luts = np.empty((tiles, tiles, hist_size), dtype=np.float32)
    for ty in range(tiles):
        for tx in range(tiles):
            tile = src[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            hist = np.bincount(tile.ravel(), minlength=hist_size).astype(np.int64)
            if clip > 0:
                clipped = int(np.sum(np.maximum(hist - clip, 0)))
                hist = np.minimum(hist, clip)
                batch = clipped // hist_size
                residual = clipped - batch * hist_size
                hist += batch
                if residual != 0:
                    step = max(hist_size // residual, 1)
                    idx = np.arange(0, hist_size, step)[:residual]
                    hist[idx] += 1
            cs = np.cumsum(hist)
            lut = np.rint(cs.astype(np.float32) * lut_scale)     # saturate_cast<T>(sum * lutScale)
            luts[ty, tx] = np.clip(lut, 0, hist_size - 1)
    inv_tw = np.float32(1.0) / np.float32(tw)
    inv_th = np.float32(1.0) / np.float32(th)

    def axis_terms(n, inv_t):
        tf = np.arange(n, dtype=np.float32) * inv_t - np.float32(0.5)
        t1 = np.floor(tf).astype(np.int64)
        a = tf - t1.astype(np.float32)
        a1 = np.float32(1.0) - a
        return np.maximum(t1, 0), np.minimum(t1 + 1, tiles - 1), a.astype(np.float32), a1.astype(np.float32)

    tx1, tx2, xa, xa1 = axis_terms(w, inv_tw)
    ty1, ty2, ya, ya1 = axis_terms(h, inv_th)
    v = img.astype(np.int64)
    out = np.empty((h, w), dtype=img.dtype)
    for y in range(h):
        l11 = luts[ty1[y], tx1, v[y]]
        l12 = luts[ty1[y], tx2, v[y]]
        l21 = luts[ty2[y], tx1, v[y]]
        l22 = luts[ty2[y], tx2, v[y]]
        res = (l11 * xa1 + l12 * xa) * ya1[y] + (l21 * xa1 + l22 * xa) * ya[y]
        out[y] = np.clip(np.rint(res), 0, hist_size - 1).astype(img.dtype)
    return out


# ----------------------------------------------------------------------------
# a12  rescale_brightness                                  solex_util.py:519-525
# ----------------------------------------------------------------------------
def rescale_brightness(img, lo, hi, alpha=1.0):
    sat = np.iinfo(img.dtype).max
    assert sat >= hi > lo
    rescaled = float(sat) * alpha * (img - lo) / (hi - lo)
    rescaled[rescaled < 0] = 0
    rescaled[rescaled > sat] = sat
    return rescaled.astype(img.dtype)


# ----------------------------------------------------------------------------
# cv2.circle(img, (x0, y0), r, color, -1) restatement (UNPINNED) solex_util.py:547
# ----------------------------------------------------------------------------
def filled_circle(img, x0, y0, radius, color):
    """Integer midpoint circle, filled (OpenCV drawing.cpp Circle(), fill=true)."""
    h, w = img.shape
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1

    def hline(y, xa, xb):
        if 0 <= y < h:
            xa, xb = max(xa, 0), min(xb, w - 1)
            if xa <= xb:
                img[y, xa:xb + 1] = color

    while dx >= dy:
        hline(y0 - dy, x0 - dx, x0 + dx)
        hline(y0 + dy, x0 - dx, x0 + dx)
        hline(y0 - dx, x0 - dy, x0 + dy)
        hline(y0 + dx, x0 - dy, x0 + dy)
        dy += 1
        err += plus
        plus += 2
        mask = -1 if err > 0 else 0       # (err <= 0) - 1
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    return img


# ----------------------------------------------------------------------------
# a11  image_process numerics                              solex_util.py:527-553
# ----------------------------------------------------------------------------
def image_process(frame, cercle, disk_display=True, delta_radius=0, img_rotate=0):
    """Returns dict(cl1, cc, protus, hc, raw) -- the arrays the reference writes."""
    frame = frame.astype(np.uint16)
    cl1 = clahe(frame, 0.8, 2)
    bright = np.percentile(frame, 99.9999)
    dark_clahe = np.percentile(cl1, 10)
    bright_clahe = np.max(cl1)
    hc = rescale_brightness(frame, bright * 0.25, bright)
    protus = rescale_brightness(frame, 0, bright * 0.18)
    cc = rescale_brightness(cl1, dark_clahe, bright_clahe)
    if not cercle == (-1, -1, -1) and disk_display:
        r = int(cercle[2]) + delta_radius
        if r > 0:
            protus = filled_circle(protus, int(cercle[0]), int(cercle[1]), r, 80)
    k = img_rotate // 90
    return dict(cl1=cl1, raw=np.rot90(frame, k), hc=np.rot90(hc, k), protus=np.rot90(protus, k),
                cc=np.rot90(cc, k))
