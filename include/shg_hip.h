/*
 * shg_hip.h -- C ABI of libshg_hip.so, the MI355X (gfx950) implementation of the
 * SHG reconstruction hot path of thelondonsmiths/Solex_ser_recon_EN.
 *
 * The reference is pure Python and has no FFI; the seam is its set of Python
 * functions.  Each entry point below replaces the NumPy /
 * OpenCV / scikit-image call sites named in its comment (reference file:line).
 * The only consumer is a ctypes binding (solex_ser_recon_en_amd/_lib.py); the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name starts with host_;
 *  - the caller owns every image, workspace and staging buffer: the kernel entry points, the stage composites and
 *    shg_scan_file allocate nothing on the device and keep nothing between calls;
 *  - `stream` is a hipStream_t (0 = default stream); all calls are asynchronous
 *    with respect to that stream and re-entrant (any number of threads, each with
 *    its own stream and buffers);
 *  - process-global state, all of it: (1) the per-thread error string; (2) the frame-pass lane of each device
 *    (shg_frame_pass_lane_set: one stream handle per device, set once) and a pair of events per (thread, device) that
 *    uses it; (3) the registry of passes launched ahead of their scans (shg_pass_a_prelaunch: workspace address ->
 *    launch plan + event, an entry lives from the prelaunch to the scan's first stage or shg_pass_a_forget); (4) every
 *    shg_pool: its threads, its queue and job map; (5) function-local
 *    one-time settings of kernel attributes (dynamic LDS sizes) and cached environment knobs (SHG_*); (6) the BLAS /
 *    LAPACK entry points and callbacks the host control plane was given (shg_host_set_*);
 *  - return value: 0 = OK, >0 = hipError_t, <0 = SHG_E_* argument error; the
 *    message is available per thread from shg_last_error_string();
 *  - "file layout" = the SER frame as stored: [Height][Width], little-endian,
 *    1 or 2 bytes per pixel.  When Width > Height the reference rotates every
 *    frame (video_reader.py:119-120): img[y][x] = raw[x][Width-1-y], ih = Width,
 *    iw = Height.  The kernels never materialise that rotation for the stack.
 */
#ifndef SHG_HIP_H
#define SHG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHG_ABI_VERSION 16

#define SHG_E_ARG        (-1)   /* bad argument (null pointer, non-positive size, ...) */
#define SHG_E_WORKSPACE  (-2)   /* workspace too small                                 */
#define SHG_E_UNSUPPORTED (-3)  /* size outside what the kernel supports               */
/* host control plane: the failure the reference's NumPy / SciPy call would raise at that point */
#define SHG_E_VALUE      (-4)   /* ValueError                                          */
#define SHG_E_TYPE       (-5)   /* TypeError                                           */
#define SHG_E_LINALG     (-6)   /* numpy.linalg.LinAlgError                            */
#define SHG_E_RUNTIME    (-7)   /* RuntimeError                                        */
#define SHG_E_QHULL      (-8)   /* scipy.spatial.QhullError                            */
#define SHG_E_ASSERT     (-9)   /* AssertionError (rescale_brightness's assert)        */
#define SHG_E_INDEX      (-10)  /* IndexError                                          */

typedef void* shg_stream_t;

int         shg_abi_version(void);
const char* shg_last_error_string(void);

/* Optional per-kernel timing (bench.py roofline leg): HIP events recorded on the launch stream
 * directly around each kernel launch, keyed by kernel tag ("accumulate", "reduce_partials",
 * "extract", "warp", "rowpair_stats", "scale_rows", "clahe_hist", "clahe_lut", "clahe_interp",
 * "hist", "rescale", ...).  Off by default.  shg_profile_get waits for the events of `tag`. */
int shg_profile_enable(int on);
int shg_profile_select(const char* tags_csv);   /* only time these tags (NULL or "" = all) */
int shg_profile_reset(void);
int shg_profile_get(const char* tag, double* total_ms, int64_t* launches);
int shg_profile_total(double* total_ms, int64_t* launches);   /* every timed entry point since the last reset */
/* Where a scan worker's HOST time goes inside the stage composites (waiting in stream synchronises, control-plane routines):
 * wall clock per section tag while enabled; the report is "tag seconds calls" lines.  Measurement aid (tools/host_budget.py). */
int shg_host_timing_enable(int on);
int shg_host_timing_report(char* buf, size_t cap);
int shg_profile_dump(const char* path);   /* "tag,stream,start_ms,stop_ms" per sample: the streams' timeline without a profiler */

/* Measurement aid: trivial read-only kernels over `bytes` bytes (16 B/lane non-temporal loads, XOR-folded;
 * out1024: 1024 uint32 words) that bench.py times to quote MEASURED read ceilings beside the spec peak
 * (SURVEY.md section 8d).  mode 0: grid-strided contiguous sweep with `blocks` workgroups; mode 1: the same with
 * pass A's arithmetic; mode 2: pass A's address pattern for frames of vecs_per_frame 16-byte vectors, `blocks`
 * = frame-axis splits.  unroll: 1, 2, 4 or 8 loads in flight per lane. */
int shg_stream_read_probe(const void* buf, int64_t bytes, int mode, int blocks, int unroll,
                          int64_t vecs_per_frame, uint32_t* out1024, shg_stream_t stream);

/* ---- pass A: sum and max over frames -------- solex_util.py:174-188 (compute_mean_max)
 * stack: n_frames frames in file layout.  sum_out[H*W] (file layout) receives the
 * integer sum of the raw samples, max_out[H*W] their maximum (raw sample units;
 * the 8-bit x256 scaling of video_reader.py:121-122 is applied in
 * shg_finalize_mean_max).  Integer sums are order independent, so the result is
 * bit-identical for any sharding of the frames (RCCL SUM / MAX all-reduce). */
size_t shg_accumulate_workspace_bytes(int64_t n_frames, int64_t height, int64_t width, int bytes_per_px);
int shg_accumulate_sum_max(const void* stack, int64_t n_frames, int64_t height, int64_t width,
                           int bytes_per_px, int64_t frame_stride_px, uint64_t* sum_out, uint16_t* max_out,
                           void* workspace, size_t workspace_bytes, shg_stream_t stream);

/* ---- decode: frames into HBM ---------------------- video_reader.py:94-123
 * The stack may keep a padded frame pitch: frame k starts frame_stride_px samples after frame k-1
 * (0 = dense, height*width).  shg_frame_pitch_bytes gives the pitch the frame-walking kernels read fastest
 * (the frame size rounded up to 8 KiB); shg_upload_frames copies n_frames dense frames from PINNED host
 * memory into such a stack with one asynchronous 2-D hipMemcpy on `stream`. */
int64_t shg_frame_pitch_bytes(int64_t frame_bytes);
int shg_upload_frames(void* dst, int64_t dst_pitch_bytes, const void* host_src, int64_t frame_bytes,
                      int64_t n_frames, shg_stream_t stream);

/* Uncompressed AVI frames (cv2.VideoCapture + COLOR_BGR2GRAY, video_reader.py:68-80, 111-113) into the
 * uint8 stack: raw holds n_frames chunk payloads raw_pitch_bytes apart, each `height` rows of row_bytes
 * (bottom_up: last row first); bits 8 (optional 256-entry grey table for palettised frames, NULL =
 * identity) or 24 (B, G, R; OpenCV 4's (B*3735 + G*19235 + R*9798 + 2^14) >> 15).  At most 65535
 * frames per call. */
int shg_unpack_dib_frames(const uint8_t* raw, int64_t n_frames, int64_t raw_pitch_bytes, int64_t height,
                          int64_t width, int bits, int64_t row_bytes, int bottom_up,
                          const uint8_t* gray_lut, uint8_t* stack, int64_t frame_stride_px,
                          shg_stream_t stream);

/* mean = trunc(sum / n_total) as uint16 (solex_util.py:188), 8-bit samples scaled by
 * 256, both images rotated into the reference's [ih][iw] orientation. */
int shg_finalize_mean_max(const uint64_t* sum, const uint16_t* max_raw, int64_t n_total,
                          int64_t height, int64_t width, int bytes_per_px,
                          uint16_t* mean_out, uint16_t* max_out, shg_stream_t stream);
/* The frame statistics of a frame-sharded scan after their exchange (the reference has no such step: its compute_mean_max,
 * solex_util.py:174-188, sees every frame): n_pieces records of piece_words 32-bit words, one per rank, each
 * [npix partial sums as 32-bit words | npix partial maxima as 16-bit words, padded to a whole word | ...] -- what ONE all-gather of
 * every rank's packed statistics brings (dist.exchange_frame_stats) -- folded into the 64-bit sums and the maxima
 * shg_finalize_mean_max takes.  Integer: the result does not depend on the number of pieces. */
int shg_reduce_frame_stats(const uint32_t* pieces, int n_pieces, int64_t piece_words, int64_t npix, uint64_t* sum_out,
                           uint16_t* max_out, shg_stream_t stream);
/* compute_mean_max (solex_util.py:174-188) for a scan that is whole on this GPU: pass A, then mean and max images straight
 * from its per-slab partials (shg_accumulate_sum_max + shg_finalize_mean_max without the 64-bit totals in between).
 * workspace: shg_accumulate_workspace_bytes. */
int shg_accumulate_mean_max(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px,
                            int64_t frame_stride_px, uint16_t* mean_out, uint16_t* max_out, void* workspace,
                            size_t workspace_bytes, shg_stream_t stream);

/* ---- cv2.blur(img_u16, (kw, kh)) ------------- solex_util.py:166, 230
 * Normalised box filter, anchor (kw/2, kh/2), BORDER_REFLECT_101, round half even.
 * tmp: uint32 scratch of h*w elements. */
int shg_box_blur_u16(const uint16_t* src, int64_t h, int64_t w, int kw, int kh,
                     uint16_t* dst, uint32_t* tmp, shg_stream_t stream);

/* first-occurrence argmin over columns [x0, x1) of every row: np.argmin(img[:, x0:x1], axis=1)
 * (solex_util.py:231, 242).  out[h] int32, relative to x0. */
int shg_row_argmin_u16(const uint16_t* img, int64_t h, int64_t w, int64_t x0, int64_t x1,
                       int32_t* out, shg_stream_t stream);

/* np.mean(img, axis=1) in float64 (solex_util.py:167). out[h]. */
int shg_row_mean_u16(const uint16_t* img, int64_t h, int64_t w, double* out, shg_stream_t stream);

/* The two uses of cv2.blur on the path in fused form (the blurred image never leaves the workgroup): row means of
 * blur(img, (kw, kh)) for detect_bord (solex_util.py:166-167), and the first arg-minimum over [x0, x1) of every
 * blurred row together with the first arg-minimum of the unblurred row (solex_util.py:230-231, 242).  Identical
 * results to shg_box_blur_u16 + shg_row_mean_u16 / shg_row_argmin_u16.  shg_blur_fits_fused: whether (8 + kh - 1)
 * rows of w horizontal sums fit the LDS tile (otherwise use the separate entry points). */
int shg_blur_fits_fused(int64_t w, int kh);
int shg_blur_row_mean_u16(const uint16_t* src, int64_t h, int64_t w, int kw, int kh, double* out, shg_stream_t stream);
int shg_blur_argmin_u16(const uint16_t* src, int64_t h, int64_t w, int kw, int kh, int64_t x0, int64_t x1,
                        int32_t* out_blur, int32_t* out_sharp, shg_stream_t stream);

/* ---- pass B: per-frame column extraction ----- solex_util.py:93-144 (read_video_improved)
 * For every frame k, shift s and slit row y:
 *     v = img[y][ind_l[s][y]] * lw[y] + img[y][ind_l[s][y] + 1] * rw[y]     (float64,
 *         two rounded products and one rounded add, no FMA; solex_util.py:131-133)
 *     disks[s][y][col(k)] = (uint16) v                                        (truncation, :134)
 * with col(k) = k_offset + k, or n_cols - 1 - (k_offset + k) when flip_x != 0
 * (np.flip(axis=1), Solex_recon.py:74-76).  ind_l is [n_shifts][ih] int32, already
 * clamped to [0, iw-2] (solex_util.py:114-119); lw, rw are [ih] float64, NOT adjusted
 * for the clamp (solex_util.py:122-123).  disks: n_shifts planes of plane_stride
 * elements, rows of row_pitch elements (row_pitch >= n_cols).
 * Rotated files (width > height, the usual SER) take the band kernel (k_extract_band: a group of shifts per workgroup,
 * every sample loaded once, nontemporal stores); SHG_EXT_GENERAL=1 in the environment keeps the kernel that serves
 * un-rotated files (k_extract) for them too -- the parity tests hold one against the other. */
int shg_extract_columns(const void* stack, int64_t n_frames, int64_t height, int64_t width,
                        int bytes_per_px, int64_t frame_stride_px, const int32_t* ind_l, const double* lw, const double* rw,
                        int n_shifts, uint16_t* disks, int64_t row_pitch, int64_t plane_stride,
                        int64_t n_cols, int64_t k_offset, int flip_x, shg_stream_t stream);

/* The same, and on the way the minimum and maximum of every plane, as the warp clips to them (ellipse_to_circle.py:
 * 112-114): minmax_slots is scratch + result, uint32 [n_shifts][64][2] slots (zeroed by the call, folded by a last tiny
 * kernel) followed by the result [n_shifts][2] = {min, max} -- n_shifts * 130 words in all; pass &result[s * 2] to
 * shg_warp_rows_minmax_u16.  Only meaningful when the call covers the whole scan (a rank's share of a sharded scan
 * gives the extrema of its columns only).  minmax_slots may be NULL. */
int shg_extract_columns_minmax(const void* stack, int64_t n_frames, int64_t height, int64_t width,
                               int bytes_per_px, int64_t frame_stride_px, const int32_t* ind_l, const double* lw, const double* rw,
                               int n_shifts, uint16_t* disks, int64_t row_pitch, int64_t plane_stride,
                               int64_t n_cols, int64_t k_offset, int flip_x, uint32_t* minmax_slots, shg_stream_t stream);

/* The same for a Doppler stack whose shifts are consecutive integers in any order (-w a:b:1 with the two implicit shifts inside
 * the range; 3 <= n_shifts <= 24, shg_extract_dense_fits): a lane loads the n_shifts + 1 distinct samples of a (row, frame)
 * once instead of 2 * n_shifts (rotated files: k_extract_band in groups of up to seven shift values, the file row two groups
 * share read by neighbouring workgroups of one XCD; un-rotated files: k_extract_dense).  host_shifts [n_shifts]: the shift of every plane; base_col [ih]: the column of the smallest
 * shift before the clamps (fit[:, 0] + min shift); ind_l as above, used for the rows whose line lies within n_shifts columns
 * of the frame's edge.  Bit-identical to shg_extract_columns_minmax. */
int shg_extract_dense_fits(const int32_t* host_shifts, int n_shifts);
int shg_extract_columns_dense(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px,
                              int64_t frame_stride_px, const int32_t* ind_l, const int32_t* base_col, const double* lw,
                              const double* rw, const int32_t* host_shifts, int n_shifts, uint16_t* disks, int64_t row_pitch,
                              int64_t plane_stride, int64_t n_cols, int64_t k_offset, int flip_x, uint32_t* minmax_slots,
                              shg_stream_t stream);

/* ---- the warp ---------------------------------- ellipse_to_circle.py:112-118
 * skimage.transform.warp(order=1, mode='constant', cval=image[0,0], clip) for a
 * transform that never moves rows: out[r][c] samples input row r at
 * x = h00*c + h01*r + h02 (float64), bilinear, then clip to [min, max] of the input
 * and (uint16)(65536 * v).  The input is the uint16 disk, interpreted as
 * value/65536 (Solex_recon.py:123, ellipse_to_circle.py:299).  minmax: 2 uint32
 * device scratch words (written by the call). */
int shg_warp_rows_u16(const uint16_t* src, int64_t h, int64_t w, int64_t src_pitch,
                      double h00, double h01, double h02,
                      uint16_t* dst, int64_t out_h, int64_t out_w, int64_t dst_pitch,
                      uint32_t* minmax, shg_stream_t stream);

/* The warp with the input's extrema already known (minmax2 = {min, max}, e.g. from shg_extract_columns_minmax): saves
 * the pass over the input that shg_warp_rows_u16 makes to find them. */
int shg_warp_rows_minmax_u16(const uint16_t* src, int64_t h, int64_t w, int64_t src_pitch,
                             double h00, double h01, double h02,
                             uint16_t* dst, int64_t out_h, int64_t out_w, int64_t dst_pitch,
                             const uint32_t* minmax2, shg_stream_t stream);

/* ---- transversalium ---------------------------- solex_util.py:383-395, 76-86
 * Per row y in (y1, y2): the mean of the 2-MAD inliers of log(img[y][a:b] / img[y-1][a:b])
 * with a, b the chord of `circle` clipped to `borders` (float64).  out[y2 - y1]
 * (out[0] = 0 as solex_util.py:386).  xa, xb: int32 [y2-y1] column bounds computed on the
 * host (solex_util.py:389-391).  Images wider than SHG_TRANSV_MAX_COLS
 * (19456 columns: the row's keys must fit the CU's 160 KiB of LDS) are rejected.
 * row_factor (may be NULL): float64 [h]; when given the image is the float64 frame
 * img[y][x] * row_factor[y] that removeVignette returns (solex_util.py:654). */
#define SHG_TRANSV_MAX_COLS 19456
int shg_rowpair_logratio_stats(const uint16_t* img, int64_t h, int64_t w, int64_t pitch,
                               int64_t y1, int64_t y2, const int32_t* xa, const int32_t* xb,
                               const double* row_factor, double* out, shg_stream_t stream);
/* The same, also storing the statistics at out_mirror (may be NULL): a second, GPU-mapped host buffer for the control
 * plane, so that no copy has to follow (shg_stage_process_frames). */
int shg_rowpair_logratio_stats_mirrored(const uint16_t* img, int64_t h, int64_t w, int64_t pitch,
                                        int64_t y1, int64_t y2, const int32_t* xa, const int32_t* xb,
                                        const double* row_factor, double* out, double* out_mirror, shg_stream_t stream);

/* scipy.ndimage.correlate1d(src, weights, axis=-1, mode='constant', cval=0) for k rows of n float64
 * samples with 2*radius+1 float64 weights (device memory), in NI_Correlate1D's order of operations:
 * the interior of scipy.signal.savgol_filter(y_ratios_r, window, 3) (solex_util.py:400).  `symmetric`
 * > 0 selects SciPy's symmetric-filter form (pairs summed first, left half of the weights used), which
 * SciPy takes when |w[R+i] - w[R-i]| <= DBL_EPSILON for every i; < 0 its antisymmetric form (pairs
 * subtracted; |w[R+i] + w[R-i]| <= DBL_EPSILON); 0 the general form. */
int shg_correlate1d_rows_f64(const double* src, int64_t k, int64_t n, const double* weights, int radius,
                             int symmetric, double* dst, shg_stream_t stream);

/* ret = min(img * c[y], 65535) truncated to uint16 (solex_util.py:489, 515-516). */
int shg_scale_rows_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, const double* c,
                       const double* row_factor, uint16_t* dst, int64_t dst_pitch, shg_stream_t stream);

/* ---- f4: "stubborn" transversalium, the line filter of apply_lin_filter + fix_edge_effect ----
 * Replaces the three cv2.filter2D calls on log(img) (solex_util.py:277-353) and the limb clean-up
 * (:356-375).  Exact BORDER_REFLECT_101 correlation: float64 sums in a fixed order, rounded once to
 * float32 for a uint16 image (np.log(uint16) is float32) or kept in float64 for the float64 image
 * img * row_factor[y] (row_factor non-NULL).
 *
 * shg_lin_filter_row_sums: hl[h][w] = sum of the `linlen` logs centred on x along row y; hf the same
 * for the image whose flagged rows are replaced by (nearest unflagged row above)/2 + (… below)/2
 * (up/dn: int32 [h], -1 = none; only read where flagged[y] != 0).  log_lut: float32 [65536], the
 * log of every uint16 value as the caller's NumPy computes it (ignored when row_factor is given).
 *
 * shg_lin_filter_apply: delta = hl/linlen - (sum of hf over rows y-half_width..y+half_width except y)
 * /(2*half_width*linlen); kept on columns [xa[y], xb[y]), copied from column xa+edge_half into
 * [xa, xa+edge_half) when edge[y] & 1, from xb-edge_half-1 into [xb-edge_half, xb) when edge[y] & 2,
 * zero elsewhere; dst = min(img * exp(-delta * taper[y]), 65535) truncated (solex_util.py:352, 423). */
int shg_lin_filter_row_sums(const uint16_t* img, int64_t h, int64_t w, int64_t pitch,
                            const double* row_factor, const float* log_lut, const uint8_t* flagged,
                            const int32_t* up, const int32_t* dn, int linlen, double* hl, double* hf,
                            shg_stream_t stream);
int shg_lin_filter_apply(const uint16_t* img, int64_t h, int64_t w, int64_t pitch,
                         const double* row_factor, const double* hl, const double* hf, int linlen,
                         int half_width, const double* taper, const int32_t* xa, const int32_t* xb,
                         const uint8_t* edge, int edge_half, uint16_t* dst, int64_t dst_pitch,
                         shg_stream_t stream);

/* np.percentile(img, q, axis) building block (removeVignette, solex_util.py:591-592): for every
 * column (axis 0) or row (axis 1) the rank_lo-th and rank_hi-th smallest values (0-based). */
int shg_line_order_stats_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, int axis,
                             int64_t rank_lo, int64_t rank_hi, uint16_t* out_lo, uint16_t* out_hi,
                             shg_stream_t stream);

/* ---- crop / pad -------------------------------- Solex_recon.py:155-171
 * dst[h][nw] = fill everywhere, then dst[:, dx0:dx0+n] = src[:, sx0:sx0+n].
 * fill: 0..65535, or negative = src[0][0] read on the device (np.full(.., img[0, 0]), :161). */
int shg_crop_pad_u16(const uint16_t* src, int64_t h, int64_t w, int64_t pitch,
                     uint16_t* dst, int64_t nw, int64_t dst_pitch,
                     int64_t sx0, int64_t dx0, int64_t n, int32_t fill, shg_stream_t stream);

/* ---- CLAHE -------------------------------------- solex_util.py:532-533, clahe_apply.py:247
 * cv2.createCLAHE(clipLimit, (tiles, tiles)).apply(img) for uint16 (hist_size 65536)
 * or uint8 (hist_size 256) images.  workspace: tiles*tiles*hist_size uint32 histograms
 * followed by tiles*tiles*hist_size LUT entries (uint16); query the size first. */
size_t shg_clahe_workspace_bytes(int tiles, int bytes_per_px);
/* The size that lets a 16-bit call build its tile histograms without global atomics (slice histograms stored whole
 * and reduced once, the LUT by 32 workgroups per tile): shg_clahe takes that path whenever workspace_bytes allows. */
size_t shg_clahe_workspace_bytes_for(int64_t h, int64_t w, int tiles, int bytes_per_px);
int shg_clahe(const void* img, int64_t h, int64_t w, int64_t pitch, int bytes_per_px,
              double clip_limit, int tiles, void* dst, int64_t dst_pitch,
              void* workspace, size_t workspace_bytes, shg_stream_t stream);

/* 65536-bin (or 256-bin) histogram of an image; hist is zeroed by the call.  Feeds
 * np.percentile / np.max on the host (solex_util.py:535-537). */
int shg_hist(const void* img, int64_t h, int64_t w, int64_t pitch, int bytes_per_px,
             uint32_t* hist, shg_stream_t stream);

/* Exact order statistics of a uint16 image: out[i] (as double) = the host_ranks[i]-th smallest pixel
 * (0-based; rank h*w-1 = np.max).  Feeds np.percentile without moving a histogram to the host. */
size_t shg_select_u16_workspace_bytes(int n_ranks);
int shg_select_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, const int64_t* host_ranks,
                   int n_ranks, double* out, void* workspace, size_t workspace_bytes, shg_stream_t stream);

/* rescale_brightness: trunc(clamp((sat*alpha)*(v-lo)/(hi-lo), 0, sat)), float64, sat = 65535
 * (solex_util.py:519-525). */
int shg_rescale_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch,
                    double lo, double hi, double alpha, uint16_t* dst, int64_t dst_pitch,
                    shg_stream_t stream);
/* The same for 8-bit images (sat = 255): clahe_apply.py:251 stretches 8-bit PNGs too. */
int shg_rescale_u8(const uint8_t* img, int64_t h, int64_t w, int64_t pitch, double lo, double hi,
                   double alpha, uint8_t* dst, int64_t dst_pitch, shg_stream_t stream);

/* cv2.circle(img, (x0, y0), r, value, -1): filled integer midpoint circle, clipped to the
 * image (solex_util.py:542-547).  scratch: unused (may be NULL); kept for ABI stability. */
int shg_fill_disc_u16(uint16_t* img, int64_t h, int64_t w, int64_t pitch,
                      int64_t x0, int64_t y0, int64_t r, uint16_t value, int32_t* scratch,
                      shg_stream_t stream);

/* skimage.transform.downscale_local_mean(img/65536, (f, f)) (ellipse_to_circle.py:299-302):
 * zero-padded block mean, float64 out [ceil(h/f)][ceil(w/f)]. */
int shg_downscale_mean_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, int factor,
                           double* dst, shg_stream_t stream);

/* ---- limb detection on the block-mean image ------ ellipse_to_circle.py:148-250
 * cv2.blur(float64 image, (k, k)) (ellipse_to_circle.py:163, 241): box sums accumulated left to
 * right then top to bottom, times 1/(k*k), BORDER_REFLECT_101, anchor k/2.  tmp: h*w doubles. */
int shg_box_blur_f64(const double* src, int64_t h, int64_t w, int k, double* dst, double* tmp,
                     shg_stream_t stream);

/* cv2.blur for an image whose values are whole numbers of 2^-20 below 1 (the 4x4 block mean of uint16 / 65536,
 * ellipse_to_circle.py:299-301), k <= 63: also keys[i] = the k x k window sum in units of 2^-20 (exact, < 2^32), so that
 * np.median / np.percentile of the blurred image can be selected on 32-bit integers: shg_select_keys_u32, out[i] = the
 * host_ranks[i]-th smallest of host_keys[i][n] turned back into the blurred value, (key * 2^-20) * (1 / (k_i * k_i)) --
 * three 11-bit radix passes instead of the eight 8-bit passes float64 keys take. */
int shg_box_blur_key_f64(const double* src, int64_t h, int64_t w, int k, double* dst, uint32_t* keys, double* tmp,
                         shg_stream_t stream);
size_t shg_select_keys_workspace_bytes(int n_pairs);
int shg_select_keys_u32(const uint32_t* const* host_keys, int64_t n, const int64_t* host_ranks, const int* host_k,
                        int n_pairs, double* out, void* workspace, size_t workspace_bytes, shg_stream_t stream);

/* Exact order statistics: out[i] = the host_ranks[i]-th smallest value (0-based) of values[n]
 * (MSB-first radix select; feeds np.median / np.percentile, ellipse_to_circle.py:165, 241).
 * host_ranks is a HOST array of n_ranks <= 8 entries. */
size_t shg_select_workspace_bytes(int n_ranks);
int shg_select_f64(const double* values, int64_t n, const int64_t* host_ranks, int n_ranks, double* out,
                   void* workspace, size_t workspace_bytes, shg_stream_t stream);
/* the same for several arrays of one length in one launch sequence: out[i] = the host_ranks[i]-th smallest value of
 * host_arrays[i] (HOST array of device pointers). */
int shg_select_multi_f64(const double* const* host_arrays, int64_t n, const int64_t* host_ranks, int n_ranks,
                         double* out, void* workspace, size_t workspace_bytes, shg_stream_t stream);

/* get_flood_image's statistics (ellipse_to_circle.py:159-169): stats[0] = np.sum(image) (image
 * values are multiples of 2^-20, as the 4x4 block mean of uint16/65536 is), and over
 * data = blurred[blurred < very_bright]: stats[1] = min, stats[2] = max,
 * counts[20] = np.histogram(data, bins=20)[0].  workspace: 256 bytes. */
int shg_flood_stats_f64(const double* image, const double* blurred, int64_t n, double very_bright,
                        double* stats, uint32_t* counts, void* workspace, shg_stream_t stream);
/* The same with very_bright = np.percentile(blurred, 99) formed on the device from the two order
 * statistics order_stats[0..1] (device memory, e.g. shg_select_multi_f64's output) by NumPy's _lerp
 * with weight gamma: b - (b-a)*(1-gamma) if gamma >= 0.5 else a + (b-a)*gamma. */
int shg_flood_stats_lerp_f64(const double* image, const double* blurred, int64_t n,
                             const double* order_stats, double gamma, double* stats, uint32_t* counts,
                             void* workspace, shg_stream_t stream);

/* skimage.feature.canny(flooded, sigma, low, high) up to its two hysteresis masks
 * (ellipse_to_circle.py:245-250), where flooded = (blurred < flood_thresh ? 0 : 65000)
 * (ellipse_to_circle.py:226-227).  host_gauss_weights: the 2*radius+1 normalised Gaussian taps
 * (HOST pointer, copied into the launch).  low_mask / high_mask: uint8 [h][w] =
 * local_maxima & (magnitude >= low / high).  The 8-connected hysteresis is a labelling step
 * the caller does on those masks. */
size_t shg_canny_workspace_bytes(int64_t h, int64_t w);
int shg_canny_masks_f64(const double* blurred, int64_t h, int64_t w, double flood_thresh,
                        const double* host_gauss_weights, int radius, double low, double high,
                        uint8_t* low_mask, uint8_t* high_mask, void* workspace, size_t workspace_bytes,
                        shg_stream_t stream);

/* canny's hysteresis and the labelling of its result (ellipse_to_circle.py:245-252): the pixels of
 * the 8-connected components of low_mask that contain a high_mask pixel, in raster order:
 * out_idx[i] = y*w + x, out_root[i] = smallest linear index of the pixel's component (sorting the
 * distinct roots gives scipy.ndimage.label's numbering), out_count[0] = number of pixels.
 * out_idx / out_root: h*w int32 each. */
size_t shg_edge_components_workspace_bytes(int64_t h, int64_t w);
int shg_edge_components(const uint8_t* low_mask, const uint8_t* high_mask, int64_t h, int64_t w,
                        int32_t* out_idx, int32_t* out_root, int32_t* out_count,
                        void* workspace, size_t workspace_bytes, shg_stream_t stream);

/* ---- image_process in two calls --------------------------------- solex_util.py:527-547
 * shg_contrast_stats_u16: cl1 = CLAHE(frame, clip_limit, tiles), then the order statistics np.percentile and np.max
 * need: out5[0..1] = frame's ranks_frame2[0..1]-th smallest pixels, out5[2..4] = cl1's ranks_cl13[0..2]-th (host
 * rank arrays).  shg_contrast_products_u16: high_contrast = rescale(frame, lo_hi6[0], lo_hi6[1]), protus =
 * rescale(frame, lo_hi6[2], lo_hi6[3]), cc = rescale(cl1, lo_hi6[4], lo_hi6[5]) (host doubles), then the filled disc
 * of value 80 on protus when disc_r > 0 (:542-547).  Same kernels and order as the separate entry points. */
size_t shg_contrast_stats_workspace_bytes(int tiles);
size_t shg_contrast_stats_workspace_bytes_for(int64_t h, int64_t w, int tiles);   /* with shg_clahe_workspace_bytes_for's room */
int shg_contrast_stats_u16(const uint16_t* frame, int64_t h, int64_t w, int64_t pitch, double clip_limit,
                           int tiles, uint16_t* cl1, int64_t cl1_pitch, const int64_t* ranks_frame2,
                           const int64_t* ranks_cl13, double* out5, void* workspace, size_t workspace_bytes,
                           shg_stream_t stream);
int shg_contrast_products_u16(const uint16_t* frame, int64_t frame_pitch, const uint16_t* cl1, int64_t cl1_pitch,
                              int64_t h, int64_t w, const double* lo_hi6, uint16_t* high_contrast,
                              uint16_t* protus, uint16_t* cc, int64_t dst_pitch, int64_t disc_x0,
                              int64_t disc_y0, int64_t disc_r, shg_stream_t stream);

/* ---- the limb stage's kernels fused through LDS tiles ------ ellipse_to_circle.py:148-291, 299-302 (csrc/limb_fused.hip)
 * The same arithmetic as shg_downscale_mean_u16 / shg_box_blur_key_f64 / shg_select_keys_u32 / shg_flood_stats_lerp_f64 and
 * shg_canny_masks_f64 / shg_edge_components, bit for bit, in 5 + 3 launches instead of 12 + 11.
 * shg_limb_fused_fits: whether the k x k blur window (k = int(0.01 * sh)) fits the tile (k <= 16).
 * shg_limb_prepare: for the uint16 disk [h][w] and its 4x4 block mean [sh = ceil(h/4)][sw = ceil(w/4)]: cv2.blur with windows k
 *   and 5 (as 32-bit window sums in units of 2^-20), host_ranks4 = the two order statistics of np.median(blur 5) and the two
 *   of np.percentile(blur k, 99), very_bright = NumPy's _lerp of the latter with weight gamma99, and np.sum(image), min, max,
 *   np.histogram(., 20) of blur k below very_bright.  packed (device or GPU-mapped host memory, 32 doubles): [0..3] order
 *   statistics, [4] sum, [5] min, [6] max, 20 uint32 counts at packed + 8.  *keys_out: the window sums of blur k, inside the
 *   workspace, for shg_limb_edges.
 * shg_limb_edges: canny (flooded = blurred < flood_thresh ? 0 : 65000; Gaussian taps as shg_canny_masks_f64) and the
 *   8-connected labelling of its LOW mask: comp (device or GPU-mapped host memory, 2 * sh * sw + 1 int32) = [m | idx[n] |
 *   root[n]]: the m low-mask pixels in raster order, root = smallest linear index of the pixel's component, bit 30 of root set
 *   where the pixel is in the HIGH mask (hysteresis = keep the components holding such a pixel: one pass for the caller). */
int shg_limb_fused_fits(int64_t sh, int64_t sw, int k);
size_t shg_limb_prepare_workspace_bytes(int64_t sh, int64_t sw, int k);
int shg_limb_prepare(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, int k, const int64_t* host_ranks4, double gamma99,
                     double* packed, const uint32_t** keys_out, void* workspace, size_t workspace_bytes, shg_stream_t stream);
size_t shg_limb_edges_workspace_bytes(int64_t sh, int64_t sw);
int shg_limb_edges(const uint32_t* keys, int64_t sh, int64_t sw, int k, double flood_thresh, const double* host_gauss_weights,
                   int radius, double low, double high, int32_t* comp, void* workspace, size_t workspace_bytes, shg_stream_t stream);

/* ==== stage composites ============================================================================
 * Each stage function of the reference as one call: the same kernels in the same order as the entry points
 * above, the scalars / 1-D vectors in between brought to the host through PINNED memory and handled by the
 * host control plane below -- no interpreter (and no interpreter lock) between the kernels, so several
 * scans run side by side in one process.  Device `workspace` and pinned `host_pinned` staging areas are
 * sized by the *_bytes queries; a stage synchronises `stream` where the control plane needs a result. */

/* compute_mean_return_fit (solex_util.py:191-259): pass A (or the given, e.g. all-reduced, sum_in / max_in),
 * mean / max images [ih][iw] (dense), detect_bord, the two argmin traces, the cubic fit.
 * host_y12 = clipped (y1, y2); host_p4 lowest power first; host_fit [ih][4]; host_trace_sharp [ih] and
 * host_mask_good [y2-y1 <= ih] (both may be NULL) feed the reference's diagnostic plot. */
size_t shg_stage_mean_fit_workspace_bytes(int64_t n_frames, int64_t height, int64_t width, int bytes_per_px);
size_t shg_stage_mean_fit_host_bytes(int64_t height, int64_t width);
int shg_stage_mean_fit(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px,
                       int64_t frame_stride_px, const uint64_t* sum_in, const uint16_t* max_in, int64_t n_total,
                       uint16_t* mean_out, uint16_t* max_out, int64_t* host_y12, double* host_p4, double* host_fit,
                       int32_t* host_trace_sharp, uint8_t* host_mask_good, void* workspace, size_t workspace_bytes,
                       void* host_pinned, size_t host_pinned_bytes, shg_stream_t stream);

/* read_video_improved (solex_util.py:93-144) from the host `fit` and shift list: sample columns and weights
 * (shg_host_column_plan), upload, shg_extract_columns_minmax -- or shg_extract_columns_dense for consecutive shifts
 * (minmax_slots may be NULL).  Asynchronous: host_pinned is read by the GPU after the call has returned and must stay untouched
 * until the work queued on `stream` up to here has run. */
size_t shg_stage_extract_workspace_bytes(int64_t height, int64_t width, int n_shifts);   /* device and pinned */
int shg_stage_extract(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px,
                      int64_t frame_stride_px, const double* host_fit, const int32_t* host_shifts, int n_shifts,
                      uint16_t* disks, int64_t row_pitch, int64_t plane_stride, int64_t n_cols, int64_t k_offset,
                      int flip_x, uint32_t* minmax_slots, void* workspace, size_t workspace_bytes, void* host_pinned,
                      size_t host_pinned_bytes, shg_stream_t stream);

/* get_edge_list on the 4x4 block mean of the disk (ellipse_to_circle.py:231-291, 299-302): flood image, canny ladder
 * (sigma 2, 1.5, 1, 0.5; host_gauss_taps = scipy's taps for those, 17 + 13 + 9 + 5 values), hysteresis + labelling,
 * region choice, convex-hull filter, row crop.  host_points [points_cap][2]: edge pixels (row, col) of the quarter-size
 * image in raster order; host_flags [points_cap]: 1 = limb point; host_counts2 = edge pixels, limb points.  The
 * ellipse fit on the limb points stays with the caller (NumPy, see csrc/hostmath.hip). */
size_t shg_stage_limb_points_workspace_bytes(int64_t h, int64_t w);
size_t shg_stage_limb_points_host_bytes(int64_t h, int64_t w);
int shg_stage_limb_points(const uint16_t* disk, int64_t h, int64_t w, int64_t pitch, const double* host_gauss_taps,
                          int32_t* host_points, uint8_t* host_flags, int64_t points_cap, int64_t* host_counts2,
                          void* workspace, size_t workspace_bytes, void* host_pinned, size_t host_pinned_bytes,
                          shg_stream_t stream);

/* ellipse_to_circle without its warp (ellipse_to_circle.py:294-314): shg_stage_limb_points, then two_step, correct_image's
 * geometry and the borders (shg_host_limb_geometry).  Flag bit 1 = kept by two_step; host_counts3 = edge pixels, limb
 * points, kept points. */
int shg_stage_limb_fit(const uint16_t* disk, int64_t h, int64_t w, int64_t pitch, const double* host_gauss_taps,
                       int32_t* host_points, uint8_t* host_flags, int64_t points_cap, int64_t* host_counts3,
                       double* host_geom16, int64_t* host_dims2, double* host_outline200, void* workspace,
                       size_t workspace_bytes, void* host_pinned, size_t host_pinned_bytes, shg_stream_t stream);

/* single_image_process for the k requested disks of a file (Solex_recon.py:136-174; solex_util.py:383-516,
 * 527-547): transversalium, crop / pad, CLAHE + order statistics, the three rescales + protuberance disc.
 * See csrc/stages.hip for the argument list. */
size_t shg_stage_process_workspace_bytes(int64_t k, int64_t h, int64_t w, int64_t crop_w, int tiles);
size_t shg_stage_process_host_bytes(int64_t k, int64_t h);
int shg_stage_process_frames(const uint16_t* const* host_frames, int64_t k, int64_t h, int64_t w, int64_t pitch,
                             int transversalium, const double* host_circle3, const double* host_borders4,
                             const double* host_taps, int64_t window, double* host_factors, int64_t crop_w,
                             int64_t sx0, int64_t dx0, int64_t ncopy, double clip_limit, int tiles, int64_t disc_x0,
                             int64_t disc_y0, int64_t disc_r, uint16_t* const* host_detrans, int64_t detrans_pitch,
                             uint16_t* const* host_final, uint16_t* const* host_cl1, uint16_t* const* host_hc,
                             uint16_t* const* host_protus, uint16_t* const* host_cc, int64_t out_pitch, void* workspace,
                             size_t workspace_bytes, void* host_pinned, size_t host_pinned_bytes, shg_stream_t stream);

/* ==== one scan, one call =========================================================================
 * The whole per-file flow of solex_do_work's loop body (Solex_recon.py:33-42: solex_read :50-83, solex_process :94-134,
 * single_image_process :136-174) as ONE call, so that nothing of a scan in flight waits for the caller's interpreter:
 * shg_stage_mean_fit -> shg_stage_extract -> (limb fit of the first disk, or the fixed ratio / slant of the options) ->
 * the warp of every requested disk -> shg_stage_process_frames.  Same kernels and host control plane as the stage
 * composites; what the caller's language would have done between them (shift bookkeeping, phi through degrees and back
 * as options['slant_fix'] stores it, the crop plan, the protuberance disc, the Savitzky-Golay window) is restated here.
 * Image sizes after the warp are only known once the limb is fitted, so the images go into two ARENAS the caller sizes by
 * guess: when one (or the workspace) turns out too small the call returns SHG_E_WORKSPACE with needed_* filled in and
 * phase_done = 3; the caller grows the buffers and calls again with start_phase = 3 and the same result struct (the raw
 * disks and the geometry are kept; nothing is computed twice). */
typedef struct shg_scan_request {
    uint32_t struct_bytes;            /* sizeof(shg_scan_request) of the caller */
    int32_t  start_phase;             /* 0: whole scan; 3: resume at the warp after SHG_E_WORKSPACE */
    /* the frame stack, file layout */
    const void* stack;
    int64_t  n_frames, height, width, frame_stride_px;
    int32_t  bytes_per_px;
    int32_t  flip_x;                  /* options['flip_x'] (Solex_recon.py:74-76) */
    /* options['shift'] after solex_read's de-duplication (:55): [ellipse_fit_shift, 0, requested...] */
    const int32_t* host_shifts;
    const uint8_t* host_requested;    /* [n_shifts]: shift in options['shift_requested'] */
    int32_t  n_shifts;
    int32_t  want_fit_image;          /* warp the first disk even when it is not requested (diagnostic plot) */
    double   ratio_fixe, slant_fix_deg;   /* options['ratio_fixe'] / ['slant_fix']; NaN = None (both NaN: fit the limb) */
    int32_t  transversalium, keep_detrans;
    int64_t  trans_strength;
    const double* host_taps;          /* savgol_coeffs(taps_window, 3) for the window the caller expects (may be NULL); */
    int64_t  taps_window;             /* another window is asked from shg_host_set_savgol_taps's callback */
    int32_t  crop_square, has_fixed_width;
    int64_t  fixed_width;
    int32_t  disk_display, tiles;
    int64_t  delta_radius;
    double   clip_limit;
    const double* host_gauss_taps;    /* as shg_stage_limb_points */
    /* device buffers of the caller */
    uint16_t* mean_out;               /* [ih][iw] dense */
    uint16_t* max_out;
    uint16_t* disks;                  /* [n_shifts] planes, rows of disk_pitch elements */
    int64_t  disk_pitch, disk_plane_stride;
    uint32_t* minmax_slots;           /* n_shifts * 130 words (shg_extract_columns_minmax) */
    void*    arena;                   /* images the caller may drop early: corrected frames, final / CLAHE / high-contrast */
    size_t   arena_bytes;
    void*    results;                 /* the images solex_process returns: protus and cc of every requested disk */
    size_t   results_bytes;
    void*    workspace;
    size_t   workspace_bytes;
    void*    host_pinned;             /* page-locked, GPU-mapped; read by the GPU until the call's last kernel has run */
    size_t   host_pinned_bytes;
    /* host outputs */
    double*  host_fit;                /* [ih][4] */
    int32_t* host_trace_sharp;        /* [ih] or NULL */
    uint8_t* host_mask_good;          /* [ih] or NULL */
    int32_t* host_points;             /* [points_cap][2] */
    uint8_t* host_flags;              /* [points_cap] */
    int64_t  points_cap;              /* ceil(ih/4) * ceil(n_frames/4) always suffices */
    double*  host_outline200;         /* or NULL */
    double*  host_factors;            /* [n_out][ih] transversalium row factors, or NULL */
} shg_scan_request;

typedef struct shg_scan_result {
    int32_t phase_done;               /* 0 nothing | 1 line fit | 2 raw disks | 3 geometry | 4 products */
    int32_t limb_fitted;              /* this call fitted the limb (options['ratio_fixe'] / ['slant_fix'] are to be set) */
    int64_t y1, y2;                   /* backup bounds */
    double  p4[4];
    int64_t counts3[3];               /* edge pixels, limb points, kept points */
    double  geom16[16];               /* shg_host_limb_geometry (when limb_fitted) */
    double  phi, ratio;               /* limb fit: its phi and ratio; else radians(slant_fix) or 0, ratio_fixe or 1 */
    double  h_first[3], h_rest[3];    /* warp row (h00, h01, h02) of the first disk / of the others (phi through degrees) */
    double  theta_first, theta_rest;
    double  circle3[3], borders4[4];  /* cercle0 ((-1,-1,-1) without a limb fit), borders */
    double  circle_out3[3];           /* the circle after the crop block */
    int64_t out_h, out_w, frame_pitch;        /* circularised frames: [out_h][out_w], rows frame_pitch elements apart */
    int64_t n_out, prod_w, prod_pitch;        /* requested disks; products [out_h][prod_w] */
    int64_t window;                   /* Savitzky-Golay window used (0: transversalium off) */
    int64_t crop4[4];                 /* crop plan (new width, source x0, destination x0, columns copied); new width 0 = none */
    int64_t disc3[3];                 /* protuberance disc x0, y0, r (r = 0: none) */
    /* byte offsets into the arena; -1 = absent */
    int64_t fit_image_off;            /* the corrected first disk when it is not a requested one */
    int64_t frames_off;               /* [n_out][out_h][frame_pitch] */
    int64_t detrans_off;              /* [n_out][out_h][frame_pitch] */
    int64_t products_off;             /* [n_out][3][out_h][prod_pitch]: final, cl1, high contrast */
    int64_t results_off;              /* into `results`: [n_out][2][out_h][prod_pitch]: protus, cc */
    size_t  needed_arena_bytes, needed_results_bytes, needed_workspace_bytes;
} shg_scan_result;

size_t shg_scan_workspace_bytes(const shg_scan_request* req);      /* phases 0-2; the rest is reported by needed_workspace_bytes */
size_t shg_scan_host_bytes(const shg_scan_request* req);           /* the whole call */
int shg_scan_file(const shg_scan_request* req, shg_scan_result* res, shg_stream_t stream);
/* savgol_coeffs(window, 3) for a window other than request.taps_window: fn(window, out[window]) returns 0, or an SHG_E_*
 * code (SHG_E_VALUE where SciPy raises ValueError). */
typedef int (*shg_savgol_taps_fn)(int64_t window, double* out);
int shg_host_set_savgol_taps(shg_savgol_taps_fn fn);

/* ==== scan pool ===================================================================================
 * The reference's Pool(4) (Solex_recon.py:30-42) as native threads: n_workers threads of the current device, thread k
 * with streams[k], run the submitted shg_scan_file requests in submission order; host_cpus (may be NULL) is where the
 * threads should run.  The request, its buffers and the result must stay alive until shg_pool_wait has returned for the
 * ticket; by then the scan's last kernel has run (the worker synchronises its stream).  shg_pool_wait: blocks until the
 * ticket is done, -> *scan_status = what shg_scan_file returned, error_buf = its message.  shg_pool_poll: 1 done, 0 not yet.
 * shg_pool_destroy runs what is queued, then joins the threads. */
typedef struct shg_pool shg_pool;
int shg_pool_create(const shg_stream_t* streams, int n_workers, const int32_t* host_cpus, int n_cpus, shg_pool** out);
int shg_pool_submit(shg_pool* pool, const shg_scan_request* req, shg_scan_result* res, int64_t* ticket);
/* The same, naming the stream the stack was produced on (0 = the null stream).  With a frame-pass lane set, the scan's pass A
 * (solex_util.py:174-188) is launched on the lane right here, behind what that stream holds now, and the worker that later runs
 * the scan finds it there: the lane does not idle while every worker is still in the chain of an earlier scan. */
int shg_pool_submit_after(shg_pool* pool, const shg_scan_request* req, shg_scan_result* res, shg_stream_t after, int64_t* ticket);
/* The pieces shg_pool_submit_after is made of (a caller with a queue of its own): start pass A of a request / of a stack ahead
 * of shg_scan_file / shg_accumulate_mean_max with the same arguments (*launched = 0: no lane, nothing done), and wait for and
 * drop a pass that nobody came to use. */
int shg_scan_prelaunch(const shg_scan_request* req, shg_stream_t after, int* launched);
int shg_pass_a_prelaunch(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px,
                         int64_t frame_stride_px, void* workspace, size_t workspace_bytes, shg_stream_t after, int* launched);
int shg_pass_a_forget(const void* workspace);
int shg_pool_poll(shg_pool* pool, int64_t ticket);
int shg_pool_wait(shg_pool* pool, int64_t ticket, int* scan_status, char* error_buf, size_t error_cap);
int shg_pool_destroy(shg_pool* pool);

/* ==== streams of a scan worker pool ==============================================================
 * The reference post-processes up to four files at once (Pool(4), Solex_recon.py:30-42).  Scans in flight share one
 * device: shg_frame_pass_lane_set names ONE stream of the current device through which pass A of every scan runs
 * (shg_accumulate_mean_max, and with it shg_stage_mean_fit / shg_scan_file) -- HBM-bound passes side by side only halve
 * each other's bandwidth; the caller's stream waits for its pass through an event.  NULL switches the lane off.
 * shg_stream_create: priority < 0 high, 0 normal, > 0 low; host_cu_mask (n_mask_words 32-bit words, bit i = CU i, may be
 * NULL) confines the stream's kernels to those CUs (hipExtStreamCreateWithCUMask; priority is ignored then). */
int shg_device_cu_count(int* out);
int shg_stream_create(int priority, const uint32_t* host_cu_mask, int n_mask_words, shg_stream_t* out);
int shg_stream_destroy(shg_stream_t stream);
int shg_frame_pass_lane_set(shg_stream_t lane);
shg_stream_t shg_frame_pass_lane_get(void);

/* ==== host control plane =======================================================================
 * The 1-D / scalar arithmetic between the kernels, restated from the reference's NumPy / SciPy calls so
 * that a pipeline stage is one call that holds no interpreter lock (several scans in flight per process).
 * Every pointer here is a HOST pointer; no GPU is needed.  The line fit follows NumPy operation by
 * operation and solves its least-squares systems with the very LAPACK routine NumPy loaded
 * (shg_host_bind_lapack: address of the ILP64 Fortran dgelsd, scipy_dgelsd_64_ in NumPy 2.x wheels), so
 * `fit` -- and with it the raw disks -- is bit-identical to the reference's on the same host. */
int shg_host_bind_lapack(void* dgelsd_ilp64);          /* NULL: fall back to a built-in Householder QR */
int shg_host_lapack_bound(void);
/* The mode of the rounded line residuals is `values[np.argpartition(-counts, kth=2)[:2][0]]` (solex_util.py:245-247):
 * one of the two most frequent values, which one being up to NumPy's selection kernel on this CPU.  A binding that
 * wants NumPy's very choice registers a picker (called with -counts, returns the index); NULL = the first most
 * frequent value (the scalar introselect's answer). */
typedef int64_t (*shg_mode_pick_fn)(const int64_t* neg_counts, int64_t n);
int shg_host_set_mode_pick(shg_mode_pick_fn pick);
/* np.polyfit(x, y, 3): coefficients, highest power first (solex_util.py:233, 238, 255; scipy _fit_edge) */
int shg_host_polyfit3(const double* host_x, const double* host_y, int64_t n, double* host_coef4);
/* detect_bord's decision on the row means of the blurred image (solex_util.py:167-172) */
int shg_host_detect_bord(const double* host_row_means, int64_t n, int64_t* lb, int64_t* ub);
/* compute_mean_return_fit from the two argmin traces on (solex_util.py:231-259): trace_blur relative to column
 * blur_offset (= 12), rows [y1, y2) fitted; p4 lowest power first, fit[ih][4] = [floor(c), c - floor(c), y, c],
 * mask_good[y2 - y1] (may be NULL).  SHG_E_VALUE when fewer than 3 distinct residuals (:246), SHG_E_TYPE on an
 * empty fit, SHG_E_LINALG when the SVD fails. */
int shg_host_line_fit(const int32_t* host_trace_blur, const int32_t* host_trace_sharp, int64_t ih, int64_t y1,
                      int64_t y2, int32_t blur_offset, double* host_p4, double* host_fit, uint8_t* host_mask_good);
/* read_video_improved's per-shift sample columns and weights (solex_util.py:113-123) */
int shg_host_column_plan(const double* host_fit, int64_t ih, int64_t iw, const int32_t* host_shifts, int n_shifts,
                         int32_t* host_ind_l, double* host_lw, double* host_rw);
/* get_flood_image's threshold from the reduced image statistics (ellipse_to_circle.py:159-225) */
int shg_host_flood_threshold(double total, int64_t h, int64_t w, double mn, double mx, const int64_t* host_counts20,
                             double* thresh_out);
/* get_edge_list after canny (ellipse_to_circle.py:251-291): region choice, convex-hull filter, row crop on the
 * labelled edge pixels (shg_edge_components' output); out_sel[m] = 1 for limb points */
int shg_host_limb_points(const int32_t* host_idx, const int32_t* host_root, int64_t m, int64_t h, int64_t w,
                         uint8_t* host_out_sel, int64_t* n_selected);
/* The limb geometry with NumPy's own BLAS / LAPACK: phi and ratio steer every sample position of the warp, so the
 * scatter matrices, 3x3 inverses and the eigen-decomposition of the ellipse fit come out of the routines NumPy calls,
 * called the way its matmul / inv / eig call them (csrc/hostmath.hip).  shg_host_bind_blas takes the ILP64 entry
 * points cblas_dgemm, cblas_dsyrk, cblas_dgemv, dgesv, dgeev of the OpenBLAS NumPy loaded; without them the
 * functions below return SHG_E_UNSUPPORTED and the caller keeps the geometry in NumPy. */
int shg_host_bind_blas(void* cblas_dgemm_ilp64, void* cblas_dsyrk_ilp64, void* cblas_dgemv_ilp64, void* dgesv_ilp64,
                       void* dgeev_ilp64);
int shg_host_blas_bound(void);
/* LsqEllipse().fit(points).as_parameters() (ellipse_to_circle.py:57-59), Halir & Flusser */
int shg_host_fit_ellipse(const double* host_points, int64_t n, double* host_center2, double* width, double* height,
                         double* phi);
/* get_correction_matrix (ellipse_to_circle.py:39-50): inverse matrix (row major) and theta */
int shg_host_correction_matrix(double phi, double r, double* host_inv4, double* theta_out);
/* two_step (ellipse_to_circle.py:62-91); host_kept[n] and outline200 = return_fit(n_points=100) may be NULL */
int shg_host_two_step(const double* host_points, int64_t n, double* host_center2, double* height_out, double* phi_out,
                      double* ratio_out, uint8_t* host_kept, int64_t* n_kept, double* host_outline200);
/* correct_image's geometry (ellipse_to_circle.py:100-122) */
int shg_host_warp_geometry(double phi, double ratio, int64_t h, int64_t w, double* host_mat3_9, double* host_inv4,
                           double* host_origin2, double* det_out, double* theta_out, int64_t* out_h, int64_t* out_w);
/* ellipse_to_circle after get_edge_list (ellipse_to_circle.py:303-314): two_step on the limb points (row, col), the geometry
 * of the corrected h x w disk, its circle and the borders of the kept points.  host_geom16 = ellipse centre x, y, height,
 * phi, ratio | circle cx, cy, r | borders[4] | mat3 row 0 (h00, h01, h02 for shg_warp_rows_u16) | theta; host_dims2 = out_h,
 * out_w; host_kept [n] and host_outline200 may be NULL. */
int shg_host_limb_geometry(const double* host_points, int64_t n, int64_t h, int64_t w, double* host_geom16,
                           int64_t* host_dims2, uint8_t* host_kept, int64_t* n_kept, double* host_outline200);
/* correct_transversalium2's chord slices (solex_util.py:384-391); xa, xb: max(y2-y1, 1) entries */
int shg_host_chord_bounds(double cx, double cy, double r, double b0, double b2, int64_t y1, int64_t y2, int64_t w,
                          int32_t* host_xa, int32_t* host_xb);
/* correction factors from the row-pair statistics (solex_util.py:400-404, 456-472); taps = savgol_coeffs(window, 3) */
int shg_host_transversalium_factors(const double* host_ratios, const double* host_interior, int64_t k, int64_t n,
                                    const double* host_taps, int64_t window, int tapered, double* host_out);
/* np.percentile's two order statistics and lerp weight; NumPy's _lerp */
int shg_host_percentile_plan(int64_t n, double q, int64_t* rank_lo, int64_t* rank_hi, double* gamma);
double shg_host_lerp(double a, double b, double gamma);

#ifdef __cplusplus
}
#endif
#endif /* SHG_HIP_H */
