// Stage composites: each stage function of the reference as ONE C call.
//
//   shg_stage_mean_fit        compute_mean_return_fit     solex_util.py:191-259
//   shg_stage_extract         read_video_improved         solex_util.py:93-144
//   shg_stage_limb_points     get_edge_list               ellipse_to_circle.py:231-291 (+ downscale :299-302)
//   shg_stage_limb_fit        ellipse_to_circle's fit     ellipse_to_circle.py:294-314 (the above + two_step, geometry, borders)
//   shg_stage_process_frames  single_image_process        Solex_recon.py:136-174 (transversalium, crop, image_process)
//
// A stage launches its kernels on the caller's stream, brings the few scalars / 1-D vectors the control plane
// needs to the host through pinned memory, runs that control plane in C++ (hostmath.hip) and goes on -- the same
// kernels in the same order as the step-by-step entry points, which stay exported (tests, rare option branches).
// What changes is who sits between the kernels: no interpreter, hence no interpreter lock, so the scan workers of
// Solex_recon.solex_do_work really run side by side.  The caller owns every buffer: a device workspace and a
// pinned host staging area, both sized by the *_bytes queries, and the outputs.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "shg_common.h"

namespace {

constexpr size_t kAlign = 256;
inline size_t up(size_t b) { return (b + kAlign - 1) / kAlign * kAlign; }

struct Arena {
    char* base;
    size_t cap, off;
    Arena(void* p, size_t bytes) : base(static_cast<char*>(p)), cap(bytes), off(0) {}
    template <typename T>
    T* take(size_t count) {
        const size_t b = up(count * sizeof(T));
        if (!base || off + b > cap) return nullptr;
        T* p = reinterpret_cast<T*>(base + off);
        off += b;
        return p;
    }
};

#define STAGE_TRY(expr)            \
    do {                           \
        if (int e_ = (expr)) return e_; \
    } while (0)

// "The host needs what this stage has launched so far."
#define STAGE_SYNC(st, who)                                                \
    do {                                                                   \
        SHG_HOST_TIME("sync");                                             \
        if (hipError_t se_ = hipStreamSynchronize(st)) {                   \
            shg::set_error("%s: %s", who, hipGetErrorString(se_));         \
            return (int)se_;                                               \
        }                                                                  \
    } while (0)

// ---- host <-> device without the copy engines ---------------------------------------------------------------
// The control plane moves a few KB per stage.  hipMemcpyAsync sends those through the SDMA queues (or the runtime's
// blit kernels): ~14 extra operations per scan, and with several scan workers the SDMA path stalls every stream of the
// process for 3-20 ms now and then (measured: tools/scan_timeline.py; HSA_ENABLE_SDMA=0 removes the stalls but makes
// every copy a blit launch).  Pinned host memory is mapped into the GPU's address space, so instead: a kernel whose
// result only the host reads stores it straight into the staging area; everything else crosses with one small
// kernel of ours (k_words), which also lets the device decide how much to send (k_edge_list).  Visibility follows the
// stream: host writes before the launch are seen by the kernel, kernel stores are seen after hipStreamSynchronize.
struct WordsArgs {
    uint32_t* dst;
    const uint32_t* src;
    size_t n_words;
    uint32_t* zero;
    size_t zero_words;
};
__global__ __launch_bounds__(256) void k_words(const WordsArgs kargs) {
    uint32_t* __restrict__ dst = kargs.dst;
    const uint32_t* __restrict__ src = kargs.src;
    const size_t n_words = kargs.n_words;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// the same, and `zero_words` words at `zero` cleared on the way (a scratch area the next kernel accumulates into: one launch instead of a
// copy and a memset -- every launch is an L2 write-back and invalidate under the other scans' kernels)
__global__ __launch_bounds__(256) void k_words_zero(const WordsArgs kargs) {
    uint32_t* __restrict__ dst = kargs.dst;
    const uint32_t* __restrict__ src = kargs.src;
    const size_t n_words = kargs.n_words, zero_words = kargs.zero_words;
    uint32_t* __restrict__ zero = kargs.zero;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < zero_words; i += (size_t)gridDim.x * 256) zero[i] = 0;
}

// comp = [m | idx[n] | root[n]] on the device -> the same layout in the staging area, only the m entries in use.
__global__ __launch_bounds__(256) void k_edge_list(const int32_t* __restrict__ comp, int64_t n, int32_t* __restrict__ dst) {
    const int64_t m = comp[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        dst[1 + i] = comp[1 + i];
        dst[1 + n + i] = comp[1 + n + i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) dst[0] = (int32_t)m;
}

inline int move_words(void* dst, const void* src, size_t bytes, hipStream_t st) {
    const size_t n_words = (bytes + 3) / 4;                      // arena slots are 256-byte aligned and padded
    if (n_words == 0) return 0;
    const unsigned blocks = (unsigned)std::min<size_t>((n_words + 255) / 256, 64);
    return shg::launch(k_words, dim3(blocks), dim3(256), 0, st, WordsArgs{static_cast<uint32_t*>(dst), static_cast<const uint32_t*>(src), n_words, nullptr, 0}, "k_words");
}

// The staging area as the GPU addresses it (the same address under unified addressing).
struct Staging {
    char* host;
    char* dev;
    template <typename T>
    T* on_device(T* host_ptr) const { return reinterpret_cast<T*>(dev + (reinterpret_cast<char*>(host_ptr) - host)); }
};

inline int map_staging(void* host_pinned, Staging* out, const char* who) {
    void* d = nullptr;                                   // asked on every call (a microsecond): an address that was pinned
    hipError_t e = hipHostGetDevicePointer(&d, host_pinned, 0);      // once may belong to pageable memory by now
    if (e != hipSuccess || !d) {
        (void)hipGetLastError();
        shg::set_error("%s: host_pinned is not page-locked, GPU-mapped memory (hipHostMalloc / hipHostRegister): %s", who,
                       e != hipSuccess ? hipGetErrorString(e) : "no device address");
        return SHG_E_ARG;
    }
    out->host = static_cast<char*>(host_pinned);
    out->dev = static_cast<char*>(d);
    return 0;
}

inline int64_t slit_rows(int64_t height, int64_t width) { return width > height ? width : height; }
inline int64_t spectral_cols(int64_t height, int64_t width) { return width > height ? height : width; }

}  // namespace

// ---- compute_mean_return_fit ------------------------------------------------------------------
extern "C" size_t shg_stage_mean_fit_workspace_bytes(int64_t n_frames, int64_t height, int64_t width, int bytes_per_px) {
    if (height <= 0 || width <= 0) return 0;
    const size_t hw = (size_t)height * (size_t)width, ih = (size_t)slit_rows(height, width);
    const size_t acc = n_frames > 0 ? shg_accumulate_workspace_bytes(n_frames, height, width, bytes_per_px) : 0;
    return up(acc) + up(hw * 8) + up(hw * 2) + up(hw * 2) + up(hw * 4) + up(ih * 8) + up(2 * ih * 4) + kAlign;
}

extern "C" size_t shg_stage_mean_fit_host_bytes(int64_t height, int64_t width) {
    if (height <= 0 || width <= 0) return 0;
    const size_t ih = (size_t)slit_rows(height, width);
    return up(ih * 8) + up(2 * ih * 4);
}

extern "C" int shg_stage_mean_fit(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px,
                                  int64_t frame_stride_px, const uint64_t* sum_in, const uint16_t* max_in, int64_t n_total,
                                  uint16_t* mean_out, uint16_t* max_out, int64_t* host_y12, double* host_p4, double* host_fit,
                                  int32_t* host_trace_sharp, uint8_t* host_mask_good, void* workspace, size_t workspace_bytes,
                                  void* host_pinned, size_t host_pinned_bytes, shg_stream_t stream) {
    SHG_HOST_TIME("stage mean_fit");
    SHG_REQUIRE(mean_out && max_out && host_y12 && host_p4 && host_fit && workspace && host_pinned, SHG_E_ARG,
                "shg_stage_mean_fit: null pointer");
    SHG_REQUIRE((sum_in == nullptr) == (max_in == nullptr), SHG_E_ARG, "shg_stage_mean_fit: sum_in and max_in go together");
    SHG_REQUIRE(sum_in || stack, SHG_E_ARG, "shg_stage_mean_fit: neither a stack nor its sums");
    SHG_REQUIRE(height > 0 && width > 0 && n_total > 0, SHG_E_ARG, "shg_stage_mean_fit: empty input");
    const int64_t ih = slit_rows(height, width), iw = spectral_cols(height, width);
    const size_t hw = (size_t)height * (size_t)width;
    hipStream_t st = shg::as_stream(stream);
    Arena dev(workspace, workspace_bytes), pin(host_pinned, host_pinned_bytes);
    const size_t acc_bytes = sum_in ? 0 : shg_accumulate_workspace_bytes(n_frames, height, width, bytes_per_px);
    char* acc_ws = dev.take<char>(acc_bytes ? acc_bytes : 1);
    uint64_t* sum = dev.take<uint64_t>(hw);
    uint16_t* mx = dev.take<uint16_t>(hw);
    uint16_t* blur = dev.take<uint16_t>(hw);
    uint32_t* tmp = dev.take<uint32_t>(hw);
    double* row_means = dev.take<double>((size_t)ih);
    int32_t* traces = dev.take<int32_t>(2 * (size_t)ih);
    double* h_means = pin.take<double>((size_t)ih);
    int32_t* h_traces = pin.take<int32_t>(2 * (size_t)ih);
    SHG_REQUIRE(acc_ws && sum && mx && blur && tmp && row_means && traces, SHG_E_WORKSPACE, "shg_stage_mean_fit: workspace too small");
    SHG_REQUIRE(h_means && h_traces, SHG_E_WORKSPACE, "shg_stage_mean_fit: pinned staging area too small");
    Staging stg;
    STAGE_TRY(map_staging(host_pinned, &stg, "shg_stage_mean_fit"));
    row_means = stg.on_device(h_means);                                                       // only the host reads these: the
    traces = stg.on_device(h_traces);                                                         // kernels store them where it looks

    if (!sum_in) {                                                                            // solex_util.py:174-188
        SHG_REQUIRE(n_total == n_frames, SHG_E_ARG, "shg_stage_mean_fit: n_total %lld != %lld frames of the stack", (long long)n_total, (long long)n_frames);
        STAGE_TRY(shg_accumulate_mean_max(stack, n_frames, height, width, bytes_per_px, frame_stride_px, mean_out, max_out, acc_ws, acc_bytes, stream));
    } else {
        STAGE_TRY(shg_finalize_mean_max(sum_in, max_in, n_total, height, width, bytes_per_px, mean_out, max_out, stream));
    }
    // detect_bord(max_img, axis=1) (:223, :165-172)
    if (shg_blur_fits_fused(iw, 5)) {
        STAGE_TRY(shg_blur_row_mean_u16(max_out, ih, iw, 5, 5, row_means, stream));
    } else {
        STAGE_TRY(shg_box_blur_u16(max_out, ih, iw, 5, 5, blur, tmp, stream));
        STAGE_TRY(shg_row_mean_u16(blur, ih, iw, row_means, stream));
    }
    STAGE_SYNC(st, "shg_stage_mean_fit");
    int64_t y1, y2;
    STAGE_TRY(shg_host_detect_bord(h_means, ih, &y1, &y2));
    const int64_t clip = (int64_t)((double)(y2 - y1) * 0.05);                                 // :224-226
    y1 = std::min(ih - 1, y1 + clip);
    y2 = std::max<int64_t>(0, y2 - clip);
    host_y12[0] = y1;
    host_y12[1] = y2;
    const int blur_w = 25, blur_h = (int)((double)(y2 - y1) * 0.01);                          // :228-229
    // a zero blur height (sunlit span <= 100 rows) fails here, as cv2.blur does in the reference (:230)
    const int64_t lo = blur_w / 2, hi = iw + (-(blur_w + 1) / 2);                             // blur[:, 12:-13]  (-25 // 2 == -13)
    if (blur_h > 0 && shg_blur_fits_fused(iw, blur_h)) {
        STAGE_TRY(shg_blur_argmin_u16(mean_out, ih, iw, blur_w, blur_h, lo, hi, traces, traces + ih, stream));
    } else {
        STAGE_TRY(shg_box_blur_u16(mean_out, ih, iw, blur_w, blur_h, blur, tmp, stream));
        STAGE_TRY(shg_row_argmin_u16(blur, ih, iw, lo, hi, traces, stream));
        STAGE_TRY(shg_row_argmin_u16(mean_out, ih, iw, 0, iw, traces + ih, stream));
    }
    STAGE_SYNC(st, "shg_stage_mean_fit");
    if (host_trace_sharp) memcpy(host_trace_sharp, h_traces + ih, (size_t)ih * 4);
    return shg_host_line_fit(h_traces, h_traces + ih, ih, y1, y2, (int32_t)lo, host_p4, host_fit, host_mask_good);
}

// ---- read_video_improved -----------------------------------------------------------------------
extern "C" size_t shg_stage_extract_workspace_bytes(int64_t height, int64_t width, int n_shifts) {
    if (height <= 0 || width <= 0 || n_shifts <= 0) return 0;
    const size_t ih = (size_t)slit_rows(height, width);
    return up((size_t)n_shifts * ih * 4) + up(2 * ih * 8) + up(ih * 4);
}

// host_pinned is read by the GPU (one small copy kernel) after the call has returned: it must stay untouched until the work
// queued on `stream` up to here has run -- e.g. until the caller's next synchronisation of that stream.
extern "C" int shg_stage_extract(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px,
                                 int64_t frame_stride_px, const double* host_fit, const int32_t* host_shifts, int n_shifts,
                                 uint16_t* disks, int64_t row_pitch, int64_t plane_stride, int64_t n_cols, int64_t k_offset,
                                 int flip_x, uint32_t* minmax_slots, void* workspace, size_t workspace_bytes, void* host_pinned,
                                 size_t host_pinned_bytes, shg_stream_t stream) {
    SHG_HOST_TIME("stage extract");
    SHG_REQUIRE(stack && host_fit && host_shifts && disks && workspace && host_pinned, SHG_E_ARG, "shg_stage_extract: null pointer");
    SHG_REQUIRE(height > 0 && width > 0 && n_shifts > 0, SHG_E_ARG, "shg_stage_extract: empty input");
    const int64_t ih = slit_rows(height, width), iw = spectral_cols(height, width);
    Arena dev(workspace, workspace_bytes), pin(host_pinned, host_pinned_bytes);
    // the two arenas have the same layout: indices, weights and base columns cross in one piece
    int32_t* ind_l = dev.take<int32_t>((size_t)n_shifts * ih);
    double* w2 = dev.take<double>(2 * (size_t)ih);
    int32_t* base = dev.take<int32_t>((size_t)ih);
    int32_t* h_ind = pin.take<int32_t>((size_t)n_shifts * ih);
    double* h_w2 = pin.take<double>(2 * (size_t)ih);
    int32_t* h_base = pin.take<int32_t>((size_t)ih);
    SHG_REQUIRE(ind_l && w2 && base, SHG_E_WORKSPACE, "shg_stage_extract: workspace too small");
    SHG_REQUIRE(h_ind && h_w2 && h_base, SHG_E_WORKSPACE, "shg_stage_extract: pinned staging area too small");
    STAGE_TRY(shg_host_column_plan(host_fit, ih, iw, host_shifts, n_shifts, h_ind, h_w2, h_w2 + ih));
    // A Doppler stack of consecutive shifts: every distinct sample once (shg_extract_columns_dense: the band kernel on rotated files,
    // 72 us against the general kernel's 97 - 105 at C4's shape; k_extract_dense on un-rotated ones, 335 against 684).
    // SHG_EXT_DENSE=0: never (read per call: tools/sweep_band.py switches it inside one process).
    const int dense_mode = [] { const char* v = getenv("SHG_EXT_DENSE"); return v ? atoi(v) : 1; }();
    const bool dense = dense_mode > 0 && iw > n_shifts && shg_extract_dense_fits(host_shifts, n_shifts);
    if (dense) {
        int lo = host_shifts[0];
        for (int i = 1; i < n_shifts; ++i) lo = std::min(lo, (int)host_shifts[i]);
        for (int64_t y = 0; y < ih; ++y) {
            const double v = host_fit[y * 4] + 1.0 * (double)lo;                          // fit[:, 0] + shift, before .astype(int) and the clamps
            h_base[y] = (v == v && v > -1e9 && v < 1e9) ? (int32_t)(int64_t)v : INT32_MIN;     // (a NaN fit: every shift clamps -- the general rule)
        }
    }
    hipStream_t st = shg::as_stream(stream);
    Staging stg;
    STAGE_TRY(map_staging(host_pinned, &stg, "shg_stage_extract"));
    const char* end = dense ? reinterpret_cast<const char*>(h_base + ih) : reinterpret_cast<const char*>(h_w2 + 2 * ih);
    if (minmax_slots) {                                  // the plan goes up and the extrema's slots are cleared in one launch
        const size_t n_words = ((size_t)(end - reinterpret_cast<const char*>(h_ind)) + 3) / 4, zero_words = (size_t)n_shifts * 130;
        const unsigned blocks = (unsigned)std::min<size_t>((std::max(n_words, zero_words) + 255) / 256, 64);
        STAGE_TRY(shg::launch(k_words_zero, dim3(blocks), dim3(256), 0, st,
                             WordsArgs{reinterpret_cast<uint32_t*>(ind_l), reinterpret_cast<const uint32_t*>(stg.on_device(h_ind)), n_words, minmax_slots, zero_words}, "k_words_zero"));
        shg::t_minmax_slots_zeroed = true;               // (read and reset by the extraction entry point this thread calls next)
    } else {
        STAGE_TRY(move_words(ind_l, stg.on_device(h_ind), (size_t)(end - reinterpret_cast<const char*>(h_ind)), st));
    }
    if (dense)
        return shg_extract_columns_dense(stack, n_frames, height, width, bytes_per_px, frame_stride_px, ind_l, base, w2, w2 + ih, host_shifts, n_shifts,
                                         disks, row_pitch, plane_stride, n_cols, k_offset, flip_x, minmax_slots, stream);
    return shg_extract_columns_minmax(stack, n_frames, height, width, bytes_per_px, frame_stride_px, ind_l, w2, w2 + ih, n_shifts, disks,
                                      row_pitch, plane_stride, n_cols, k_offset, flip_x, minmax_slots, stream);
}

// ---- ellipse_to_circle: limb detection and fit --------------------------------------------------------
namespace {
constexpr int kFactor = 4;                 // downscale_local_mean(image, (4, 4)), ellipse_to_circle.py:299-301
inline int64_t small_dim(int64_t v) { return (v + kFactor - 1) / kFactor; }

// The fused kernels of limb_fused.hip (8 launches) take the stage when the blur window fits their tile; SHG_LIMB_FUSED=0
// keeps the one-kernel-per-call chain of limb.hip (23 launches), which also serves the larger windows.
inline bool limb_fused(int64_t sh, int64_t sw, int k) {
    const char* v = getenv("SHG_LIMB_FUSED");                // (asked per call: the tests run both chains in one process)
    return !(v && v[0] == '0') && k > 0 && shg_limb_fused_fits(sh, sw, k);
}

// skimage's hysteresis on the emitted LOW-mask pixels (limb_fused.hip: bit 30 of root = the pixel is in the HIGH mask): keep
// the components that hold a high pixel, in place, raster order kept.  -> pixels kept
int64_t keep_strong_components(int32_t* idx, int32_t* root, int64_t m, int64_t n) {
    thread_local std::vector<uint8_t> strong;
    if ((int64_t)strong.size() < n) strong.assign((size_t)n, 0);
    for (int64_t i = 0; i < m; ++i)
        if (root[i] & (1 << 30)) strong[(size_t)(root[i] & 0x3fffffff)] = 1;
    int64_t kept = 0;
    for (int64_t i = 0; i < m; ++i) {
        const int32_t r = root[i] & 0x3fffffff;
        if (strong[(size_t)r]) { idx[kept] = idx[i]; root[kept] = r; ++kept; }
    }
    for (int64_t i = 0; i < kept; ++i) strong[(size_t)root[i]] = 0;          // (every marked root kept at least its marking pixel)
    return kept;
}
}  // namespace

extern "C" size_t shg_stage_limb_points_workspace_bytes(int64_t h, int64_t w) {
    if (h <= 0 || w <= 0) return 0;
    const int64_t sh = small_dim(h), sw = small_dim(w);
    const size_t n = (size_t)sh * (size_t)sw;
    const int k = (int)((double)sh * 0.01);
    if (limb_fused(sh, sw, k)) return up(shg_limb_prepare_workspace_bytes(sh, sw, k)) + up(shg_limb_edges_workspace_bytes(sh, sw)) + kAlign;
    return 4 * up(n * 8) + up(std::max(shg_select_workspace_bytes(4), shg_select_keys_workspace_bytes(4))) + 2 * up(n * 4) + up(4 * 8) + up(3 * 8) + up(20 * 4) + up(256) + up(32 * 8) + 2 * up(n) +
           up(shg_canny_workspace_bytes(sh, sw)) + up(shg_edge_components_workspace_bytes(sh, sw)) + up((2 * n + 1) * 4) + kAlign;
}

extern "C" size_t shg_stage_limb_points_host_bytes(int64_t h, int64_t w) {
    if (h <= 0 || w <= 0) return 0;
    const size_t n = (size_t)small_dim(h) * (size_t)small_dim(w);
    return up(32 * 8) + up((2 * n + 1) * 4);
}

// disk: the raw uint16 disk [h][w].  host_gauss_taps: scipy's Gaussian taps for sigma = 2, 1.5, 1, 0.5 (17, 13, 9, 5
// values, packed one after the other) -- canny's retry ladder (ellipse_to_circle.py:245-256).
// out: host_points int32 [points_cap][2]: the edge pixels (row, col) of the quarter-size image in raster order;
//      host_flags [points_cap]: 1 = limb point (get_edge_list's X); host_counts2 = edge pixels, limb points.
// points_cap < the number of edge pixels: SHG_E_WORKSPACE (ceil(h/4) * ceil(w/4) always suffices).
extern "C" int shg_stage_limb_points(const uint16_t* disk, int64_t h, int64_t w, int64_t pitch, const double* host_gauss_taps,
                                     int32_t* host_points, uint8_t* host_flags, int64_t points_cap, int64_t* host_counts2,
                                     void* workspace, size_t workspace_bytes, void* host_pinned, size_t host_pinned_bytes,
                                     shg_stream_t stream) {
    SHG_REQUIRE(disk && host_gauss_taps && host_points && host_flags && host_counts2 && workspace && host_pinned, SHG_E_ARG,
                "shg_stage_limb_points: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w, SHG_E_ARG, "shg_stage_limb_points: bad image size");
    const int64_t sh = small_dim(h), sw = small_dim(w);
    const int64_t n = sh * sw;
    const int k = (int)((double)sh * 0.01);                                                  // cv2.blur kernel, :163
    if (k <= 0) {
        shg::set_error("ellipse fit: the scan needs at least 400 slit rows (cv2.blur kernel int(0.01 * h/4) = 0)");
        return SHG_E_RUNTIME;
    }
    hipStream_t st = shg::as_stream(stream);
    if (limb_fused(sh, sw, k)) {
        Arena dev(workspace, workspace_bytes), pin(host_pinned, host_pinned_bytes);
        const size_t prep_bytes = shg_limb_prepare_workspace_bytes(sh, sw, k), edge_bytes = shg_limb_edges_workspace_bytes(sh, sw);
        char* prep_ws = dev.take<char>(prep_bytes);
        char* edge_ws = dev.take<char>(edge_bytes);
        double* h_packed = pin.take<double>(32);
        int32_t* h_comp = pin.take<int32_t>(2 * (size_t)n + 1);
        SHG_REQUIRE(prep_ws && edge_ws, SHG_E_WORKSPACE, "shg_stage_limb_points: workspace too small");
        SHG_REQUIRE(h_packed && h_comp, SHG_E_WORKSPACE, "shg_stage_limb_points: pinned staging area too small");
        Staging stg;
        STAGE_TRY(map_staging(host_pinned, &stg, "shg_stage_limb_points"));
        int64_t ranks[4];
        double gamma99;
        ranks[0] = (n & 1) ? n / 2 : n / 2 - 1;                                               // np.median(blur 5x5) (:241)
        ranks[1] = n / 2;
        STAGE_TRY(shg_host_percentile_plan(n, 99.0, &ranks[2], &ranks[3], &gamma99));          // np.percentile(blurred, 99) (:165)
        const uint32_t* keys = nullptr;
        STAGE_TRY(shg_limb_prepare(disk, h, w, pitch, k, ranks, gamma99, stg.on_device(h_packed), &keys, prep_ws, prep_bytes, stream));
        STAGE_SYNC(st, "shg_stage_limb_points");
        const double median5 = (n & 1) ? h_packed[0] : (h_packed[0] + h_packed[1]) / 2;
        const double low = median5 / 10, high = low * 1.5;                                   // :241-243
        int64_t counts64[20];
        const uint32_t* hc = reinterpret_cast<const uint32_t*>(h_packed + 8);
        for (int i = 0; i < 20; ++i) counts64[i] = hc[i];
        double thresh3;
        STAGE_TRY(shg_host_flood_threshold(h_packed[4], sh, sw, h_packed[5], h_packed[6], counts64, &thresh3));
        int64_t m = 0;
        const double* taps = host_gauss_taps;
        int32_t* idx = h_comp + 1;
        int32_t* root = h_comp + 1 + n;
        for (int rung = 0;; ++rung) {                                                         // sigma = 2, 1.5, 1, 0.5
            if (rung == 4) { shg::set_error("ellipse fit: could not find any edges of the solar disk"); return SHG_E_RUNTIME; }
            const double sigma = 2.0 - 0.5 * rung;
            const int radius = (int)(4.0 * sigma + 0.5);
            STAGE_TRY(shg_limb_edges(keys, sh, sw, k, thresh3, taps, radius, low, high, stg.on_device(h_comp), edge_ws, edge_bytes, stream));
            STAGE_SYNC(st, "shg_stage_limb_points");
            const int64_t m_low = h_comp[0];
            SHG_REQUIRE(m_low >= 0 && m_low <= n, SHG_E_RUNTIME, "shg_stage_limb_points: %lld edge pixels in an image of %lld", (long long)m_low, (long long)n);
            m = keep_strong_components(idx, root, m_low, n);
            if (m > 0) break;
            taps += 2 * radius + 1;                                                           // try again with less blur (:254-256)
        }
        SHG_REQUIRE(points_cap >= m, SHG_E_WORKSPACE, "shg_stage_limb_points: %lld edge pixels, room for %lld", (long long)m, (long long)points_cap);
        int64_t n_sel = 0;
        STAGE_TRY(shg_host_limb_points(idx, root, m, sh, sw, host_flags, &n_sel));
        for (int64_t i = 0; i < m; ++i) {
            host_points[2 * i] = idx[i] / (int32_t)sw;
            host_points[2 * i + 1] = idx[i] % (int32_t)sw;
        }
        host_counts2[0] = m;
        host_counts2[1] = n_sel;
        return 0;
    }
    Arena dev(workspace, workspace_bytes), pin(host_pinned, host_pinned_bytes);
    double* small = dev.take<double>((size_t)n);
    double* blurred = dev.take<double>((size_t)n);
    double* blur5 = dev.take<double>((size_t)n);
    double* tmp = dev.take<double>((size_t)n);
    const size_t sel_bytes = std::max(shg_select_workspace_bytes(4), shg_select_keys_workspace_bytes(4));
    char* sel_ws = dev.take<char>(sel_bytes);
    uint32_t* keys_k = dev.take<uint32_t>((size_t)n);
    uint32_t* keys_5 = dev.take<uint32_t>((size_t)n);
    double* packed = dev.take<double>(32);                 // [0..3] order statistics, [4..6] flood stats, [8..17] the 20 counts
    uint32_t* counts = packed ? reinterpret_cast<uint32_t*>(packed + 8) : nullptr;
    char* flood_ws = dev.take<char>(256);
    uint8_t* low_mask = dev.take<uint8_t>((size_t)n);
    uint8_t* high_mask = dev.take<uint8_t>((size_t)n);
    const size_t canny_bytes = shg_canny_workspace_bytes(sh, sw), cc_bytes = shg_edge_components_workspace_bytes(sh, sw);
    char* canny_ws = dev.take<char>(canny_bytes);
    char* cc_ws = dev.take<char>(cc_bytes);
    int32_t* comp = dev.take<int32_t>(2 * (size_t)n + 1);  // [count | idx[n] | root[n]]
    double* h_packed = pin.take<double>(32);
    int32_t* h_comp = pin.take<int32_t>(2 * (size_t)n + 1);
    SHG_REQUIRE(small && blurred && blur5 && tmp && sel_ws && keys_k && keys_5 && packed && counts && flood_ws && low_mask && high_mask && canny_ws && cc_ws && comp,
                SHG_E_WORKSPACE, "shg_stage_limb_points: workspace too small");
    SHG_REQUIRE(h_packed && h_comp, SHG_E_WORKSPACE, "shg_stage_limb_points: pinned staging area too small");
    Staging stg;
    STAGE_TRY(map_staging(host_pinned, &stg, "shg_stage_limb_points"));

    STAGE_TRY(shg_downscale_mean_u16(disk, h, w, pitch, kFactor, small, stream));
    // np.median(blur 5x5) (:241) and np.percentile(blurred, 99) (:165): their order statistics
    int64_t ranks[4];
    double gamma99;
    ranks[0] = (n & 1) ? n / 2 : n / 2 - 1;
    ranks[1] = n / 2;
    STAGE_TRY(shg_host_percentile_plan(n, 99.0, &ranks[2], &ranks[3], &gamma99));
    if (k <= 63) {
        // the block means are whole numbers of 2^-20: select on the integer window sums (three passes instead of eight)
        STAGE_TRY(shg_box_blur_key_f64(small, sh, sw, k, blurred, keys_k, tmp, stream));
        if (k == 5) {                                    // 2000-2399 slit rows: cv2.blur(img, (k, k)) is the 5 x 5 blur itself
            blur5 = blurred;
            keys_5 = keys_k;
        } else {
            STAGE_TRY(shg_box_blur_key_f64(small, sh, sw, 5, blur5, keys_5, tmp, stream));
        }
        const uint32_t* karr[4] = {keys_5, keys_5, keys_k, keys_k};
        const int kk[4] = {5, 5, k, k};
        STAGE_TRY(shg_select_keys_u32(karr, n, ranks, kk, 4, packed, sel_ws, sel_bytes, stream));
    } else {
        STAGE_TRY(shg_box_blur_f64(small, sh, sw, k, blurred, tmp, stream));
        STAGE_TRY(shg_box_blur_f64(small, sh, sw, 5, blur5, tmp, stream));      // (k > 63 here, never 5)
        const double* arrays[4] = {blur5, blur5, blurred, blurred};
        STAGE_TRY(shg_select_multi_f64(arrays, n, ranks, 4, packed, sel_ws, sel_bytes, stream));
    }
    STAGE_TRY(shg_flood_stats_lerp_f64(small, blurred, n, packed + 2, gamma99, packed + 4, counts, flood_ws, stream));
    STAGE_TRY(move_words(stg.on_device(h_packed), packed, 8 * 8 + 20 * 4, st));
    STAGE_SYNC(st, "shg_stage_limb_points");
    const double median5 = (n & 1) ? h_packed[0] : (h_packed[0] + h_packed[1]) / 2;
    const double low = median5 / 10, high = low * 1.5;                                       // :241-243
    int64_t counts64[20];
    const uint32_t* hc = reinterpret_cast<const uint32_t*>(h_packed + 8);
    for (int i = 0; i < 20; ++i) counts64[i] = hc[i];
    double thresh3;
    STAGE_TRY(shg_host_flood_threshold(h_packed[4], sh, sw, h_packed[5], h_packed[6], counts64, &thresh3));

    int64_t m = 0;
    const double* taps = host_gauss_taps;
    for (int rung = 0;; ++rung) {                                                             // sigma = 2, 1.5, 1, 0.5
        if (rung == 4) { shg::set_error("ellipse fit: could not find any edges of the solar disk"); return SHG_E_RUNTIME; }
        const double sigma = 2.0 - 0.5 * rung;
        const int radius = (int)(4.0 * sigma + 0.5);
        STAGE_TRY(shg_canny_masks_f64(blurred, sh, sw, thresh3, taps, radius, low, high, low_mask, high_mask, canny_ws, canny_bytes, stream));
        STAGE_TRY(shg_edge_components(low_mask, high_mask, sh, sw, comp + 1, comp + 1 + n, comp, cc_ws, cc_bytes, stream));
        k_edge_list<<<16, 256, 0, st>>>(comp, n, stg.on_device(h_comp));                      // the device knows how many: one trip
        STAGE_TRY(shg::check_launch("k_edge_list"));
        STAGE_SYNC(st, "shg_stage_limb_points");
        m = h_comp[0];
        SHG_REQUIRE(m >= 0 && m <= n, SHG_E_RUNTIME, "shg_stage_limb_points: %lld edge pixels in an image of %lld", (long long)m, (long long)n);
        if (m > 0) break;
        taps += 2 * radius + 1;                                                               // try again with less blur (:254-256)
    }
    const int32_t* idx = h_comp + 1;
    const int32_t* root = h_comp + 1 + n;
    SHG_REQUIRE(points_cap >= m, SHG_E_WORKSPACE, "shg_stage_limb_points: %lld edge pixels, room for %lld", (long long)m, (long long)points_cap);
    int64_t n_sel = 0;
    STAGE_TRY(shg_host_limb_points(idx, root, m, sh, sw, host_flags, &n_sel));
    for (int64_t i = 0; i < m; ++i) {
        host_points[2 * i] = idx[i] / (int32_t)sw;
        host_points[2 * i + 1] = idx[i] % (int32_t)sw;
    }
    host_counts2[0] = m;
    host_counts2[1] = n_sel;
    return 0;
}

// ellipse_to_circle without its warp (ellipse_to_circle.py:294-314): shg_stage_limb_points, then the two-step ellipse
// fit, correct_image's geometry and the borders (shg_host_limb_geometry; NumPy's own BLAS / LAPACK routines).
// host_points / host_flags as shg_stage_limb_points, flag bit 1 = kept by two_step; host_counts3 = edge pixels, limb
// points, kept points; host_geom16 / host_dims2 / host_outline200 as shg_host_limb_geometry.
extern "C" int shg_stage_limb_fit(const uint16_t* disk, int64_t h, int64_t w, int64_t pitch, const double* host_gauss_taps,
                                  int32_t* host_points, uint8_t* host_flags, int64_t points_cap, int64_t* host_counts3,
                                  double* host_geom16, int64_t* host_dims2, double* host_outline200, void* workspace,
                                  size_t workspace_bytes, void* host_pinned, size_t host_pinned_bytes, shg_stream_t stream) {
    SHG_HOST_TIME("stage limb_fit");
    SHG_REQUIRE(host_counts3 && host_geom16 && host_dims2, SHG_E_ARG, "shg_stage_limb_fit: null pointer");
    STAGE_TRY(shg_stage_limb_points(disk, h, w, pitch, host_gauss_taps, host_points, host_flags, points_cap, host_counts3, workspace,
                                    workspace_bytes, host_pinned, host_pinned_bytes, stream));
    const int64_t m = host_counts3[0], n_sel = host_counts3[1];
    std::vector<double> X((size_t)std::max<int64_t>(n_sel, 1) * 2);
    std::vector<uint8_t> kept((size_t)std::max<int64_t>(n_sel, 1));
    for (int64_t i = 0, j = 0; i < m; ++i)
        if (host_flags[i]) {                                                  // down-scaled, then upscaled back (:301-302)
            X[2 * j] = (double)host_points[2 * i] * kFactor;
            X[2 * j + 1] = (double)host_points[2 * i + 1] * kFactor;
            ++j;
        }
    STAGE_TRY(shg_host_limb_geometry(X.data(), n_sel, h, w, host_geom16, host_dims2, kept.data(), &host_counts3[2], host_outline200));
    for (int64_t i = 0, j = 0; i < m; ++i)
        if (host_flags[i]) { if (kept[j]) host_flags[i] |= 2; ++j; }
    return 0;
}

// ---- single_image_process for the requested disks of one file --------------------------------------------
extern "C" size_t shg_stage_process_workspace_bytes(int64_t k, int64_t h, int64_t w, int64_t crop_w, int tiles) {
    if (k <= 0 || h <= 0 || w <= 0) return 0;
    const size_t pitch = ((size_t)std::max(w, crop_w) + 63) / 64 * 64;
    return 2 * up((size_t)h * 4) + 2 * up((size_t)k * h * 8) + up((size_t)k * h * 8) + up(4096 * 8) +
           (size_t)k * up((size_t)h * pitch * 2) + (size_t)k * up(shg_contrast_stats_workspace_bytes_for(h, crop_w > 0 ? crop_w : w, tiles)) + up((size_t)k * 5 * 8) + kAlign;
}

extern "C" size_t shg_stage_process_host_bytes(int64_t k, int64_t h) {
    if (k <= 0 || h <= 0) return 0;
    return 2 * up((size_t)h * 4) + 2 * up((size_t)k * h * 8) + up((size_t)k * h * 8) + up((size_t)k * 5 * 8) + up(4096 * 8);
}

// host_frames[k]: device pointers of the circularised uint16 frames [h][w] (rows `pitch` elements apart).
// transversalium != 0: correct_transversalium2 with `circle` / `borders` as the caller resolved them (Solex_recon.py:
//   142-148: the limb circle, or (0, 0, 99999) with the backup bounds); host_taps[window] = savgol_coeffs(window, 3) for
//   window = min(trans_strength, (y2-y1)//2*2-1).  host_factors (may be NULL): float64 [k][h], the row factors c.
// crop_w > 0: the crop / pad block (Solex_recon.py:155-171) as dst[:, dx0:dx0+ncopy] = src[:, sx0:sx0+ncopy], fill img[0,0].
// Outputs per frame i (device pointers in host arrays, images [h][out_w] with rows out_pitch apart, out_w = crop_w or w):
//   host_detrans[i]  (array may be NULL) the frame after the transversalium stage, [h][w], rows detrans_pitch apart
//   host_final[i]    the frame image_process works on ("uncontrasted")
//   host_cl1[i], host_hc[i], host_protus[i], host_cc[i]: CLAHE image, high contrast, protus, contrasted CLAHE.
// disc_r > 0: the filled disc of value 80 on protus (solex_util.py:542-547).
extern "C" int shg_stage_process_frames(const uint16_t* const* host_frames, int64_t k, int64_t h, int64_t w, int64_t pitch,
                                        int transversalium, const double* host_circle3, const double* host_borders4,
                                        const double* host_taps, int64_t window, double* host_factors,
                                        int64_t crop_w, int64_t sx0, int64_t dx0, int64_t ncopy,
                                        double clip_limit, int tiles, int64_t disc_x0, int64_t disc_y0, int64_t disc_r,
                                        uint16_t* const* host_detrans, int64_t detrans_pitch,
                                        uint16_t* const* host_final, uint16_t* const* host_cl1, uint16_t* const* host_hc,
                                        uint16_t* const* host_protus, uint16_t* const* host_cc, int64_t out_pitch,
                                        void* workspace, size_t workspace_bytes, void* host_pinned, size_t host_pinned_bytes,
                                        shg_stream_t stream) {
    SHG_HOST_TIME("stage process_frames");
    SHG_REQUIRE(host_frames && host_final && host_cl1 && host_hc && host_protus && host_cc && workspace && host_pinned, SHG_E_ARG,
                "shg_stage_process_frames: null pointer");
    SHG_REQUIRE(k > 0 && h > 0 && w > 0 && pitch >= w, SHG_E_ARG, "shg_stage_process_frames: bad image size");
    const int64_t out_w = crop_w > 0 ? crop_w : w;
    SHG_REQUIRE(out_pitch >= out_w, SHG_E_ARG, "shg_stage_process_frames: out_pitch < output width");
    hipStream_t st = shg::as_stream(stream);
    Arena dev(workspace, workspace_bytes), pin(host_pinned, host_pinned_bytes);
    int32_t* xa = dev.take<int32_t>((size_t)h);                                                // xa | xb | taps: the same order in
    int32_t* xb = dev.take<int32_t>((size_t)h);                                                // the staging area, they cross together
    double* taps_d = dev.take<double>(4096);
    double* stats = dev.take<double>((size_t)k * h);
    double* interior = dev.take<double>((size_t)k * h);
    double* factors = dev.take<double>((size_t)k * h);
    const size_t tpitch = ((size_t)w + 63) / 64 * 64;
    std::vector<uint16_t*> scaled((size_t)k, nullptr);
    // The row scaling and the crop / pad block ride on CLAHE's histogram kernel (k_tile_hist16_slices<true>: it forms the pixels of
    // the image it counts) wherever the contrast stage takes its batched route and nobody wants the scaled frame itself
    // (save_fit's _detransversaliumed.fits): two launches and two image passes less per disk.  SHG_FUSE_SCALE=0: the separate kernels.
    const char* fuse_env = getenv("SHG_FUSE_SCALE");            // (asked per call: the tests run both ways in one process)
    const bool fuse_on = !(fuse_env && fuse_env[0] == '0');
    const bool fuse = fuse_on && !host_detrans && shg::contrast_stats_batches(h, crop_w > 0 ? crop_w : w, tiles, clip_limit) && pitch >= w;
    const bool need_tmp = transversalium && crop_w > 0 && !host_detrans && !fuse;
    if (need_tmp)
        for (int64_t i = 0; i < k; ++i) scaled[i] = dev.take<uint16_t>((size_t)h * tpitch);
    const size_t cs_each = up(shg_contrast_stats_workspace_bytes_for(h, out_w, tiles));      // one area per disk: all disks go through a kernel together
    SHG_REQUIRE(cs_each != 0, SHG_E_ARG, "shg_stage_process_frames: unsupported tile count %d", tiles);
    const size_t cs_bytes = (size_t)k * cs_each;
    char* cs_ws = dev.take<char>(cs_bytes);
    double* out5 = dev.take<double>((size_t)k * 5);
    int32_t* h_xa = pin.take<int32_t>((size_t)h);
    int32_t* h_xb = pin.take<int32_t>((size_t)h);
    double* h_taps = pin.take<double>(4096);
    double* h_stats = pin.take<double>((size_t)k * h);
    double* h_interior = pin.take<double>((size_t)k * h);
    double* h_factors = pin.take<double>((size_t)k * h);
    double* h_out5 = pin.take<double>((size_t)k * 5);
    SHG_REQUIRE(xa && xb && stats && interior && factors && taps_d && cs_ws && out5 && (!need_tmp || scaled[k - 1]), SHG_E_WORKSPACE,
                "shg_stage_process_frames: workspace too small");
    SHG_REQUIRE(h_xa && h_xb && h_stats && h_interior && h_factors && h_out5 && h_taps, SHG_E_WORKSPACE, "shg_stage_process_frames: pinned staging area too small");
    Staging stg;
    STAGE_TRY(map_staging(host_pinned, &stg, "shg_stage_process_frames"));

    // ---- correct_transversalium2 (solex_util.py:383-516) ----
    std::vector<const uint16_t*> cur((size_t)k);
    int64_t cur_pitch = pitch;
    for (int64_t i = 0; i < k; ++i) cur[i] = host_frames[i];
    if (transversalium) {
        SHG_REQUIRE(host_circle3 && host_borders4 && host_taps, SHG_E_ARG, "shg_stage_process_frames: transversalium needs circle, borders, taps");
        const double cx = host_circle3[0], cy = host_circle3[1], r = host_circle3[2];
        const int64_t y1 = (int64_t)ceil(std::max(cy - r, host_borders4[1])), y2 = (int64_t)floor(std::min(cy + r, host_borders4[3]));
        int64_t n = 1;
        bool have_rows = y2 - y1 >= 1;
        const double* use_interior = nullptr;
        if (have_rows) {
            SHG_REQUIRE(y1 >= 0 && y2 <= h, SHG_E_ARG, "shg_stage_process_frames: rows [%lld, %lld) outside the image", (long long)y1, (long long)y2);
            n = y2 - y1;
            STAGE_TRY(shg_host_chord_bounds(cx, cy, r, host_borders4[0], host_borders4[2], y1, y2, w, h_xa, h_xb));
            const bool gpu_interior = window > 3 && window <= n && window / 2 <= 1024 && window <= 4096;
            std::vector<double> rev;
            if (gpu_interior) {
                rev.resize((size_t)window);
                for (int64_t i = 0; i < window; ++i) rev[i] = host_taps[window - 1 - i];
                memcpy(h_taps, rev.data(), (size_t)window * 8);
            }
            const char* plan_end = gpu_interior ? reinterpret_cast<const char*>(h_taps + window) : reinterpret_cast<const char*>(h_xb + n);
            STAGE_TRY(move_words(xa, stg.on_device(h_xa), (size_t)(plan_end - reinterpret_cast<const char*>(h_xa)), st));
            STAGE_TRY(shg::rowpair_stats_batch(host_frames, k, h, w, pitch, y1, y2, xa, xb, nullptr, stats, stg.on_device(h_stats), stream));
            if (gpu_interior) {
                // the interior of the Savitzky-Golay trend while the statistics are still on the GPU (SciPy's own order of operations)
                const int radius = (int)(window / 2);
                int sym = 1, anti = 1;
                for (int i = 1; i <= radius; ++i) {
                    if (fabs(rev[radius + i] - rev[radius - i]) > 2.220446049250313e-16) sym = 0;
                    if (fabs(rev[radius + i] + rev[radius - i]) > 2.220446049250313e-16) anti = 0;
                }
                // only the host reads the interior: the kernel stores it in the staging area
                STAGE_TRY(shg_correlate1d_rows_f64(stats, k, n, taps_d, radius, sym ? 1 : (anti ? -1 : 0), stg.on_device(h_interior), stream));
                use_interior = h_interior;
            }
            STAGE_SYNC(st, "shg_stage_process_frames");
        } else {
            for (int64_t i = 0; i < k; ++i) h_stats[i] = 0.0;                                 // y_ratios_r = [0], :386
        }
        std::vector<double> corr((size_t)k * n);
        STAGE_TRY(shg_host_transversalium_factors(h_stats, use_interior, k, n, host_taps, window, 1, corr.data()));
        for (int64_t i = 0; i < k; ++i) {                                                     // c = ones(h); c[y1:y2] = correction_t (:478-479)
            double* c = h_factors + i * h;
            for (int64_t y = 0; y < h; ++y) c[y] = 1.0;
            if (have_rows) memcpy(c + y1, corr.data() + i * n, (size_t)n * 8);
        }
        if (host_factors) memcpy(host_factors, h_factors, (size_t)k * h * 8);
        // k_scale_rows reads one factor per workgroup: straight from the staging area (a copy kernel first: 3.8 us for the launch,
        // 2.8 us saved in the reader)
        factors = stg.on_device(h_factors);
      if (!fuse) {
        std::vector<uint16_t*> dsts((size_t)k);
        int64_t dpitch = out_pitch;
        for (int64_t i = 0; i < k; ++i) {
            if (host_detrans) { dsts[i] = host_detrans[i]; dpitch = detrans_pitch; }
            else if (crop_w > 0) { dsts[i] = scaled[i]; dpitch = (int64_t)tpitch; }
            else { dsts[i] = host_final[i]; dpitch = out_pitch; }
            SHG_REQUIRE(dsts[i] && dpitch >= w, SHG_E_ARG, "shg_stage_process_frames: bad de-transversalium output");
            cur[i] = dsts[i];
        }
        cur_pitch = dpitch;
        STAGE_TRY(shg::scale_rows_batch(host_frames, k, h, w, pitch, factors, nullptr, dsts.data(), dpitch, stream));
      }
    }
    // ---- crop / pad (Solex_recon.py:155-171) ----
    if (fuse) {
        // (done by the histogram kernel below)
    } else if (crop_w > 0) {
        STAGE_TRY(shg::crop_pad_batch(cur.data(), k, h, w, cur_pitch, host_final, crop_w, out_pitch, sx0, dx0, ncopy, -1, stream));
    } else if (cur[0] != host_final[0]) {
        // no transversalium into host_final and no crop: the frame itself is the image to contrast; copy it (a crop of
        // the full width) so that every product lives in the caller's output block
        STAGE_TRY(shg::crop_pad_batch(cur.data(), k, h, w, cur_pitch, host_final, w, out_pitch, 0, 0, w, -1, stream));
    }
    // ---- image_process (solex_util.py:527-547) ----
    const int64_t n_px = h * out_w;
    int64_t ranks_frame[2], ranks_cl1[3];
    double g_bright, g_dark;
    STAGE_TRY(shg_host_percentile_plan(n_px, 99.9999, &ranks_frame[0], &ranks_frame[1], &g_bright));
    STAGE_TRY(shg_host_percentile_plan(n_px, 10.0, &ranks_cl1[0], &ranks_cl1[1], &g_dark));
    ranks_cl1[2] = n_px - 1;                                                                  // np.max
    shg::FrameSource from = {host_frames, pitch, transversalium ? factors : nullptr, crop_w > 0 ? sx0 : 0, crop_w > 0 ? dx0 : 0, crop_w > 0 ? ncopy : w};
    STAGE_TRY(shg::contrast_stats_batch(host_final, k, h, out_w, out_pitch, clip_limit, tiles, host_cl1, out_pitch, ranks_frame, ranks_cl1, out5, cs_ws,
                                        cs_bytes, stream, fuse ? &from : nullptr));
    // The three rescales and the disc follow without a word from the host: the kernel forms the six bounds from the order
    // statistics where they lie (and leaves a copy where the host reads them).  What the host still owes the reference is
    // rescale_brightness's assert (solex_util.py:521): checked once the products kernel has run.
    STAGE_TRY(shg::contrast_products_batch(host_final, out_pitch, host_cl1, out_pitch, k, h, out_w, nullptr, host_hc, host_protus, host_cc, out_pitch,
                                           disc_x0, disc_y0, disc_r, stream, out5, g_bright, g_dark, stg.on_device(h_out5)));
    STAGE_SYNC(st, "shg_stage_process_frames");
    // A tile grid that does not divide the image and saturated histograms: the frame's percentile could not be told from the pixels of
    // the reflected border where more than a handful of the very brightest were mirrored in it (hist_rank_top_job) -- NaN.  Those disks
    // take the select over the image and their products again.
    bool again = false;
    for (int64_t i = 0; i < k; ++i) {
        const double* s = h_out5 + i * 5;
        if (s[0] == s[0] && s[1] == s[1]) continue;
        again = true;
        SHG_HOST_TIME("frame percentile selected over the image");      // (tests read the count of these)
        STAGE_TRY(shg_select_u16(host_final[i], h, out_w, out_pitch, ranks_frame, 2, out5 + 5 * i, cs_ws + (size_t)i * cs_each, cs_each, stream));
        STAGE_TRY(shg::contrast_products_batch(host_final + i, out_pitch, host_cl1 + i, out_pitch, 1, h, out_w, nullptr, host_hc + i, host_protus + i, host_cc + i,
                                               out_pitch, disc_x0, disc_y0, disc_r, stream, out5 + 5 * i, g_bright, g_dark, stg.on_device(h_out5 + 5 * i)));
    }
    if (again) STAGE_SYNC(st, "shg_stage_process_frames");
    for (int64_t i = 0; i < k; ++i) {
        const double* s = h_out5 + i * 5;
        const double bright = shg_host_lerp(s[0], s[1], g_bright);                            // basically the same as max
        const double dark_clahe = shg_host_lerp(s[2], s[3], g_dark);
        const double bright_clahe = (double)(int64_t)s[4];
        if (!(65535 >= bright && bright > bright * 0.25 && 65535 >= bright * 0.18 && bright * 0.18 > 0 && 65535 >= bright_clahe &&
              bright_clahe > dark_clahe)) {
            shg::set_error("rescale_brightness: assert sat >= hi > lo (bright %g, clahe %g .. %g)", bright, dark_clahe, bright_clahe);
            return SHG_E_ASSERT;
        }
    }
    return 0;
}
