// Shared helpers for the libshg_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include "../../include/shg_hip.h"

namespace shg {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}


// One kernel launch with its error check: k(a) on `st`; `a` is the kernel's one trivially copyable argument struct.
template <typename A>
struct same_type { using type = A; };
template <typename A>
inline int launch(void (*k)(A), dim3 grid, dim3 block, size_t lds, hipStream_t st, const typename same_type<A>::type& a, const char* what) {
    hipLaunchKernelGGL(k, grid, block, lds, st, a);
    return check_launch(what);
}

#define SHG_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            shg::set_error(__VA_ARGS__);        \
            return (code);                      \
        }                                       \
    } while (0)

inline hipStream_t as_stream(shg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// The frame-pass lane (streams.hip): launch(stream, arg) runs on the current device's lane when one is set -- `st` then
// waits for it through an event -- and on `st` itself otherwise.
int on_frame_pass_lane(hipStream_t st, int (*launch)(hipStream_t, void*), void* arg);

// Set by a stage that has already cleared the extrema's slots (stages.hip: in the launch that uploads the plan); the extraction entry point
// this thread calls next then skips its memset, and resets the flag.
extern thread_local bool t_minmax_slots_zeroed;
// A scratch area the NEXT stage accumulates into, cleared by the extraction's last (tiny) launch instead of by a launch of its own
// (every launch of a scan costs the frame pass running beside it about a microsecond): shg_scan_file names the limb stage's
// accumulators here before it calls the extraction, k_fold_minmax clears them on its way and the pointer moves to t_prezeroed,
// where shg_limb_prepare finds it (and skips its own zeroing) or shg_scan_file drops it.
extern thread_local void* t_zero_with_fold;
extern thread_local size_t t_zero_with_fold_words;
extern thread_local void* t_prezeroed;
size_t limb_prepare_zero_words(int64_t h, int64_t w);          // words shg_limb_prepare zeroes for an [h][w] disk (0: not the fused path's)
// The same launch with nobody waiting: after what `after` holds so far, *done recorded behind it (the caller's to destroy).
// -> 1 when the device has no lane (nothing launched), 0 when launched, another value on error.
int prelaunch_on_lane(hipStream_t after, int (*launch)(hipStream_t, void*), void* arg, hipEvent_t* done);

// Optional per-kernel timing with HIP events recorded on the launch stream, right around
// the launch (bench.py's roofline leg).  Disabled by default: no events, no overhead.
struct ProfScope {
    ProfScope(const char* tag, hipStream_t st);
    ~ProfScope();
    int slot;
    hipStream_t stream;
    unsigned generation;
};
#define SHG_PROF(tag, st) shg::ProfScope shg_prof_scope_(tag, st)

// Host-side wall clock of a section of a stage composite (waiting in a stream synchronise, a control-plane routine, ...),
// accumulated per tag while shg_host_timing_enable(1): where a scan worker's time goes between the kernels
// (tools/host_budget.py).  Off by default: one relaxed load.
struct HostScope {
    explicit HostScope(const char* tag);
    ~HostScope();
    const char* tag;
    double t0;
};
#define SHG_HOST_CAT2(a, b) a##b
#define SHG_HOST_CAT(a, b) SHG_HOST_CAT2(a, b)
#define SHG_HOST_TIME(tag) shg::HostScope SHG_HOST_CAT(shg_host_scope_, __LINE__)(tag)

// The images of one launch over several disks of a file (a Doppler stack: Solex_recon.py:105-133 loops over the shifts):
// blockIdx.z picks the disk, the kernel its pointers from this by-value table.  At most kMaxBatch disks per launch.
constexpr int kMaxBatch = 32;
template <int N>
struct PtrBatchN {
    const void* p[N];
    template <typename T>
    __device__ __forceinline__ T* at(int i) const { return static_cast<T*>(const_cast<void*>(p[i])); }
};
using PtrBatch = PtrBatchN<kMaxBatch>;
// (a kernel that takes several tables, or other per-disk arguments by value, takes narrower tables: 4 KB of kernel arguments in all)
template <int N, typename T>
inline PtrBatchN<N> make_batch_n(T* const* host_ptrs, int first, int count) {
    PtrBatchN<N> b = {};
    for (int i = 0; i < count && i < N; ++i) b.p[i] = host_ptrs[first + i];
    return b;
}
template <typename T>
inline PtrBatch make_batch(T* const* host_ptrs, int first, int count) { return make_batch_n<kMaxBatch>(host_ptrs, first, count); }

// One launch per kernel for the k disks of a file (host arrays of device pointers; any k, cut into launches of kMaxBatch):
// the per-disk stages of shg_stage_process_frames and the warps of shg_scan_file.  The single-image entry points of the C ABI
// are these with k = 1.
int rowpair_stats_batch(const uint16_t* const* host_imgs, int64_t k, int64_t h, int64_t w, int64_t pitch, int64_t y1, int64_t y2,
                        const int32_t* xa, const int32_t* xb, const double* row_factor, double* out, double* out_mirror, shg_stream_t stream);
int scale_rows_batch(const uint16_t* const* host_imgs, int64_t k, int64_t h, int64_t w, int64_t pitch, const double* c,
                     const double* row_factor, uint16_t* const* host_dsts, int64_t dst_pitch, shg_stream_t stream);
int crop_pad_batch(const uint16_t* const* host_srcs, int64_t k, int64_t h, int64_t w, int64_t pitch, uint16_t* const* host_dsts, int64_t nw,
                   int64_t dst_pitch, int64_t sx0, int64_t dx0, int64_t n, int32_t fill, shg_stream_t stream);
int contrast_products_batch(const uint16_t* const* host_frames, int64_t frame_pitch, const uint16_t* const* host_cl1, int64_t cl1_pitch,
                            int64_t k, int64_t h, int64_t w, const double* host_lo_hi6, uint16_t* const* host_hc, uint16_t* const* host_protus,
                            uint16_t* const* host_cc, int64_t dst_pitch, int64_t disc_x0, int64_t disc_y0, int64_t disc_r, shg_stream_t stream,
                            const double* stats5 = nullptr, double g_bright = 0.0, double g_dark = 0.0, double* mirror5 = nullptr);
int warp_rows_batch(const uint16_t* const* host_srcs, int64_t k, int64_t h, int64_t w, int64_t src_pitch, const double* host_h3,
                    uint16_t* const* host_dsts, int64_t out_h, int64_t out_w, int64_t dst_pitch, const uint32_t* const* host_minmax2,
                    shg_stream_t stream);
// from (may be NULL): host_frames do not exist yet -- the CLAHE histogram kernel makes them on its way from the frames before
// them in single_image_process: host_raw[i][y][sx0 + j] * factors[i][y] (saturated, truncated; factors NULL: as they are) ->
// host_frames[i][y][dx0 + j] for j < ncopy, host_raw[i][0][0] (scaled) elsewhere.  Only where contrast_stats_batches() says so.
struct FrameSource {
    const uint16_t* const* host_raw;
    int64_t raw_pitch;
    const double* factors;           // device-readable [k][h], or NULL
    int64_t sx0, dx0, ncopy;
};
bool contrast_stats_batches(int64_t h, int64_t w, int tiles, double clip_limit);
int contrast_stats_batch(const uint16_t* const* host_frames, int64_t k, int64_t h, int64_t w, int64_t pitch, double clip_limit, int tiles,
                         uint16_t* const* host_cl1, int64_t cl1_pitch, const int64_t* ranks_frame2, const int64_t* ranks_cl13, double* out5,
                         void* workspace, size_t workspace_bytes, shg_stream_t stream, const FrameSource* from = nullptr);

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kCUs = 256;          // MI355X

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// A 32-bit value folded over the wave with an associative, commutative `op` (min, max, packed min / max, or, add of integers: the order
// does not matter), the result in every lane -- without LDS: four DPP steps inside each row of 16 lanes (quad butterflies, the mirrored
// half, the mirrored row), then the four rows' results through scalar registers.  A __shfl_xor ladder is six DEPENDENT ds_bpermute round
// trips (~1000 cycles at the end of a workgroup's chain; round 6 found two of them to be 13 % of k_rowpair_stats).
template <typename Op>
__device__ __forceinline__ uint32_t wave_fold_u32(uint32_t v, Op op) {
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));      // quad_perm [1, 0, 3, 2]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));      // quad_perm [2, 3, 0, 1]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));     // row_half_mirror
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));     // row_mirror
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                   r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return op(op(r0, r1), op(r2, r3));
}

// The same for a 64-bit value (both halves travel together): sums of 64-bit integers, the minimum / maximum of order-preserving keys,
// an arg-min as the minimum of (value << 32 | index).
template <typename Op>
__device__ __forceinline__ uint64_t wave_fold_u64(uint64_t v, Op op) {
    auto step = [&](auto ctrl) {
        const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, decltype(ctrl)::value, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp((int)(v >> 32), (int)(v >> 32), decltype(ctrl)::value, 0xf, 0xf, false);
        v = op(v, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    };
    step(std::integral_constant<int, 0xB1>{});
    step(std::integral_constant<int, 0x4E>{});
    step(std::integral_constant<int, 0x141>{});
    step(std::integral_constant<int, 0x140>{});
    auto at = [&](int l) {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    };
    return op(op(at(0), at(16)), op(at(32), at(48)));
}
__device__ __forceinline__ uint64_t wave_sum(uint64_t v) { return wave_fold_u64(v, [](uint64_t a, uint64_t b) { return a + b; }); }
__device__ __forceinline__ uint64_t wave_min(uint64_t v) { return wave_fold_u64(v, [](uint64_t a, uint64_t b) { return a < b ? a : b; }); }
__device__ __forceinline__ uint64_t wave_max(uint64_t v) { return wave_fold_u64(v, [](uint64_t a, uint64_t b) { return a > b ? a : b; }); }

// Integer sums over the wave through the same DPP steps (exact whatever the order): the total in every lane; the total of a lane's
// group of 8 lanes; the totals of the two 32-lane halves; and the inclusive prefix sum (row shifts inside a row of 16 -- a lane without a
// source adds 0 --, then the totals of the rows before the lane's own from scalar registers).  Workgroups are one-dimensional: a wave
// is 64 consecutive threads.
template <int CTRL>
__device__ __forceinline__ int dpp_get(int v, int fallback) {                 // lane's source under CTRL, `fallback` where it has none
    return __builtin_amdgcn_update_dpp(fallback, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_sum(int v) {
    v += dpp_get<0xB1>(v, v);
    v += dpp_get<0x4E>(v, v);
    v += dpp_get<0x141>(v, v);
    v += dpp_get<0x140>(v, v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return (uint32_t)wave_sum((int)v); }
__device__ __forceinline__ int sum_of_8_lanes(int v) {                       // lanes 8 g .. 8 g + 7 all get their group's total
    v += dpp_get<0xB1>(v, v);
    v += dpp_get<0x4E>(v, v);
    v += dpp_get<0x141>(v, v);
    return v;
}
__device__ __forceinline__ uint32_t sum_of_32_lanes(uint32_t u) {            // lanes 0 .. 31 get their half's total, lanes 32 .. 63 theirs
    int v = (int)u;
    v += dpp_get<0xB1>(v, v);
    v += dpp_get<0x4E>(v, v);
    v += dpp_get<0x141>(v, v);
    v += dpp_get<0x140>(v, v);
    const int lo = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16), hi = __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
    return (uint32_t)((threadIdx.x & 32) ? hi : lo);
}
__device__ __forceinline__ int wave_scan(int v) {
    v += dpp_get<0x111>(v, 0);           // row_shr:1
    v += dpp_get<0x112>(v, 0);           // row_shr:2
    v += dpp_get<0x114>(v, 0);           // row_shr:4
    v += dpp_get<0x118>(v, 0);           // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int lane = threadIdx.x & 63;
    return v + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
}
__device__ __forceinline__ int64_t wave_scan(int64_t v) {
    auto shifted = [&](auto ctrl) {
        const int lo = dpp_get<decltype(ctrl)::value>((int)(uint32_t)v, 0), hi = dpp_get<decltype(ctrl)::value>((int)(v >> 32), 0);
        return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    };
    v += shifted(std::integral_constant<int, 0x111>{});
    v += shifted(std::integral_constant<int, 0x112>{});
    v += shifted(std::integral_constant<int, 0x114>{});
    v += shifted(std::integral_constant<int, 0x118>{});
    auto at = [&](int l) {
        return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l));
    };
    const int64_t t0 = at(15), t1 = at(31), t2 = at(47);
    const int lane = threadIdx.x & 63;
    return v + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
}

// BORDER_REFLECT_101 index (gfedcb|abcdefgh|gfedcba), valid for any i
__host__ __device__ __forceinline__ int64_t reflect101(int64_t i, int64_t n) {
    if (n == 1) return 0;
    const int64_t period = 2 * n - 2;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

}  // namespace shg
