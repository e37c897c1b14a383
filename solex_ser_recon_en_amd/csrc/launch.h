// One dispatch for the same kernel of several scans.
//
// The reference post-processes up to four files at once (Pool(4), Solex_recon.py:30-42); here a scan pool runs W scans at once,
// and each scan is a chain of ~26 small kernels that leave most of the chip idle (a quarter-size limb image is 1000 workgroups of
// 256 lanes; a kernel boundary alone costs the device about a microsecond beside a frame pass).  The scans are independent, so the
// SAME kernel of different scans can share a dispatch: launch() does not launch when the calling thread belongs to a scan pool --
// it records (kernel, grid, arguments), and when every scan in flight has reached a point where it needs its results
// (stream_sync), whatever has been recorded is merged position by position: the k-th launch of scan A and the k-th launch of scan B
// become one dispatch of k_<name>_multi when they are the same kernel (combine.hip).  Nothing else changes: the stage composites,
// their order, their host control plane and every fallback stay as they are; a thread outside a pool launches at once, as before.
//
// A mergeable kernel is written as a body that takes its block index and grid as arguments (they SHADOW the builtins, so the body
// reads `blockIdx` / `gridDim` as any kernel does) and its parameters as one trivially copyable struct:
//     struct FooArgs { const uint16_t* img; int64_t h, w; ... };
//     SHG_MERGEABLE(k_foo, FooArgs, __launch_bounds__(256)) { const uint16_t* img = kargs.img; ... blockIdx.x ... }
// which defines k_foo(FooArgs) -- the plain kernel -- and k_foo_multi(Multi<FooArgs>), whose workgroups first find the sub-launch
// their linear block index falls into (a handful of scalar compares) and then run the same body with that sub-launch's grid,
// block index and arguments.  Device helpers must not read blockIdx / gridDim themselves (they would see the merged grid).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <functional>
#include <memory>
#include <vector>

namespace shg {

constexpr int kMaxMerge = 8;                 // sub-launches per dispatch at most
constexpr size_t kMultiArgBytes = 3584;      // of the 4 KB a kernel's arguments may take

template <typename A>
struct Multi {
    static_assert(sizeof(A) <= kMultiArgBytes, "argument struct too large for a merged dispatch");
    static constexpr int kCap = (int)(kMultiArgBytes / sizeof(A) > (size_t)kMaxMerge ? (size_t)kMaxMerge : kMultiArgBytes / sizeof(A));
    int n;
    uint32_t start[kMaxMerge + 1];           // first linear block of sub-launch i; start[n] = all blocks
    uint32_t gx[kMaxMerge], gy[kMaxMerge];
    A args[kCap];
};

// the sub-launch of this workgroup, its block index and its grid (called with the REAL builtins in scope)
template <typename A>
__device__ __forceinline__ int multi_locate(const Multi<A>& m, uint32_t linear, dim3& bidx, dim3& gdim) {
    int i = 0;
    while (i + 1 < m.n && linear >= m.start[i + 1]) ++i;
    const uint32_t local = linear - m.start[i];
    const uint32_t gx = m.gx[i], gy = m.gy[i];
    const uint32_t q = local / gx;
    const uint32_t by = q % gy;
    bidx = dim3(local - q * gx, by, q / gy);
    gdim = dim3(gx, gy, (m.start[i + 1] - m.start[i]) / (gx * gy));
    return i;
}

#define SHG_TPL(...) __VA_ARGS__

#define SHG_MERGEABLE(name, ArgsT, BOUNDS)                                                                          \
    __device__ __forceinline__ void name##_body(const dim3 blockIdx, const dim3 gridDim, const ArgsT& kargs);           \
    __global__ BOUNDS void name(ArgsT a) { name##_body(dim3(blockIdx.x, blockIdx.y, blockIdx.z), dim3(gridDim.x, gridDim.y, gridDim.z), a); } \
    __global__ BOUNDS void name##_multi(shg::Multi<ArgsT> m) {                                                      \
        dim3 bi, gd;                                                                                                \
        const int i = shg::multi_locate(m, blockIdx.x, bi, gd);                                                     \
        name##_body(bi, gd, m.args[i]);                                                                             \
    }                                                                                                               \
    __device__ __forceinline__ void name##_body(const dim3 blockIdx, const dim3 gridDim, const ArgsT& kargs)

// the same for a kernel template: TDECL = SHG_TPL(template <int P>), TUSE = SHG_TPL(<P>)
#define SHG_MERGEABLE_T(TDECL, TUSE, name, ArgsT, BOUNDS)                                                           \
    TDECL __device__ __forceinline__ void name##_body(const dim3 blockIdx, const dim3 gridDim, const ArgsT& kargs);     \
    TDECL __global__ BOUNDS void name(ArgsT a) { name##_body TUSE(dim3(blockIdx.x, blockIdx.y, blockIdx.z), dim3(gridDim.x, gridDim.y, gridDim.z), a); } \
    TDECL __global__ BOUNDS void name##_multi(shg::Multi<ArgsT> m) {                                                \
        dim3 bi, gd;                                                                                                \
        const int i = shg::multi_locate(m, blockIdx.x, bi, gd);                                                     \
        name##_body TUSE(bi, gd, m.args[i]);                                                                        \
    }                                                                                                               \
    TDECL __device__ __forceinline__ void name##_body(const dim3 blockIdx, const dim3 gridDim, const ArgsT& kargs)

// ---- host side ----------------------------------------------------------------------------------------------------------
struct Combiner;

struct LaunchRec {
    const void* single;                       // the plain kernel: what identifies launches that can share a dispatch
    const void* multi;
    dim3 grid, block;
    size_t lds;
    size_t arg_off, arg_bytes;                // the argument struct, in the recorder's blob
    int cap;                                  // Multi<A>::kCap
    int (*flush)(const LaunchRec* const* recs, const unsigned char* const* args, int n, hipStream_t st);
    const char* what;
};

// What one scan has launched since it last waited for the device.
struct Cohort;
struct Recorder {
    std::vector<LaunchRec> recs;
    std::vector<unsigned char> blob;
    Combiner* comb = nullptr;
    std::shared_ptr<Cohort> cohort;           // the scans this one moves in step with (combine.hip)
    hipStream_t own = nullptr;                // the scan worker's own stream: where code that launches directly still launches
    hipEvent_t ev = nullptr;                  // recorded behind the scan's last launch of a flush
    bool flushed = false, direct = false;
    int error = 0;
    char error_text[256] = {0};
};

extern thread_local Recorder* t_rec;          // set while the calling thread runs a scan for a pool with a combiner

// stream_sync: "the host needs what this scan has launched so far" -- hipStreamSynchronize(st) outside a pool; inside one, the
// scan's recorded launches are handed to the combiner and the thread sleeps until they have run.  -> 0 or an error code (set_error done)
int stream_sync(hipStream_t st, const char* who);
// A thread of a pool that is about to wait for something else (its pass A on the lane): it will not launch anything meanwhile.
void pool_wait_begin();
void pool_wait_end();
// Code that launches the plain way on its own stream (rare branches that were not made mergeable): everything recorded so far runs
// first, and what is recorded afterwards waits for that stream.  A no-op outside a pool.
int direct_launches_follow(hipStream_t st);
#define SHG_DIRECT(st)                                            \
    do {                                                          \
        if (int e_ = shg::direct_launches_follow(st)) return e_;  \
    } while (0)

int record_launch(Recorder* r, const LaunchRec& rec, const void* args);

template <typename A>
int flush_launches(const LaunchRec* const* recs, const unsigned char* const* args, int n, hipStream_t st) {
    if (n == 1) {
        A a;
        memcpy(&a, args[0], sizeof(A));
        auto fn = reinterpret_cast<void (*)(A)>(const_cast<void*>(recs[0]->single));
        hipLaunchKernelGGL(fn, recs[0]->grid, recs[0]->block, recs[0]->lds, st, a);
        return check_launch(recs[0]->what);
    }
    Multi<A> m;
    m.n = n;
    uint32_t total = 0;
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        m.start[i] = total;
        m.gx[i] = recs[i]->grid.x;
        m.gy[i] = recs[i]->grid.y;
        total += recs[i]->grid.x * recs[i]->grid.y * recs[i]->grid.z;
        memcpy(&m.args[i], args[i], sizeof(A));
        lds = recs[i]->lds > lds ? recs[i]->lds : lds;
    }
    for (int i = n; i <= kMaxMerge; ++i) m.start[i] = total;
    auto fn = reinterpret_cast<void (*)(Multi<A>)>(const_cast<void*>(recs[0]->multi));
    hipLaunchKernelGGL(fn, dim3(total), recs[0]->block, lds, st, m);
    return check_launch(recs[0]->what);
}

// Launch `single(a)` on `st` -- or, on a pool thread, record it for the combiner.
template <typename A>
int launch(void (*single)(A), void (*multi)(Multi<A>), dim3 grid, dim3 block, size_t lds, hipStream_t st, const A& a, const char* what) {
    static_assert(std::is_trivially_copyable<A>::value, "kernel arguments must be trivially copyable");
    if (Recorder* r = t_rec) {
        LaunchRec rec;
        rec.single = reinterpret_cast<const void*>(single);
        rec.multi = reinterpret_cast<const void*>(multi);
        rec.grid = grid;
        rec.block = block;
        rec.lds = lds;
        rec.arg_off = 0;
        rec.arg_bytes = sizeof(A);
        rec.cap = Multi<A>::kCap;
        rec.flush = &flush_launches<A>;
        rec.what = what;
        return record_launch(r, rec, &a);
    }
    hipLaunchKernelGGL(single, grid, block, lds, st, a);
    return check_launch(what);
}

#define SHG_LAUNCH(name, grid, block, lds, st, ...) shg::launch(name, name##_multi, grid, block, lds, st, __VA_ARGS__, #name)
// a kernel template: SHG_LAUNCH_T(k_foo, SHG_TPL(<int, true>), grid, ...)
#define SHG_LAUNCH_T(name, targs, grid, block, lds, st, ...) shg::launch(name targs, name##_multi targs, grid, block, lds, st, __VA_ARGS__, #name)

}  // namespace shg
