// Shared helpers for the libshg_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kCUs = 256;          // MI355X

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// BORDER_REFLECT_101 index (gfedcb|abcdefgh|gfedcba), valid for any i
__host__ __device__ __forceinline__ int64_t reflect101(int64_t i, int64_t n) {
    if (n == 1) return 0;
    const int64_t period = 2 * n - 2;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

}  // namespace shg
