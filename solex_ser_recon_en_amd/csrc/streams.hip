// Streams of a scan worker pool, and the frame-pass lane.
//
// The reference overlaps up to four files (Pool(4), Solex_recon.py:30-42).  Here every scan in flight has a stream of
// its own, and what the scans share is the device.  Two facts decide how they should share it:
//   * pass A (k_accumulate_vec, 1.6 GB at C2) is bound by HBM: two of them side by side each take twice as long,
//     nothing is gained, and every later kernel of both scans starts later;
//   * everything else of a scan is a chain of small, latency-bound kernels that leave HBM almost idle.
// So pass A of ALL scans goes through one stream -- the lane -- where they run back to back in the order the scans
// asked, each alone with the small kernels of the other scans, and the chains run on the workers' own streams
// (optionally confined to a subset of the CUs, hipExtStreamCreateWithCUMask).  A scan waits for its own pass A through
// an event (on the host by default, see on_frame_pass_lane).
#include <stdlib.h>
#include <mutex>
#include <vector>
#include "shg_common.h"

namespace shg {
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_lane_mu;                         // guards g_lanes, and keeps one scan's (launch, record) pair together
hipStream_t g_lanes[kMaxDevices] = {};
struct LaneEvents {
    hipEvent_t ev[kMaxDevices] = {};
    hipEvent_t before[kMaxDevices] = {};
    ~LaneEvents() {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : before)
            if (e) (void)hipEventDestroy(e);
    }
};
thread_local LaneEvents t_events;             // a pair of events per (thread, device): a thread has one pass A pending at most
}  // namespace

hipStream_t frame_pass_lane(int device) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    return g_lanes[device];                   // read without the lock: set once per process before the workers start
}

// Launch `launch(stream)` on the device's lane when there is one (and make `st` wait for it), else on `st`.
int on_frame_pass_lane(hipStream_t st, int (*launch)(hipStream_t, void*), void* arg) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = -1;
    hipStream_t lane = frame_pass_lane(device);
    if (!lane || lane == st) return launch(st, arg);
    hipEvent_t& ev = t_events.ev[device];
    hipEvent_t& before = t_events.before[device];
    if (!ev || !before) {
        hipError_t e = ev ? hipSuccess : hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess && !before) e = hipEventCreateWithFlags(&before, hipEventDisableTiming);
        if (e != hipSuccess) { set_error("frame-pass lane: %s", hipGetErrorString(e)); return (int)e; }
    }
    hipError_t e;
    {
        std::lock_guard<std::mutex> lk(g_lane_mu);
        // the pass reads what the caller's stream has produced so far (a stack a kernel has only just written, a workspace
        // the previous scan's last kernel still uses): the lane waits for that point of `st` -- in the scan pool, where a
        // worker synchronises its stream after every scan, there is nothing to wait for
        // (a stream with nothing pending -- the usual case -- needs no barrier on the lane: every packet between two passes
        // there is a few microseconds in which the lane, the one thing the rate of a batch is bound by, does nothing)
        if (hipStreamQuery(st) != hipSuccess) {
            (void)hipGetLastError();
            e = hipEventRecord(before, st);
            if (e == hipSuccess) e = hipStreamWaitEvent(lane, before, 0);
            if (e != hipSuccess) { set_error("frame-pass lane: %s", hipGetErrorString(e)); return (int)e; }
        }
        if (int le = launch(lane, arg)) return le;
        e = hipEventRecord(ev, lane);
    }
    // How the scan's own stream learns that its pass has run.  A barrier in the stream (hipStreamWaitEvent) costs the host
    // nothing, but the runtime multiplexes streams onto a few hardware queues (4 by default) and a barrier that waits for
    // the lane holds up every stream that shares the queue -- with more scan workers than queues, whole chains of OTHER scans.
    // Waiting on the host keeps the queue free; the stage synchronises a few kernels later anyway.
    static const bool host_wait = [] { const char* v = getenv("SHG_LANE_WAIT"); return !(v && v[0] == 's'); }();
    if (e == hipSuccess) {
        SHG_HOST_TIME("lane wait (queue + pass A)");
        if (host_wait) {
            e = hipEventSynchronize(ev);
        } else {
            e = hipStreamWaitEvent(st, ev, 0);
        }
    }
    if (e != hipSuccess) { set_error("frame-pass lane: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

// Launch on the lane without anybody waiting for it here: the pass of a scan that is still queued (shg_pass_a_prelaunch).  The lane
// first waits for what `after` has queued so far (the stack's producer, if it is still running); *done gets an event recorded
// behind the launch (the caller owns it).  -> 1 when there is no lane (nothing launched), 0 launched, else an error.
int prelaunch_on_lane(hipStream_t after, int (*launch)(hipStream_t, void*), void* arg, hipEvent_t* done) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return 1;
    hipStream_t lane = frame_pass_lane(device);
    if (!lane) return 1;
    hipEvent_t before = nullptr, ev = nullptr;
    const bool idle = hipStreamQuery(after) == hipSuccess;    // nothing pending there: no barrier on the lane (see on_frame_pass_lane)
    if (!idle) (void)hipGetLastError();
    hipError_t e = idle ? hipSuccess : hipEventCreateWithFlags(&before, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_lane_mu);
        if (!idle) {
            e = hipEventRecord(before, after);
            if (e == hipSuccess) e = hipStreamWaitEvent(lane, before, 0);
        }
        if (e == hipSuccess) {
            if (int le = launch(lane, arg)) {
                if (before) (void)hipEventDestroy(before);
                (void)hipEventDestroy(ev);
                return le;
            }
            e = hipEventRecord(ev, lane);
        }
    }
    if (before) (void)hipEventDestroy(before);               // (destroying a recorded event is fine: the wait has been queued)
    if (e != hipSuccess) {
        if (ev) (void)hipEventDestroy(ev);
        set_error("frame-pass lane: %s", hipGetErrorString(e));
        return (int)e;
    }
    *done = ev;
    return 0;
}
}  // namespace shg

extern "C" int shg_device_cu_count(int* out) {
    SHG_REQUIRE(out, SHG_E_ARG, "shg_device_cu_count: null pointer");
    int device = 0, n = 0;
    hipError_t e = hipGetDevice(&device);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) { shg::set_error("shg_device_cu_count: %s", hipGetErrorString(e)); return (int)e; }
    *out = n;
    return 0;
}

extern "C" int shg_stream_create(int priority, const uint32_t* host_cu_mask, int n_mask_words, shg_stream_t* out) {
    SHG_REQUIRE(out, SHG_E_ARG, "shg_stream_create: null pointer");
    SHG_REQUIRE((host_cu_mask == nullptr) == (n_mask_words == 0) && n_mask_words >= 0, SHG_E_ARG, "shg_stream_create: mask and its length go together");
    hipStream_t st = nullptr;
    hipError_t e;
    if (host_cu_mask) {
        bool any = false;
        for (int i = 0; i < n_mask_words; ++i) any = any || host_cu_mask[i] != 0;
        SHG_REQUIRE(any, SHG_E_ARG, "shg_stream_create: empty CU mask");
        // (There is no non-blocking variant of this call: a CU-masked stream synchronises implicitly with the null stream, which
        // the thread that feeds a scan pool and torch use by default.  The SHG_CHAIN_CUS / SHG_CHAIN_CU_MASK experiment knobs
        // therefore serialise the chains against whatever that stream holds: fine for the sweep they exist for, which found no
        // mask that pays -- tools/sweep_lane.sh -- but not a way to run production scans.)
        e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_mask_words, host_cu_mask);
    } else {
        int least = 0, greatest = 0;                              // numerically: greatest priority <= least priority
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess) {
            const int p = priority < 0 ? greatest : (priority > 0 ? least : (least + greatest) / 2);
            e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, p);
        }
    }
    if (e != hipSuccess) { (void)hipGetLastError(); shg::set_error("shg_stream_create: %s", hipGetErrorString(e)); return (int)e; }
    *out = st;
    return 0;
}

extern "C" int shg_stream_destroy(shg_stream_t stream) {
    hipStream_t st = shg::as_stream(stream);
    if (!st) return 0;
    {
        std::lock_guard<std::mutex> lk(shg::g_lane_mu);
        for (hipStream_t& l : shg::g_lanes)
            if (l == st) l = nullptr;
    }
    hipError_t e = hipStreamDestroy(st);
    if (e != hipSuccess) { shg::set_error("shg_stream_destroy: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

extern "C" int shg_frame_pass_lane_set(shg_stream_t lane) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) { shg::set_error("shg_frame_pass_lane_set: %s", hipGetErrorString(e)); return (int)e; }
    SHG_REQUIRE(device >= 0 && device < shg::kMaxDevices, SHG_E_UNSUPPORTED, "shg_frame_pass_lane_set: device %d", device);
    std::lock_guard<std::mutex> lk(shg::g_lane_mu);
    shg::g_lanes[device] = shg::as_stream(lane);
    return 0;
}

extern "C" shg_stream_t shg_frame_pass_lane_get(void) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    return shg::frame_pass_lane(device);
}
