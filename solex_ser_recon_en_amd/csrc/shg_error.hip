#include <stdarg.h>
#include "shg_common.h"

namespace shg {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
}  // namespace shg

// ---- event profiler ---------------------------------------------------------------
#include <mutex>
#include <string>
#include <vector>
namespace shg {
namespace {
struct Sample { std::string tag; hipEvent_t a, b; hipStream_t stream; unsigned generation; };
unsigned g_generation = 0;                // bumped by shg_profile_reset: a scope that outlives a reset drops its sample
std::mutex g_mu;
bool g_enabled = false;
std::vector<Sample> g_samples;
std::vector<std::string> g_only;      // empty = every tag
}  // namespace

ProfScope::ProfScope(const char* tag, hipStream_t st) : slot(-1), stream(st), generation(0) {
    if (!g_enabled) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_only.empty()) {
        bool hit = false;
        for (auto& t : g_only) hit = hit || t == tag;
        if (!hit) return;
    }
    Sample s;
    s.tag = tag;
    s.stream = st;
    s.generation = g_generation;
    if (hipEventCreate(&s.a) != hipSuccess) return;
    if (hipEventCreate(&s.b) != hipSuccess) { (void)hipEventDestroy(s.a); return; }
    (void)hipEventRecord(s.a, st);
    g_samples.push_back(s);
    slot = (int)g_samples.size() - 1;
    generation = g_generation;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    // another thread may have reset the profiler while this scope was open (several scan workers share it): the sample is
    // gone then, and the slot may belong to somebody else's
    if (generation != g_generation || slot >= (int)g_samples.size()) return;
    (void)hipEventRecord(g_samples[slot].b, stream);
}
}  // namespace shg

// ---- host-side section timer ------------------------------------------------------------
#include <atomic>
#include <chrono>
#include <map>
namespace shg {
namespace {
std::atomic<int> g_host_timing{0};
std::mutex g_host_mu;
std::map<std::string, std::pair<double, int64_t>> g_host_sections;
inline double host_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace
HostScope::HostScope(const char* tag_) : tag(tag_), t0(g_host_timing.load(std::memory_order_relaxed) ? host_now() : -1.0) {}
HostScope::~HostScope() {
    if (t0 < 0) return;
    const double dt = host_now() - t0;
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto& e = g_host_sections[tag];
    e.first += dt;
    e.second += 1;
}
}  // namespace shg

extern "C" int shg_host_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(shg::g_host_mu);
    if (on) shg::g_host_sections.clear();
    shg::g_host_timing.store(on ? 1 : 0);
    return 0;
}

// "tag seconds calls" lines into buf (truncated to cap - 1 characters); returns the number of sections.
extern "C" int shg_host_timing_report(char* buf, size_t cap) {
    if (!buf || cap == 0) { shg::set_error("shg_host_timing_report: null pointer"); return SHG_E_ARG; }
    std::lock_guard<std::mutex> lk(shg::g_host_mu);
    size_t off = 0;
    buf[0] = 0;
    for (auto& kv : shg::g_host_sections) {
        int n = snprintf(buf + off, cap - off, "%s %.9f %lld\n", kv.first.c_str(), kv.second.first, (long long)kv.second.second);
        if (n < 0 || (size_t)n >= cap - off) break;
        off += (size_t)n;
    }
    return (int)shg::g_host_sections.size();
}

extern "C" int shg_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(shg::g_mu);
    shg::g_enabled = on != 0;
    return 0;
}

extern "C" int shg_profile_select(const char* tags_csv) {
    std::lock_guard<std::mutex> lk(shg::g_mu);
    shg::g_only.clear();
    if (!tags_csv) return 0;
    std::string cur;
    for (const char* p = tags_csv;; ++p) {
        if (*p == ',' || *p == 0) {
            if (!cur.empty()) shg::g_only.push_back(cur);
            cur.clear();
            if (*p == 0) break;
        } else {
            cur.push_back(*p);
        }
    }
    return 0;
}

extern "C" int shg_profile_reset(void) {
    std::lock_guard<std::mutex> lk(shg::g_mu);
    for (auto& s : shg::g_samples) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    shg::g_samples.clear();
    ++shg::g_generation;
    return 0;
}

// Every sample since the last reset as "tag,stream,start_ms,stop_ms" lines (times relative to the first sample's start):
// the device's timeline as the streams saw it, without a profiler attached (tools/timeline.py).
extern "C" int shg_profile_dump(const char* path) {
    if (!path) { shg::set_error("shg_profile_dump: null pointer"); return SHG_E_ARG; }
    std::lock_guard<std::mutex> lk(shg::g_mu);
    FILE* f = fopen(path, "w");
    if (!f) { shg::set_error("shg_profile_dump: cannot write %s", path); return SHG_E_ARG; }
    fprintf(f, "tag,stream,start_ms,stop_ms\n");
    if (!shg::g_samples.empty()) {
        hipEvent_t ref = shg::g_samples[0].a;
        for (auto& s : shg::g_samples) {
            float t0 = 0.f, t1 = 0.f;
            if (hipEventSynchronize(s.b) != hipSuccess || hipEventElapsedTime(&t0, ref, s.a) != hipSuccess ||
                hipEventElapsedTime(&t1, ref, s.b) != hipSuccess) { (void)hipGetLastError(); continue; }
            fprintf(f, "%s,%p,%.4f,%.4f\n", s.tag.c_str(), (void*)s.stream, t0, t1);
        }
    }
    fclose(f);
    return 0;
}

extern "C" int shg_profile_get(const char* tag, double* total_ms, int64_t* launches) {
    if (!tag || !total_ms || !launches) { shg::set_error("shg_profile_get: null pointer"); return SHG_E_ARG; }
    std::lock_guard<std::mutex> lk(shg::g_mu);
    double ms = 0.0;
    int64_t n = 0;
    for (auto& s : shg::g_samples) {
        if (s.tag != tag) continue;
        if (hipError_t e = hipEventSynchronize(s.b)) { shg::set_error("shg_profile_get: %s", hipGetErrorString(e)); return (int)e; }
        float t = 0.f;
        if (hipError_t e = hipEventElapsedTime(&t, s.a, s.b)) { shg::set_error("shg_profile_get: %s", hipGetErrorString(e)); return (int)e; }
        ms += t;
        ++n;
    }
    *total_ms = ms;
    *launches = n;
    return 0;
}

extern "C" int shg_profile_total(double* total_ms, int64_t* launches) {
    if (!total_ms || !launches) { shg::set_error("shg_profile_total: null pointer"); return SHG_E_ARG; }
    std::lock_guard<std::mutex> lk(shg::g_mu);
    double ms = 0.0;
    for (auto& s : shg::g_samples) {
        if (hipError_t e = hipEventSynchronize(s.b)) { shg::set_error("shg_profile_total: %s", hipGetErrorString(e)); return (int)e; }
        float t = 0.f;
        if (hipError_t e = hipEventElapsedTime(&t, s.a, s.b)) { shg::set_error("shg_profile_total: %s", hipGetErrorString(e)); return (int)e; }
        ms += t;
    }
    *total_ms = ms;
    *launches = (int64_t)shg::g_samples.size();
    return 0;
}

extern "C" int shg_abi_version(void) { return SHG_ABI_VERSION; }
extern "C" const char* shg_last_error_string(void) { return shg::g_error; }
