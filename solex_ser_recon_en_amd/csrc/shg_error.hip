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
struct Sample { std::string tag; hipEvent_t a, b; };
std::mutex g_mu;
bool g_enabled = false;
std::vector<Sample> g_samples;
std::vector<std::string> g_only;      // empty = every tag
}  // namespace

ProfScope::ProfScope(const char* tag, hipStream_t st) : slot(-1), stream(st) {
    if (!g_enabled) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_only.empty()) {
        bool hit = false;
        for (auto& t : g_only) hit = hit || t == tag;
        if (!hit) return;
    }
    Sample s;
    s.tag = tag;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
    (void)hipEventRecord(s.a, st);
    g_samples.push_back(s);
    slot = (int)g_samples.size() - 1;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_samples[slot].b, stream);
}
}  // namespace shg

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
