// Scan pool: the reference's Pool(4) (Solex_recon.py:30-42) as native threads.
//
// A scan is one shg_scan_file call; a pool is W threads, each with a stream of its own, that take such calls from a
// queue in the order they were submitted.  The caller's language only builds a request (buffers, options) and, some time
// later, looks at the result: while scans are in flight no thread of the pool touches the caller's interpreter, so the
// number of scans in flight is not limited by an interpreter lock and ONE caller thread can feed any number of workers
// (measured before: four interpreter threads making the same calls spent 0.4 ms per scan waiting for each other's lock,
// tools/host_budget.py).  The pool owns no memory: every buffer travels in the request.
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "shg_common.h"

struct shg_pool {
    int device = 0;
    std::vector<hipStream_t> streams;
    std::vector<std::thread> threads;
    std::vector<int> cpus;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    struct Job {
        const shg_scan_request* rq;
        shg_scan_result* rs;
        int status = 0;
        bool done = false;
        bool ahead = false;                                   // its pass A is on the lane already
        std::string error;
    };
    std::deque<int64_t> queue;
    std::map<int64_t, Job> jobs;
    int64_t next_ticket = 1;
    bool stop = false;
};

namespace {
void pool_worker(shg_pool* p, int k) {
    (void)hipSetDevice(p->device);
    if (!p->cpus.empty()) {                                   // the caller's placement (one L3 group next to the GPU): optional
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int c : p->cpus)
            if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
        (void)sched_setaffinity(0, sizeof(set), &set);
    }
    hipStream_t st = p->streams[(size_t)k];
    for (;;) {
        int64_t ticket;
        shg_pool::Job job;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_work.wait(lk, [&] { return p->stop || !p->queue.empty(); });
            if (p->queue.empty()) return;                     // stop, and nothing left to run
            ticket = p->queue.front();
            p->queue.pop_front();
            job = p->jobs[ticket];
        }
        int status = shg_scan_file(job.rq, job.rs, reinterpret_cast<shg_stream_t>(st));
        std::string err;
        if (status != 0) err = shg_last_error_string();
        if (job.ahead) (void)shg_pass_a_forget(job.rq->workspace);   // a scan that failed before it got to its pass
        // the caller may look at every output as soon as the ticket is done: the scan's last kernels have run by then
        const hipError_t e = hipStreamSynchronize(st);
        if (status == 0 && e != hipSuccess) { status = (int)e; err = std::string("scan pool: ") + hipGetErrorString(e); }
        {
            std::lock_guard<std::mutex> lk(p->mu);
            shg_pool::Job& j = p->jobs[ticket];
            j.status = status;
            j.error = err;
            j.done = true;
        }
        p->cv_done.notify_all();
    }
}
}  // namespace

extern "C" int shg_pool_create(const shg_stream_t* streams, int n_workers, const int32_t* host_cpus, int n_cpus, shg_pool** out) {
    SHG_REQUIRE(streams && out && n_workers > 0 && n_workers <= 64, SHG_E_ARG, "shg_pool_create: bad argument");
    SHG_REQUIRE((host_cpus != nullptr) == (n_cpus > 0), SHG_E_ARG, "shg_pool_create: the cpu list and its length go together");
    shg_pool* p = new shg_pool;
    hipError_t e = hipGetDevice(&p->device);
    if (e != hipSuccess) { delete p; shg::set_error("shg_pool_create: %s", hipGetErrorString(e)); return (int)e; }
    for (int i = 0; i < n_workers; ++i) p->streams.push_back(shg::as_stream(streams[i]));
    for (int i = 0; i < n_cpus; ++i) p->cpus.push_back(host_cpus[i]);
    try {
        for (int i = 0; i < n_workers; ++i) p->threads.emplace_back(pool_worker, p, i);
    } catch (...) {
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->stop = true;
        }
        p->cv_work.notify_all();
        for (auto& t : p->threads) t.join();
        delete p;
        shg::set_error("shg_pool_create: cannot start %d threads", n_workers);
        return SHG_E_RUNTIME;
    }
    *out = p;
    return 0;
}

extern "C" int shg_pool_submit(shg_pool* p, const shg_scan_request* rq, shg_scan_result* rs, int64_t* ticket) {
    return shg_pool_submit_after(p, rq, rs, nullptr, ticket);
}

// The scan's pass A starts now, in submission order on the lane, behind what `after` (the stream the stack was produced on; 0 =
// the null stream) holds at this moment: the workers are usually all busy with the chains of earlier scans, and a lane that
// waited for one of them to get to this scan would idle.  SHG_PASS_AHEAD=0 turns that off.
extern "C" int shg_pool_submit_after(shg_pool* p, const shg_scan_request* rq, shg_scan_result* rs, shg_stream_t after, int64_t* ticket) {
    SHG_REQUIRE(p && rq && rs && ticket, SHG_E_ARG, "shg_pool_submit: null pointer");
    static const bool pass_ahead = [] { const char* v = getenv("SHG_PASS_AHEAD"); return !(v && v[0] == '0'); }();
    {
        std::lock_guard<std::mutex> lk(p->mu);                // before anything is launched for a scan nobody will run
        SHG_REQUIRE(!p->stop, SHG_E_RUNTIME, "shg_pool_submit: the pool is shutting down");
    }
    int ahead = 0;
    if (pass_ahead)
        if (int e = shg_scan_prelaunch(rq, after, &ahead)) return e;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (p->stop) {                                        // (shut down between the two locks: the pass on the lane is waited for and dropped)
            if (ahead) (void)shg_pass_a_forget(rq->workspace);
            shg::set_error("shg_pool_submit: the pool is shutting down");
            return SHG_E_RUNTIME;
        }
        *ticket = p->next_ticket++;
        shg_pool::Job j;
        j.rq = rq;
        j.rs = rs;
        j.ahead = ahead != 0;
        p->jobs[*ticket] = j;
        p->queue.push_back(*ticket);
    }
    p->cv_work.notify_one();
    return 0;
}

extern "C" int shg_pool_poll(shg_pool* p, int64_t ticket) {
    if (!p) return SHG_E_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    auto it = p->jobs.find(ticket);
    if (it == p->jobs.end()) return SHG_E_ARG;
    return it->second.done ? 1 : 0;
}

extern "C" int shg_pool_wait(shg_pool* p, int64_t ticket, int* scan_status, char* error_buf, size_t error_cap) {
    SHG_REQUIRE(p && scan_status, SHG_E_ARG, "shg_pool_wait: null pointer");
    std::unique_lock<std::mutex> lk(p->mu);
    auto it = p->jobs.find(ticket);
    SHG_REQUIRE(it != p->jobs.end(), SHG_E_ARG, "shg_pool_wait: unknown ticket %lld", (long long)ticket);
    p->cv_done.wait(lk, [&] { return p->jobs[ticket].done; });
    shg_pool::Job& j = p->jobs[ticket];
    *scan_status = j.status;
    if (error_buf && error_cap > 0) {
        strncpy(error_buf, j.error.c_str(), error_cap - 1);
        error_buf[error_cap - 1] = 0;
    }
    p->jobs.erase(ticket);
    return 0;
}

extern "C" int shg_pool_destroy(shg_pool* p) {
    if (!p) return 0;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;                                       // queued scans still run: their callers wait for them
    }
    p->cv_work.notify_all();
    for (auto& t : p->threads) t.join();
    delete p;
    return 0;
}

