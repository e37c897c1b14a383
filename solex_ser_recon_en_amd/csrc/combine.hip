// The launch combiner of a scan pool (launch.h says why): scans in flight record their kernel launches instead of making them,
// and the same kernel of several scans goes to the device as ONE dispatch.
//
// Two things make scans meet at the same kernel:
//   * the gate -- scans reach their chains one pass A apart (a quarter of a millisecond); a scan whose pass has finished waits
//     until `group` of them stand together (SHG_COMBINE_GROUP, default 4), for SHG_COMBINE_GATE_US (default 120) at most, and not
//     at all when no other scan is running.  The scans released together are a COHORT;
//   * the cohort's rhythm -- a cohort's launches are flushed when every scan of it that is still running has reached a point where
//     it needs its results (stream_sync): the k-th launch of each, merged position by position, on one of the combiner's
//     streams; the scans wake together, do their host work side by side (each on its pool thread) and meet again at the next
//     point.  A scan that has waited SHG_COMBINE_WAIT_US (default 150) for the others flushes what the cohort has: one slow host
//     computation does not hold the others' kernels back for longer than that.
// Cohorts are independent of each other (own pending list, the streams taken in turn), a lone scan is a cohort of one and is
// flushed the moment it waits: the combiner costs it a mutex.
#include <stdlib.h>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include "shg_common.h"

namespace shg {

thread_local Recorder* t_rec = nullptr;

struct Cohort {
    int members = 0;                          // scans of the cohort that are still running
    std::vector<Recorder*> pending;           // those that have posted their launches and wait
};

struct Combiner {
    std::mutex mu;
    std::condition_variable cv;
    static constexpr int kStreams = 4;
    hipStream_t streams[kStreams] = {};
    int turn = 0;
    long wait_us = 150;
    int on_lane = 0, group = 4, busy = 0;
    std::vector<Recorder*> at_gate;
    unsigned long long gate_generation = 0;
    long gate_us = 120, gate_hard_us = 4000;  // a scan waits at the gate for the scans still on the lane (4 ms at most), gate_us when there are none
    // statistics (shg_pool_combiner_stats): launches recorded, dispatches made, flushes
    unsigned long long n_recorded = 0, n_dispatches = 0, n_flushes = 0;
};

namespace {

void fail(Recorder* r, int code) {
    if (r->error == 0) {
        r->error = code;
        const char* msg = shg_last_error_string();
        strncpy(r->error_text, msg ? msg : "", sizeof(r->error_text) - 1);
    }
}

// mu held.  What the cohort's waiting scans have recorded goes to one of the combiner's streams, merged position by position.
void flush_locked(Combiner* c, Cohort* co) {
    if (co->pending.empty()) return;
    ++c->n_flushes;
    hipStream_t st = c->streams[c->turn];
    c->turn = (c->turn + 1) % Combiner::kStreams;
    std::vector<Recorder*>& pend = co->pending;
    const size_t np = pend.size();
    std::vector<size_t> cursor(np, 0);
    const LaunchRec* group[kMaxMerge];
    const unsigned char* gargs[kMaxMerge];
    size_t members[kMaxMerge];
    for (;;) {
        size_t first = np;
        for (size_t p = 0; p < np; ++p)
            if (pend[p]->error == 0 && cursor[p] < pend[p]->recs.size()) { first = p; break; }
        if (first == np) break;
        const LaunchRec& head = pend[first]->recs[cursor[first]];
        int n = 0;
        uint64_t blocks = 0;
        for (size_t p = first; p < np && n < head.cap && n < kMaxMerge; ++p) {
            Recorder* r = pend[p];
            if (r->error != 0 || cursor[p] >= r->recs.size()) continue;
            const LaunchRec& q = r->recs[cursor[p]];
            if (q.single != head.single || q.block.x != head.block.x || q.block.y != head.block.y || q.block.z != head.block.z) continue;
            const uint64_t b = (uint64_t)q.grid.x * q.grid.y * q.grid.z;
            if (blocks + b >= (1ull << 31)) continue;
            blocks += b;
            group[n] = &q;
            gargs[n] = r->blob.data() + q.arg_off;
            members[n] = p;
            ++n;
        }
        ++c->n_dispatches;
        const int e = head.flush(group, gargs, n, st);
        for (int i = 0; i < n; ++i) {
            Recorder* r = pend[members[i]];
            if (e != 0) fail(r, e);
            if (++cursor[members[i]] == r->recs.size() && r->error == 0) {
                const hipError_t he = hipEventRecord(r->ev, st);
                if (he != hipSuccess) { set_error("launch combiner: %s", hipGetErrorString(he)); fail(r, (int)he); }
            }
        }
    }
    for (Recorder* r : pend) {
        r->recs.clear();
        r->blob.clear();
        r->flushed = true;
    }
    pend.clear();
    c->cv.notify_all();
}

void open_gate(Combiner* c) {
    if (!c->at_gate.empty()) {
        std::shared_ptr<Cohort> co = std::make_shared<Cohort>();
        co->members = (int)c->at_gate.size();
        for (Recorder* r : c->at_gate) r->cohort = co;
        c->at_gate.clear();
    }
    ++c->gate_generation;
    c->cv.notify_all();
}

// mu held.  Post the scan's recorded launches to its cohort, flush when the cohort is complete, wait for the device.
int post_and_wait(Recorder* r, std::unique_lock<std::mutex>& lk) {
    Combiner* c = r->comb;
    c->n_recorded += r->recs.size();
    if (r->recs.empty()) {
        const int err = r->error;
        if (err != 0) set_error("%s", r->error_text);
        r->error = 0;
        return err;
    }
    Cohort* co = r->cohort.get();
    r->flushed = false;
    co->pending.push_back(r);
    {
        SHG_HOST_TIME("combiner: waiting for the cohort, flush");
        if ((int)co->pending.size() >= co->members) flush_locked(c, co);
        while (!r->flushed)
            if (c->cv.wait_for(lk, std::chrono::microseconds(c->wait_us)) == std::cv_status::timeout && !r->flushed) flush_locked(c, co);
    }
    int err = r->error;
    if (err == 0) {
        lk.unlock();
        hipError_t he;
        {
            SHG_HOST_TIME("combiner: waiting for the device");
            he = hipEventSynchronize(r->ev);
        }
        lk.lock();
        if (he != hipSuccess) { set_error("launch combiner: %s", hipGetErrorString(he)); err = (int)he; }
    } else {
        set_error("%s", r->error_text);
    }
    r->error = 0;
    return err;
}

}  // namespace

int record_launch(Recorder* r, const LaunchRec& rec, const void* args) {
    if (r->direct) {                                        // plain launches went to the scan's own stream since the last wait
        const hipError_t he = hipStreamSynchronize(r->own);
        r->direct = false;
        if (he != hipSuccess) { set_error("%s: %s", rec.what, hipGetErrorString(he)); return (int)he; }
    }
    const size_t off = (r->blob.size() + 15) / 16 * 16;
    r->blob.resize(off + rec.arg_bytes);
    memcpy(r->blob.data() + off, args, rec.arg_bytes);
    r->recs.push_back(rec);
    r->recs.back().arg_off = off;
    return 0;
}

int stream_sync(hipStream_t st, const char* who) {
    Recorder* r = t_rec;
    if (!r) {
        const hipError_t he = hipStreamSynchronize(st);
        if (he != hipSuccess) { set_error("%s: %s", who, hipGetErrorString(he)); return (int)he; }
        return 0;
    }
    if (r->direct) {
        const hipError_t he = hipStreamSynchronize(r->own);
        r->direct = false;
        if (he != hipSuccess) { set_error("%s: %s", who, hipGetErrorString(he)); return (int)he; }
    }
    std::unique_lock<std::mutex> lk(r->comb->mu);
    return post_and_wait(r, lk);
}

// Around a pool thread's wait for its pass A on the lane.  What it has recorded (normally nothing) goes out first; when the pass has
// finished the scan stands at the gate.
void pool_wait_begin() {
    Recorder* r = t_rec;
    if (!r) return;
    std::unique_lock<std::mutex> lk(r->comb->mu);
    const int err = post_and_wait(r, lk);
    if (err != 0) fail(r, err);                              // (surfaces at the scan's next stream_sync)
    ++r->comb->on_lane;
}

void pool_wait_end() {
    Recorder* r = t_rec;
    if (!r) return;
    Combiner* c = r->comb;
    std::unique_lock<std::mutex> lk(c->mu);
    --c->on_lane;
    if (c->group <= 1) return;
    // leave the cohort this scan has been in so far (a cohort of its own since it began: nothing is pending there)
    if (r->cohort) --r->cohort->members;
    r->cohort.reset();
    SHG_HOST_TIME("gate (waiting for the cohort to form)");
    const unsigned long long mine = c->gate_generation;
    c->at_gate.push_back(r);
    // Enough of them, or nobody else in the pool is running a scan (a lone file: nothing to wait for): go.  Otherwise wait for the
    // scans whose pass is still on the lane (they arrive one pass apart); with none there, give stragglers gate_us -- scans that
    // became free together (a cohort that has just finished) arrive within microseconds of each other.
    if ((int)c->at_gate.size() >= c->group || (int)c->at_gate.size() >= c->busy) open_gate(c);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->gate_generation == mine) {
        c->cv.wait_for(lk, std::chrono::microseconds(c->gate_us));
        if (c->gate_generation != mine) break;
        const long waited = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if ((c->on_lane == 0 && waited >= c->gate_us) || waited >= c->gate_hard_us) open_gate(c);
    }
}

int direct_launches_follow(hipStream_t st) {
    Recorder* r = t_rec;
    if (!r) return 0;
    if (!r->recs.empty())
        if (int e = stream_sync(st, "launch combiner")) return e;
    r->direct = true;
    return 0;
}

// ---- what the pool calls ---------------------------------------------------------------------------------------------
Combiner* combiner_create() {
    Combiner* c = new Combiner;
    for (hipStream_t& s : c->streams)
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            for (hipStream_t t : c->streams)
                if (t) (void)hipStreamDestroy(t);
            delete c;
            return nullptr;
        }
    if (const char* v = getenv("SHG_COMBINE_WAIT_US")) {
        const long us = atol(v);
        if (us > 0) c->wait_us = us;
    }
    if (const char* v = getenv("SHG_COMBINE_GROUP")) {
        const int g = atoi(v);
        if (g >= 1 && g <= 64) c->group = g;
    }
    if (const char* v = getenv("SHG_COMBINE_GATE_US")) {
        const long us = atol(v);
        if (us > 0) c->gate_us = us;
    }
    return c;
}

void combiner_destroy(Combiner* c) {
    if (!c) return;
    for (hipStream_t s : c->streams)
        if (s) (void)hipStreamDestroy(s);
    delete c;
}

// A pool thread starts / ends a scan.  rec: the thread's recorder (its event is made on first use).
int combiner_enter(Combiner* c, Recorder* rec, hipStream_t own) {
    if (!rec->ev) {
        const hipError_t he = hipEventCreateWithFlags(&rec->ev, hipEventDisableTiming);
        if (he != hipSuccess) { set_error("launch combiner: %s", hipGetErrorString(he)); return (int)he; }
    }
    rec->comb = c;
    rec->own = own;
    rec->recs.clear();
    rec->blob.clear();
    rec->flushed = rec->direct = false;
    rec->error = 0;
    rec->cohort = std::make_shared<Cohort>();                // alone until the gate puts it with others
    rec->cohort->members = 1;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        ++c->busy;
    }
    t_rec = rec;
    return 0;
}

void combiner_leave(Combiner* c) {
    Recorder* r = t_rec;
    t_rec = nullptr;
    if (!r) return;
    std::unique_lock<std::mutex> lk(c->mu);
    --c->busy;
    if (!c->at_gate.empty() && (int)c->at_gate.size() >= c->busy) open_gate(c);       // (those at the gate were waiting for this scan alone)
    if (r->cohort) {
        Cohort* co = r->cohort.get();
        --co->members;
        if (co->members > 0 && (int)co->pending.size() >= co->members) flush_locked(c, co);     // the others were waiting for this scan alone
        r->cohort.reset();
    }
}

void combiner_stats(Combiner* c, unsigned long long* out3) {
    std::lock_guard<std::mutex> lk(c->mu);
    out3[0] = c->n_recorded;
    out3[1] = c->n_dispatches;
    out3[2] = c->n_flushes;
}

}  // namespace shg
