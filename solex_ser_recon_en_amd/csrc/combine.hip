// The launch combiner of a scan pool (launch.h says why): scans in flight record their kernel launches; when every one of them is
// waiting -- for its results, or for its pass A on the lane -- whatever has been recorded goes to the device, the same kernel of
// different scans as ONE dispatch.
//
// The rule "flush when every busy thread waits" needs no tuning and costs a lone scan nothing (its own wait is the moment); under
// load it puts the scans of a pool in step with each other: W scans advance one segment (the launches between two host decisions)
// per round, and a segment of W scans is launched as often as a segment of one.  A thread that has waited SHG_COMBINE_WAIT_US
// (default 60) for the others flushes what there is: a long host computation of one scan (the line fit: 150 us) does not hold the
// others' kernels back for longer than that.
#include <stdlib.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include "shg_common.h"

namespace shg {

thread_local Recorder* t_rec = nullptr;

struct Combiner {
    std::mutex mu;
    std::condition_variable cv;
    int busy = 0, waiting = 0;
    std::vector<Recorder*> pending;
    hipStream_t streams[2] = {nullptr, nullptr};
    int turn = 0;
    long wait_us = 60;
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

// mu held.  Everything pending goes to one of the combiner's streams, merged position by position.
void flush_locked(Combiner* c) {
    if (c->pending.empty()) return;
    ++c->n_flushes;
    hipStream_t st = c->streams[c->turn];
    c->turn ^= 1;
    const size_t np = c->pending.size();
    std::vector<size_t> cursor(np, 0);
    const LaunchRec* group[kMaxMerge];
    const unsigned char* gargs[kMaxMerge];
    size_t members[kMaxMerge];
    for (;;) {
        size_t first = np;
        for (size_t p = 0; p < np; ++p)
            if (c->pending[p]->error == 0 && cursor[p] < c->pending[p]->recs.size()) { first = p; break; }
        if (first == np) break;
        const LaunchRec& head = c->pending[first]->recs[cursor[first]];
        int n = 0;
        uint64_t blocks = 0;
        for (size_t p = first; p < np && n < head.cap && n < kMaxMerge; ++p) {
            Recorder* r = c->pending[p];
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
            Recorder* r = c->pending[members[i]];
            if (e != 0) fail(r, e);
            if (++cursor[members[i]] == r->recs.size() && r->error == 0) {
                const hipError_t he = hipEventRecord(r->ev, st);
                if (he != hipSuccess) { set_error("launch combiner: %s", hipGetErrorString(he)); fail(r, (int)he); }
            }
        }
    }
    for (Recorder* r : c->pending) {
        r->recs.clear();
        r->blob.clear();
        r->flushed = true;
    }
    c->pending.clear();
    c->cv.notify_all();
}

// The calling pool thread stops launching until leave_wait(): its recorded launches (if any) are posted.
void enter_wait(Recorder* r, std::unique_lock<std::mutex>& lk) {
    Combiner* c = r->comb;
    r->posted = !r->recs.empty();
    if (r->posted) {
        r->flushed = false;
        c->pending.push_back(r);
    }
    ++c->waiting;
    if (c->waiting >= c->busy) flush_locked(c);
}

int leave_wait(Recorder* r, std::unique_lock<std::mutex>& lk) {
    Combiner* c = r->comb;
    if (r->posted) {
        while (!r->flushed) {
            if (c->cv.wait_for(lk, std::chrono::microseconds(c->wait_us)) == std::cv_status::timeout && !r->flushed) flush_locked(c);
        }
    }
    int err = r->error;
    if (r->posted && err == 0) {
        lk.unlock();
        const hipError_t he = hipEventSynchronize(r->ev);        // (the others' flushes do not need this thread)
        lk.lock();
        if (he != hipSuccess) { set_error("launch combiner: %s", hipGetErrorString(he)); err = (int)he; }
    } else if (err != 0) {
        set_error("%s", r->error_text);
    }
    r->posted = false;
    r->error = 0;
    --c->waiting;
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
    r->comb->n_recorded += r->recs.size();
    enter_wait(r, lk);
    return leave_wait(r, lk);
}

void pool_wait_begin() {
    Recorder* r = t_rec;
    if (!r) return;
    std::unique_lock<std::mutex> lk(r->comb->mu);
    r->comb->n_recorded += r->recs.size();
    enter_wait(r, lk);
}

void pool_wait_end() {
    Recorder* r = t_rec;
    if (!r) return;
    std::unique_lock<std::mutex> lk(r->comb->mu);
    (void)leave_wait(r, lk);                                 // (an error of a recorded launch surfaces at the scan's next stream_sync: kept below)
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
    rec->posted = rec->flushed = rec->direct = false;
    rec->error = 0;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        ++c->busy;
    }
    t_rec = rec;
    return 0;
}

void combiner_leave(Combiner* c) {
    t_rec = nullptr;
    std::unique_lock<std::mutex> lk(c->mu);
    --c->busy;
    if (c->waiting >= c->busy) flush_locked(c);             // the others may have been waiting for this thread alone
}

void combiner_stats(Combiner* c, unsigned long long* out3) {
    std::lock_guard<std::mutex> lk(c->mu);
    out3[0] = c->n_recorded;
    out3[1] = c->n_dispatches;
    out3[2] = c->n_flushes;
}

}  // namespace shg
