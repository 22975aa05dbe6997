// gc_l1_batcher.h -- the level-1 epoch batcher (SURVEY.md section 7 "Latency vs batching", section 8b "one instance per channel
// thread"): queue, lanes, per-thread waiters, window sharing and the host-buffer registry, WITHOUT any HIP type.
//
// The reference runs one Cpu_Multicorrelator_Real_Codes per channel on that channel's scheduler thread and every thread calls
// Carrier_wipeoff_multicorrelator_resampler once per code period (dll_pll_veml_tracking.cc:886-911, gnss_flowgraph.cc:496-499).
// Here those synchronous calls meet in a per-context queue: a caller that finds a free lane becomes the leader of everything
// queued at that moment with the same kernel shape (taps, mode, sample format, slices), builds ONE launch for the batch (one
// channel descriptor + parameter record per request, results scattered to each caller's corr_out) and wakes the others;
// requests that arrive while a batch is on the GPU form the next one, so the batch size follows the load by itself (group
// commit: no timer, a lone caller is served at once).  Two lanes (own stream and staging each) keep a batch in preparation
// while another executes.  Callers whose windows overlap (channels reading neighbouring positions of one GNU Radio buffer)
// share one copy of the union into HBM instead of one PCIe read each.
//
// Header-only and templated on a Backend (launch + wait + copy + page-locking), the way gc_reader_table.h is templated on its
// event policy: gc_tracking.hip instantiates it with HIP streams and the tracking kernel; tests/l1_batcher_selftest.cpp with a
// host-side backend whose "GPU" is a sleep + a checksum, under ThreadSanitizer (64 threads x 1000 calls, overlapping windows,
// buffers registered and unregistered while calls are in flight).
//
// Backend concept:
//   typedef Chan;    channel descriptor copied into the lane's array; has a member `const void* iq`
//   typedef Params;  per-call parameter record
//   struct Lane { Chan* h_chans; Params* h_params; char* h_out;            page-locked arrays of MAXB entries
//                 char* d_span; size_t span_cap;                           device buffer for shared windows
//                 char* h_span; const char* dv_span; size_t hspan_cap;     page-locked staging for unions + its device view
//                 ... };
//   struct Guard { explicit Guard(Backend&); };                            RAII "this thread talks to the batcher's device"
//   bool lane_init(Lane&, int maxb, int max_out_bytes); void lane_free(Lane&);
//   bool span_reserve(Lane&, size_t); bool hspan_reserve(Lane&, size_t);
//   int  copy(Lane&, const void* src_dev_view, void* dst, size_t bytes);   enqueue on the lane's stream; 0 = ok
//   int  launch(Lane&, const Request& key, int B, int lds_floats);         enqueue the batch's kernel; 0 = ok
//   int  wait(Lane&);                                                      block until the lane's stream has drained; 0 = ok
//   int  oom_error() const; const char* error_string(int) const;
//   void host_unregister(const void* base);
#ifndef GC_L1_BATCHER_H
#define GC_L1_BATCHER_H
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

// What a calling thread sleeps on.  Its own mutex, so that waking a batch's callers does not queue them up on the batcher's
// mutex; held through shared_ptr by the request and by whoever is about to signal it (the signal is sent after the batcher's
// mutex is released, when the request may already be gone).
struct gc_l1_waiter
{
    std::mutex m;
    std::condition_variable cv;
    bool signaled = false;
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return signaled; });
        signaled = false;
    }
    void signal()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            signaled = true;
        }
        cv.notify_one();
    }
};

template <class Backend>
class gc_l1_batcher_t
{
public:
    typedef typename Backend::Chan Chan;
    typedef typename Backend::Params Params;
    typedef typename Backend::Lane Lane;
    static constexpr int MAXB = 256, LANES = 2;
    enum
    {
        ST_OK = 0,
        ST_BACKEND = 1
    };

    struct Request
    {
        Chan chan;
        Params params;
        int n_corr = 0, mode = 0, fmt = 0, n_slices = 1, lds_floats = 0;
        const char* host_sig = nullptr;  // caller's sig_in
        size_t sig_bytes = 0;
        // where the window is read from: (a) a registered host region (region >= 0: no staging copy, device view reg_dev); (b) this
        // request's page-locked copy (self_copied: made by the calling thread before it queued, in parallel with the others);
        // (c) deferred: the request overlapped an earlier queued one when it arrived and copied nothing -- the leader of its batch
        // stages the UNION of overlapping windows once (channels read neighbouring positions of one stream buffer)
        int region = -1;
        const char* reg_dev = nullptr;
        const void* own_dev = nullptr;  // device view of this request's page-locked buffer
        bool self_copied = false;
        bool copy_done = false;  // queued and ready to be taken
        void* out_host = nullptr;
        size_t out_bytes = 0;
        int status = ST_OK;
        char err[200];
        bool taken = false;
        std::atomic<bool> done{false};        // set last by the batch's leader; the owner returns on it without taking any lock
        std::shared_ptr<gc_l1_waiter> waiter;  // the owner thread's
        Request() { err[0] = 0; }
    };

    // One gc_ctx_register_host_buffer().  A slot is never erased (queued calls hold indices into the list); an unregistered
    // slot is EMPTY: bytes == 0 and host == dev == nullptr, and is matched by nothing.
    struct Region
    {
        const char* host = nullptr;
        size_t bytes = 0;
        const char* dev = nullptr;
        int users = 0;         // queued or running requests that read through `dev`
        bool closing = false;  // an unregister is waiting for users to reach 0: no new request may pick the slot
    };

    struct Stats
    {
        unsigned long long n_batches = 0, n_requests = 0, n_shared = 0;
        int max_batch = 0;
        double t_prep = 0, t_launch = 0, t_sync = 0, t_scatter = 0, t_queue = 0;  // microseconds, summed
    };

    explicit gc_l1_batcher_t(Backend* be, int max_out_bytes) : be_(be)
    {
        bool ok = true;
        for (auto& l : lanes_)
            {
                ok = ok && be_->lane_init(l, MAXB, max_out_bytes);
                // the span buffers of a typical batch up front (page-locking megabytes takes milliseconds: not inside a call)
                if (ok) (void)be_->span_reserve(l, (size_t)8 << 20);
                if (ok) (void)be_->hspan_reserve(l, (size_t)8 << 20);
            }
        ok_ = ok;
    }
    ~gc_l1_batcher_t()
    {
        for (auto& l : lanes_) be_->lane_free(l);
        for (auto& r : regions_)
            if (r.host) be_->host_unregister(r.host);  // live slots only: an emptied slot's memory belongs to its owner again
    }
    bool ok() const { return ok_; }
    Backend* backend() { return be_; }
    int min_second_lane = 16;  // ready calls needed to start a batch while another one is running

    static bool same_shape(const Request* a, const Request* b)
    {
        return a->n_corr == b->n_corr && a->mode == b->mode && a->fmt == b->fmt && a->n_slices == b->n_slices;
    }
    // a request the next leader can take: its window is in place (own copy finished, or a registered region)
    static bool ready(const Request* r) { return !r->taken && r->copy_done; }

    // The synchronous call: queues rq, leads a batch or sleeps until a leader has served it.  own_pinned / own_pinned_dev: the
    // calling correlator's page-locked window buffer and its device view.  Returns rq->status.
    int submit(Request* rq, void* own_pinned, const void* own_pinned_dev)
    {
        const double t_in = now_us();
        struct Acc
        {
            std::atomic<unsigned long long>* t;
            double t0;
            ~Acc() { t->fetch_add((unsigned long long)((now_us() - t0) * 1e3), std::memory_order_relaxed); }  // runs with the lock released
        } acc{&t_queue_ns_, t_in};
        static thread_local std::shared_ptr<gc_l1_waiter> my_waiter = std::make_shared<gc_l1_waiter>();
        rq->waiter = my_waiter;
        std::unique_lock<std::mutex> lk(m_);
        // registered region?  (live slots only; a slot that is being unregistered takes no new readers)
        for (size_t i = 0; i < regions_.size(); i++)
            {
                Region& g = regions_[i];
                if (g.bytes != 0 && !g.closing && rq->host_sig >= g.host && rq->host_sig + rq->sig_bytes <= g.host + g.bytes)
                    {
                        rq->region = (int)i;
                        rq->reg_dev = g.dev + (rq->host_sig - g.host);
                        rq->chan.iq = rq->reg_dev;
                        rq->copy_done = true;
                        g.users++;
                        break;
                    }
            }
        // the staged copy keeps the 16-byte phase of the caller's pointer, like the shared and the registered paths do: the kernel's
        // pair alignment, and with it the order of its sums, is then the same however the window reaches the GPU
        const size_t own_lead = (uintptr_t)rq->host_sig & 15;
        rq->own_dev = static_cast<const char*>(own_pinned_dev) + own_lead;
        if (rq->region < 0 && rq->sig_bytes > 0)
            {
                // an earlier, still queued call of the same kernel shape whose window overlaps this one: copy nothing, the leader of
                // the batch stages the union once
                for (Request* r : queue_)
                    if (!r->taken && r->region < 0 && r->sig_bytes > 0 && same_shape(r, rq) && rq->host_sig < r->host_sig + r->sig_bytes &&
                        r->host_sig < rq->host_sig + rq->sig_bytes)
                        {
                            rq->copy_done = true;
                            break;
                        }
            }
        else
            rq->copy_done = true;
        queue_.push_back(rq);
        bool need_copy = !rq->copy_done;
        for (;;)
            {
                if (need_copy)
                    {
                        lk.unlock();
                        if (rq->sig_bytes > 0) std::memcpy(static_cast<char*>(own_pinned) + own_lead, rq->host_sig, rq->sig_bytes);
                        lk.lock();
                        rq->self_copied = true;
                        rq->copy_done = true;
                        need_copy = false;
                    }
                if (rq->done.load(std::memory_order_acquire)) break;
                Lane* lane = nullptr;
                int lane_i = -1;
                if (!rq->taken)
                    for (int i = 0; i < LANES; i++)
                        if (!busy_[i])
                            {
                                lane = &lanes_[i];
                                lane_i = i;
                                break;
                            }
                const Request* key = nullptr;
                if (lane)
                    {
                        int n_ready = 0;
                        bool other_busy = false;
                        for (int i = 0; i < LANES; i++) other_busy |= busy_[i];
                        for (Request* r : queue_)
                            if (ready(r))
                                {
                                    if (!key) key = r;
                                    n_ready++;
                                }
                        // A batch costs about the same whether it carries one call or fifty (launch + completion latency), so a
                        // second lane is opened only for a batch worth it; a handful of calls wait for the running batch to finish
                        // and are joined by everything that arrives meanwhile.
                        if (other_busy && n_ready < min_second_lane) key = nullptr;
                    }
                if (!lane || !key)
                    {
                        // sleep on this thread's own waiter: whoever completes the request, frees a lane for it or detaches it
                        // from its representative signals it
                        lk.unlock();
                        rq->waiter->wait();
                        if (rq->done.load(std::memory_order_acquire)) return rq->status;
                        lk.lock();
                        continue;
                    }
                // lead: everything queued right now with the shape of the oldest ready request
                std::vector<Request*> batch;
                for (auto it = queue_.begin(); it != queue_.end() && (int)batch.size() < MAXB;)
                    {
                        Request* r = *it;
                        if (!r->taken && r->copy_done && same_shape(r, key))
                            {
                                r->taken = true;
                                batch.push_back(r);
                                it = queue_.erase(it);
                            }
                        else
                            ++it;
                    }
                // the regions as they are now: a slot a member of this batch reads through cannot be emptied before the batch is
                // done (users > 0 holds the unregister back), so the copy stays valid for the batch's lifetime
                const std::vector<Region> regions = regions_;
                busy_[lane_i] = true;
                lk.unlock();
                int n_shared = 0;
                run_batch(*lane, batch, regions, &n_shared);
                lk.lock();
                busy_[lane_i] = false;
                stats_.n_batches++;
                stats_.n_requests += batch.size();
                stats_.n_shared += (unsigned long long)n_shared;
                stats_.max_batch = std::max(stats_.max_batch, (int)batch.size());
                // the batch no longer reads its registered regions: an unregister waiting for them may go ahead
                bool region_freed = false;
                for (Request* r : batch)
                    if (r->region >= 0)
                        {
                            Region& g = regions_[(size_t)r->region];
                            if (--g.users == 0 && g.closing) region_freed = true;
                        }
                // who to wake, collected under the lock and signalled after it is released: the batch's callers, the owner of the
                // oldest ready request (it leads next), and followers whose representative left without them
                std::vector<std::shared_ptr<gc_l1_waiter>> wake;
                bool handed = false;
                for (Request* r : queue_)
                    if (r != rq && !handed && ready(r))
                        {
                            wake.push_back(r->waiter);
                            handed = true;
                        }
                for (Request* r : batch)
                    {
                        if (r != rq) wake.push_back(r->waiter);
                        r->done.store(true, std::memory_order_release);  // r may be gone from here on (its owner returns on the flag)
                    }
                lk.unlock();
                if (region_freed) region_cv_.notify_all();
                for (auto& w : wake) w->signal();
                lk.lock();
            }
        lk.unlock();
        return rq->status;
    }

    // gc_ctx_register_host_buffer: `dev` is the device view of the (already page-locked) memory
    void add_region(const void* base, size_t bytes, const void* dev)
    {
        std::lock_guard<std::mutex> lk(m_);
        Region g;
        g.host = static_cast<const char*>(base);
        g.bytes = bytes;
        g.dev = static_cast<const char*>(dev);
        for (auto& r : regions_)
            if (r.host == nullptr && r.bytes == 0 && !r.closing)  // reuse an emptied slot: nothing refers to it any more
                {
                    r = g;
                    return;
                }
        regions_.push_back(g);
    }

    // gc_ctx_unregister_host_buffer: matches LIVE slots only; takes the slot out of service, waits until no queued or running
    // request reads through its device view, unpins, and leaves the slot empty.  false: `base` is not registered.
    bool remove_region(const void* base)
    {
        std::unique_lock<std::mutex> lk(m_);
        for (size_t i = 0; i < regions_.size(); i++)
            if (regions_[i].bytes != 0 && !regions_[i].closing && regions_[i].host == base)
                {
                    regions_[i].closing = true;  // no new readers; calls inside the range stage their windows from now on
                    region_cv_.wait(lk, [&] { return regions_[i].users == 0; });
                    lk.unlock();
                    be_->host_unregister(base);
                    lk.lock();
                    regions_[i] = Region();  // empty: host == dev == nullptr, bytes == 0
                    return true;
                }
        return false;
    }

    Stats stats()
    {
        std::lock_guard<std::mutex> lk(m_);
        Stats s = stats_;
        s.t_queue = (double)t_queue_ns_.load(std::memory_order_relaxed) * 1e-3;
        return s;
    }
    size_t live_regions()
    {
        std::lock_guard<std::mutex> lk(m_);
        size_t n = 0;
        for (auto& r : regions_) n += (r.bytes != 0);
        return n;
    }
    size_t region_slots()
    {
        std::lock_guard<std::mutex> lk(m_);
        return regions_.size();
    }

private:
    static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // runs one batch on `lane` (no lock held); fills status / err of every request
    void run_batch(Lane& lane, std::vector<Request*>& batch, const std::vector<Region>& regions, int* n_shared_out)
    {
        typename Backend::Guard g(*be_);
        const double ta = now_us();
        const int B = (int)batch.size();
        const Request& k = *batch[0];
        int lds_floats = 0;
        for (Request* r : batch) lds_floats = std::max(lds_floats, r->lds_floats);
        int e = 0;
        int n_shared = 0;
        size_t span_off = 0;
        // room for every shared window of this batch, reserved before any pointer into the buffers is handed out
        bool can_share = false, can_stage = false;
        {
            size_t need = 0, need_host = 0;
            for (Request* r : batch)
                if (r->sig_bytes > 0)
                    {
                        need += r->sig_bytes + 512;
                        if (r->region < 0) need_host += r->sig_bytes + 512;
                    }
            can_share = be_->span_reserve(lane, need);
            can_stage = need_host == 0 || be_->hspan_reserve(lane, need_host);
        }
        // (1) windows inside a registered host region: the callers' windows of one region overlap when the channels read
        // neighbouring positions of one stream buffer.  When the union of the windows is clearly smaller than their sum, the union
        // crosses PCIe ONCE (one DMA straight from the caller's page-locked memory into HBM) and every request reads its piece of
        // it; otherwise each request reads its window from the registered memory in place.
        for (size_t ri = 0; ri < regions.size() && e == 0 && can_share; ri++)
            {
                const char* lo = nullptr;
                const char* hi = nullptr;
                size_t sum = 0;
                int members = 0;
                for (Request* r : batch)
                    if (r->region == (int)ri && r->sig_bytes > 0)
                        {
                            lo = (!lo || r->host_sig < lo) ? r->host_sig : lo;
                            hi = (!hi || r->host_sig + r->sig_bytes > hi) ? r->host_sig + r->sig_bytes : hi;
                            sum += r->sig_bytes;
                            members++;
                        }
                if (members < 2) continue;
                // the copy starts on a 16-byte boundary of the caller's memory, so every window keeps its alignment
                const char* lo_al = lo - ((uintptr_t)lo & 15);
                if (lo_al < regions[ri].host) continue;  // an unaligned region start: read in place
                const size_t uni = (size_t)(hi - lo_al);
                const size_t off = (span_off + 255) & ~(size_t)255;
                if (uni * 3 > sum * 2 || off + uni > lane.span_cap) continue;  // nothing to gain: read in place
                e = be_->copy(lane, regions[ri].dev + (lo_al - regions[ri].host), lane.d_span + off, uni);
                for (Request* r : batch)
                    if (r->region == (int)ri && r->sig_bytes > 0)
                        {
                            r->chan.iq = lane.d_span + off + (r->host_sig - lo_al);
                            n_shared++;
                        }
                span_off = off + uni;
            }
        // (2) unregistered input.  A window staged by its own calling thread is read in place from that thread's page-locked buffer.
        // Windows that overlap (identical pointers; channels at neighbouring read positions of one GNU Radio buffer) form a
        // cluster: the leader copies the cluster's UNION once from the callers' memory -- every byte of it lies inside the window
        // of some member, and every member is blocked in its synchronous call, so the memory is valid and stable -- into the lane's
        // page-locked span, one copy kernel moves it into HBM and each member reads its piece there.
        size_t hoff = 0;
        {
            std::vector<Request*> un;
            for (Request* r : batch)
                if (r->region < 0 && r->sig_bytes > 0) un.push_back(r);
            std::sort(un.begin(), un.end(), [](const Request* a, const Request* c) { return a->host_sig < c->host_sig; });
            for (size_t i = 0; i < un.size() && e == 0;)
                {
                    size_t j = i + 1;
                    const char* lo = un[i]->host_sig;
                    const char* hi = lo + un[i]->sig_bytes;
                    size_t sum = un[i]->sig_bytes;
                    bool any_deferred = !un[i]->self_copied;
                    while (j < un.size() && un[j]->host_sig <= hi)
                        {
                            hi = std::max(hi, un[j]->host_sig + un[j]->sig_bytes);
                            sum += un[j]->sig_bytes;
                            any_deferred |= !un[j]->self_copied;
                            j++;
                        }
                    const size_t uni = (size_t)(hi - lo), lead = (uintptr_t)lo & 15;
                    const bool cluster = (j - i >= 2) && (any_deferred || uni * 3 <= sum * 2);
                    if ((cluster || any_deferred) && !can_stage)
                        {
                            e = be_->oom_error();
                            break;
                        }
                    if (cluster)
                        {
                            const size_t off = (hoff + 255) & ~(size_t)255;
                            std::memcpy(lane.h_span + off + lead, lo, uni);
                            hoff = off + lead + uni;
                            const size_t doff = (span_off + 255) & ~(size_t)255;
                            if (can_share && doff + lead + uni <= lane.span_cap)
                                {
                                    e = be_->copy(lane, lane.dv_span + off, lane.d_span + doff, lead + uni);
                                    for (size_t t = i; t < j; t++) un[t]->chan.iq = lane.d_span + doff + lead + (un[t]->host_sig - lo);
                                    span_off = doff + lead + uni;
                                }
                            else
                                for (size_t t = i; t < j; t++) un[t]->chan.iq = lane.dv_span + off + lead + (un[t]->host_sig - lo);  // read in place over PCIe
                            n_shared += (int)(j - i);
                        }
                    else
                        for (size_t t = i; t < j; t++)
                            {
                                Request* r = un[t];
                                if (r->self_copied)
                                    r->chan.iq = r->own_dev;
                                else
                                    {
                                        // deferred, but what it overlapped went into another batch: the leader stages this window
                                        const size_t off = (hoff + 255) & ~(size_t)255, ld = (uintptr_t)r->host_sig & 15;
                                        std::memcpy(lane.h_span + off + ld, r->host_sig, r->sig_bytes);
                                        hoff = off + ld + r->sig_bytes;
                                        r->chan.iq = lane.dv_span + off + ld;
                                    }
                            }
                    i = j;
                }
        }
        for (int i = 0; i < B; i++)
            {
                lane.h_chans[i] = batch[i]->chan;
                lane.h_params[i] = batch[i]->params;
            }
        const double tb = now_us();
        if (e == 0) e = be_->launch(lane, k, B, lds_floats);
        const double tc = now_us();
        if (e == 0) e = be_->wait(lane);
        const double td = now_us();
        for (int i = 0; i < B; i++)
            {
                Request* r = batch[i];
                if (e != 0)
                    {
                        r->status = ST_BACKEND;
                        std::snprintf(r->err, sizeof r->err, "tracking kernel (batch of %d) failed: %s", B, be_->error_string(e));
                    }
                else
                    std::memcpy(r->out_host, lane.h_out + (size_t)i * r->out_bytes, r->out_bytes);
            }
        *n_shared_out = n_shared;
        const double te = now_us();
        std::lock_guard<std::mutex> lk(m_);
        stats_.t_prep += tb - ta;
        stats_.t_launch += tc - tb;
        stats_.t_sync += td - tc;
        stats_.t_scatter += te - td;
    }

    Backend* be_;
    std::mutex m_;
    std::condition_variable region_cv_;
    std::deque<Request*> queue_;
    Lane lanes_[LANES];
    bool busy_[LANES] = {false, false};
    std::vector<Region> regions_;
    bool ok_ = false;
    Stats stats_;
    std::atomic<unsigned long long> t_queue_ns_{0};
};

#endif
