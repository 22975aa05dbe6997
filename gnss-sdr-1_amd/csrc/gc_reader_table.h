// gc_reader_table.h -- bookkeeping of the launches that read an RF stream ring (gc_stream).
//
// A launch that reads the ring is protected against eviction from the moment its slot is RESERVED, which
// happens before the residency check and before the kernel is enqueued; the slot is COMMITTED (its completion
// event recorded) right after the enqueue.  A push that would evict samples at or above a reserved floor waits
// -- for the commit if it has not happened yet, then for the event.  Header-only and templated on the event
// policy so that the protocol is exercised on the CPU with host-side events (tests/reader_table_selftest.cpp);
// gc_stream.hip instantiates it with HIP events.
//
// The caller serialises every method with the stream's mutex and passes the held unique_lock to the two
// methods that may block.
#ifndef GC_READER_TABLE_H
#define GC_READER_TABLE_H
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <vector>

template <class EventPolicy>
class gc_reader_table
{
public:
    typedef typename EventPolicy::event_t event_t;
    typedef typename EventPolicy::stream_t stream_t;
    static constexpr uint64_t FLOOR_OLDEST = ~0ull;  // reserve(): "whatever is oldest right now"

    struct Reader
    {
        uint64_t min_index = 0;  // oldest absolute sample index the launch may touch
        event_t done{};          // completion of the launch (valid once committed)
        bool active = false;     // reserved or committed and not yet known to have finished
        bool pending = false;    // reserved, launch not enqueued yet (no event to wait on)
    };

    // events are created by the owner (n slots) and handed over
    void init(const std::vector<event_t>& events)
    {
        readers_.assign(events.size(), Reader());
        for (size_t i = 0; i < events.size(); i++) readers_[i].done = events[i];
    }
    const std::vector<Reader>& readers() const { return readers_; }

    // Reserves a slot with floor `min_index`.  `oldest()` is the oldest resident index, evaluated under the lock (again after
    // every wait: the ring may move on while this call waits for a slot).  Returns the slot (>= 0), or -1 when the floor is
    // older than the oldest resident index (nothing reserved); *floor_out receives the floor that was reserved.
    template <class OldestFn>
    int reserve(std::unique_lock<std::mutex>& lk, uint64_t min_index, OldestFn oldest, uint64_t* floor_out = nullptr)
    {
        const bool follow_oldest = (min_index == FLOOR_OLDEST);
        for (;;)
            {
                int free_slot = -1, oldest_committed = -1;
                for (size_t i = 0; i < readers_.size(); i++)
                    {
                        Reader& r = readers_[i];
                        if (r.active && !r.pending && EventPolicy::query(r.done)) r.active = false;
                        if (!r.active && free_slot < 0) free_slot = (int)i;
                        if (r.active && !r.pending && (oldest_committed < 0 || r.min_index < readers_[oldest_committed].min_index)) oldest_committed = (int)i;
                    }
                const uint64_t o = oldest();
                if (follow_oldest) min_index = o;
                if (min_index < o) return -1;
                if (free_slot >= 0)
                    {
                        Reader& r = readers_[free_slot];
                        if (floor_out) *floor_out = min_index;
                        r.min_index = min_index;
                        r.active = true;
                        r.pending = true;
                        return free_slot;
                    }
                if (oldest_committed >= 0)
                    {
                        // every slot is taken: wait (on the host) for the committed launch with the oldest floor
                        event_t ev = readers_[oldest_committed].done;
                        lk.unlock();
                        EventPolicy::synchronize(ev);
                        lk.lock();
                    }
                else
                    cv_.wait(lk);  // all of them reserved by other threads and not yet committed
            }
    }

    // The launch of `slot` has been enqueued on `st`: record its completion.
    bool commit(int slot, stream_t st)
    {
        Reader& r = readers_[slot];
        const bool ok = EventPolicy::record(r.done, st);
        r.pending = false;
        if (!ok) r.active = false;  // nothing to wait on: the caller reports the failure
        cv_.notify_all();
        return ok;
    }

    // The launch of `slot` was not enqueued after all.
    void cancel(int slot)
    {
        readers_[slot].active = false;
        readers_[slot].pending = false;
        cv_.notify_all();
    }

    // Blocks until no launch may still read samples below `new_oldest` (called by a push before it overwrites them).
    void wait_evictable(std::unique_lock<std::mutex>& lk, uint64_t new_oldest)
    {
        for (;;)
            {
                Reader* blocker = nullptr;
                for (auto& r : readers_)
                    {
                        if (r.active && !r.pending && EventPolicy::query(r.done)) r.active = false;
                        if (r.active && r.min_index < new_oldest && (!blocker || (blocker->pending && !r.pending))) blocker = &r;
                    }
                if (!blocker) return;
                if (blocker->pending)
                    {
                        cv_.wait(lk);  // reserved, not enqueued yet: its commit (or cancel) wakes us
                        continue;
                    }
                event_t ev = blocker->done;
                lk.unlock();
                EventPolicy::synchronize(ev);
                lk.lock();
            }
    }

    // every committed launch has finished (teardown)
    void drain(std::unique_lock<std::mutex>& lk)
    {
        wait_evictable(lk, ~0ull);
    }

private:
    std::vector<Reader> readers_;
    std::condition_variable cv_;
};

#endif
