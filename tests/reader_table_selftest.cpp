// Threaded CPU test of gc_reader_table.h (the eviction bookkeeping of the RF stream ring) with host-side events:
// a producer "pushes" (advances head, overwrites ring cells after wait_evictable) while consumers reserve a floor,
// check residency, "launch" (a worker thread that reads the cells later, then signals the event) and commit.
// Every cell a launch reads must still hold the sample number it was validated for.  Built by tests/test_reader_table.py.
#include "gc_reader_table.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <random>
#include <thread>

struct HostEvent
{
    std::mutex m;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;  // a record hands out a sequence number; the "device" completes them in order
};
struct HostStream
{
    // work enqueued on the fake stream runs on the consumer's worker thread; an event record captures what must finish first
    std::atomic<uint64_t> enqueued{0}, finished{0};
};
struct HostPolicy
{
    typedef HostEvent* event_t;
    typedef HostStream* stream_t;
    struct Rec { HostStream* st; uint64_t upto; };
    static std::mutex& gm() { static std::mutex m; return m; }
    static std::vector<std::pair<HostEvent*, Rec>>& recs() { static std::vector<std::pair<HostEvent*, Rec>> v; return v; }
    static Rec find(HostEvent* e)
    {
        std::lock_guard<std::mutex> lk(gm());
        for (auto& p : recs())
            if (p.first == e) return p.second;
        return Rec{nullptr, 0};
    }
    static bool query(HostEvent* e)
    {
        Rec r = find(e);
        return !r.st || r.st->finished.load() >= r.upto;
    }
    static void synchronize(HostEvent* e)
    {
        while (!query(e)) std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    static std::atomic<int>& fail_records() { static std::atomic<int> n{0}; return n; }  // > 0: the next records fail (error path)
    static bool record(HostEvent* e, HostStream* st)
    {
        if (fail_records().load() > 0)
            {
                fail_records().fetch_sub(1);
                return false;
            }
        std::lock_guard<std::mutex> lk(gm());
        for (auto& p : recs())
            if (p.first == e)
                {
                    p.second = Rec{st, st->enqueued.load()};
                    return true;
                }
        recs().push_back({e, Rec{st, st->enqueued.load()}});
        return true;
    }
};

int main()
{
    const uint64_t capacity = 4096, total = 1500000;
    std::vector<std::atomic<uint64_t>> ring(capacity);
    for (auto& c : ring) c.store(~0ull);
    std::mutex mtx;
    gc_reader_table<HostPolicy> table;
    std::vector<std::unique_ptr<HostEvent>> evs;
    std::vector<HostEvent*> ev_ptrs;
    for (int i = 0; i < 4; i++)
        {
            evs.emplace_back(new HostEvent());
            ev_ptrs.push_back(evs.back().get());
        }
    table.init(ev_ptrs);
    uint64_t head = 0, evicting_below = 0;
    // ---- error paths of a launch: "reserve, then fail before the enqueue" must cancel, "enqueue, then the event
    // record fails" must commit -- either way the slot is released, so a push that evicts below its floor does not block and the
    // teardown drain returns.  A slot left pending would hang here (the test's timeout is the assertion).
    {
        std::unique_lock<std::mutex> lk(mtx);
        auto zero = []() { return (uint64_t)0; };
        HostStream st;
        int s0 = table.reserve(lk, 10, zero);
        table.cancel(s0);                    // the launch was never enqueued
        int s1 = table.reserve(lk, 20, zero);
        HostPolicy::fail_records().store(1);
        const bool ok = table.commit(s1, &st);  // enqueued, but its completion cannot be recorded: reported, slot released
        table.wait_evictable(lk, 1000);
        table.drain(lk);
        int live = 0;
        for (auto& r : table.readers()) live += r.active;
        if (ok || live != 0 || s0 < 0 || s1 < 0)
            {
                std::printf("reader table: error paths left %d slot(s) reserved (commit returned %d)\n", live, (int)ok);
                return 1;
            }
    }
    std::atomic<bool> stop{false};
    std::atomic<long> bad{0}, reads{0}, behind{0};

    auto oldest_of = [&]() {
        uint64_t o = head > capacity ? head - capacity : 0;
        return o > evicting_below ? o : evicting_below;
    };

    std::thread producer([&] {
        std::mt19937_64 rng(1);
        while (head < total)
            {
                const uint64_t n = 1 + rng() % 700;
                std::unique_lock<std::mutex> lk(mtx);
                const uint64_t new_head = head + n;
                const uint64_t new_oldest = new_head > capacity ? new_head - capacity : 0;
                if (new_oldest > evicting_below) evicting_below = new_oldest;
                table.wait_evictable(lk, new_oldest);
                for (uint64_t a = head; a < new_head; a++) ring[a % capacity].store(a);
                head = new_head;
                lk.unlock();
                if (rng() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 200));
            }
        stop.store(true);
    });

    HostStream streams[3];  // outlive the consumers: the final drain still queries the events recorded on them
    auto consumer = [&](int id) {
        std::mt19937_64 rng(100 + id);
        HostStream& st = streams[id];
        struct Job { uint64_t first, n; };
        std::mutex qm;
        std::vector<Job> q;
        std::atomic<bool> done{false};
        // the "device": executes enqueued jobs later, in order
        std::thread device([&] {
            std::mt19937_64 drng(200 + id);
            for (;;)
                {
                    Job j{0, 0};
                    bool have = false;
                    {
                        std::lock_guard<std::mutex> lk(qm);
                        if (st.finished.load() < q.size())
                            {
                                j = q[st.finished.load()];
                                have = true;
                            }
                    }
                    if (!have)
                        {
                            if (done.load()) return;
                            std::this_thread::sleep_for(std::chrono::microseconds(50));
                            continue;
                        }
                    std::this_thread::sleep_for(std::chrono::microseconds(drng() % 300));  // queued behind other work
                    for (uint64_t a = j.first; a < j.first + j.n; a++)
                        if (ring[a % capacity].load() != a) bad.fetch_add(1);
                    reads.fetch_add((long)j.n);
                    st.finished.fetch_add(1);
                }
        });
        while (!stop.load())
            {
                uint64_t first, n;
                int slot;
                {
                    std::unique_lock<std::mutex> lk(mtx);
                    const uint64_t o = oldest_of();
                    if (head - o < 64)
                        {
                            lk.unlock();
                            std::this_thread::yield();
                            continue;
                        }
                    // near the oldest resident sample, where a concurrent push bites first
                    first = o + rng() % 16;
                    n = 1 + rng() % std::min<uint64_t>(head - first, 512);
                    slot = table.reserve(lk, first, oldest_of);
                    if (slot < 0)
                        {
                            behind.fetch_add(1);
                            continue;
                        }
                    if (first + n > head)  // residency check against the range seen WITH the reservation
                        {
                            table.cancel(slot);
                            continue;
                        }
                }
                std::this_thread::sleep_for(std::chrono::microseconds(rng() % 100));  // the window the old protocol left open
                {
                    std::lock_guard<std::mutex> lk(qm);
                    q.push_back(Job{first, n});
                    st.enqueued.fetch_add(1);
                }
                {
                    std::lock_guard<std::mutex> lk(mtx);
                    table.commit(slot, &st);
                }
            }
        done.store(true);
        device.join();
    };
    std::thread c0(consumer, 0), c1(consumer, 1), c2(consumer, 2);
    producer.join();
    c0.join();
    c1.join();
    c2.join();
    {
        std::unique_lock<std::mutex> lk(mtx);
        table.drain(lk);
    }
    std::printf("reader table: %ld samples read by launches, %ld stale, %ld reservations refused, head %llu\n", reads.load(), bad.load(), behind.load(),
        (unsigned long long)head);
    return (bad.load() == 0 && reads.load() > 10000) ? 0 : 1;
}
