// CPU self-test of the level-1 epoch batcher (gnss-sdr-1_amd/csrc/gc_l1_batcher.h) with a host-side backend: the "GPU" is a
// list of operations per lane that run when the lane is waited on (copies first, then the batch's "kernel": a checksum of every
// request's window times its salt), with a 20-50 us sleep for the launch-to-completion latency.  What is checked:
//   * every call returns the value computed directly from the caller's window, whatever way the window reached the backend
//     (registered region read in place, union of overlapping windows copied once, the caller's own staged copy, leader-staged);
//   * calls are served in FEWER launches than calls when many threads call at once (the point of the batcher);
//   * register -> unregister -> register -> unregister of one buffer, then a call: the second unregister finds the
//     LIVE slot, emptied slots are reused, and no launch ever reads through the device view of memory that is no longer registered
//     -- also with a thread that registers / unregisters the buffer while 16 threads call into it;
//   * no data race: tests/test_l1_batcher.py also builds this file with -fsanitize=thread.
// Contract being protected: dll_pll_veml_tracking.cc:886-911 called from one scheduler thread per channel.
#include "gc_l1_batcher.h"

#include <atomic>
#include <cassert>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

static const uintptr_t VIEW = (uintptr_t)1 << 62;  // device view of registered host memory = host address + VIEW

struct FakeChan
{
    const void* iq;
    unsigned long long n_iq;  // bytes here
};
struct FakeParams
{
    uint32_t salt;
};

struct FakeBackend
{
    typedef FakeChan Chan;
    typedef FakeParams Params;
    struct Op
    {
        int kind;  // 0 copy, 1 launch
        const void* src;
        void* dst;
        size_t bytes;
        int B;
    };
    struct Lane
    {
        Chan* h_chans = nullptr;
        Params* h_params = nullptr;
        char* h_out = nullptr;
        char* d_span = nullptr;
        size_t span_cap = 0;
        char* h_span = nullptr;
        const char* dv_span = nullptr;
        size_t hspan_cap = 0;
        std::vector<Op> pending;
        int out_bytes = 0;
        std::mt19937 rng{12345};
    };
    struct Guard
    {
        explicit Guard(FakeBackend&) {}
    };
    // the "driver's" table of page-locked registrations
    std::mutex reg_m;
    std::vector<std::pair<const char*, size_t>> live;
    std::atomic<long> stale_reads{0}, launches{0}, copies{0}, unregisters{0};

    void host_register(const void* base, size_t bytes)
    {
        std::lock_guard<std::mutex> lk(reg_m);
        live.emplace_back(static_cast<const char*>(base), bytes);
    }
    void host_unregister(const void* base)
    {
        std::lock_guard<std::mutex> lk(reg_m);
        for (size_t i = 0; i < live.size(); i++)
            if (live[i].first == base)
                {
                    live.erase(live.begin() + (long)i);
                    unregisters++;
                    return;
                }
        std::fprintf(stderr, "host_unregister of memory that is not registered\n");
        std::abort();
    }
    // a "device" read address -> host address; a view of registered memory must be registered NOW
    const char* resolve(const void* p, size_t bytes)
    {
        uintptr_t a = (uintptr_t)p;
        if (a < VIEW) return static_cast<const char*>(p);
        const char* h = reinterpret_cast<const char*>(a - VIEW);
        std::lock_guard<std::mutex> lk(reg_m);
        for (auto& r : live)
            if (h >= r.first && h + bytes <= r.first + r.second) return h;
        stale_reads++;
        return h;
    }
    bool lane_init(Lane& l, int maxb, int max_out_bytes)
    {
        l.h_chans = new Chan[maxb];
        l.h_params = new Params[maxb];
        l.h_out = new char[(size_t)maxb * max_out_bytes];
        l.out_bytes = max_out_bytes;
        return true;
    }
    void lane_free(Lane& l)
    {
        delete[] l.h_chans;
        delete[] l.h_params;
        delete[] l.h_out;
        std::free(l.d_span);
        std::free(l.h_span);
        l = Lane();
    }
    bool span_reserve(Lane& l, size_t need)
    {
        if (need <= l.span_cap) return true;
        std::free(l.d_span);
        l.d_span = static_cast<char*>(std::malloc(need + 4096));
        l.span_cap = need + 4096;
        return true;
    }
    bool hspan_reserve(Lane& l, size_t need)
    {
        if (need <= l.hspan_cap) return true;
        std::free(l.h_span);
        l.h_span = static_cast<char*>(std::malloc(need + 4096));
        l.dv_span = l.h_span;
        l.hspan_cap = need + 4096;
        return true;
    }
    int copy(Lane& l, const void* src, void* dst, size_t bytes)
    {
        l.pending.push_back(Op{0, src, dst, bytes, 0});
        copies++;
        return 0;
    }
    template <class Rq>
    int launch(Lane& l, const Rq& k, int B, int)
    {
        (void)k;
        l.pending.push_back(Op{1, nullptr, nullptr, 0, B});
        launches++;
        return 0;
    }
    static uint64_t checksum(const char* p, size_t n, uint32_t salt)
    {
        uint64_t h = 1469598103934665603ull ^ salt;
        for (size_t i = 0; i < n; i++) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
        return h;
    }
    int wait(Lane& l)
    {
        std::this_thread::sleep_for(std::chrono::microseconds(20 + (int)(l.rng() % 31)));
        for (const Op& op : l.pending)
            {
                if (op.kind == 0)
                    std::memcpy(op.dst, resolve(op.src, op.bytes), op.bytes);
                else
                    for (int i = 0; i < op.B; i++)
                        {
                            const Chan& c = l.h_chans[i];
                            const uint64_t v = checksum(resolve(c.iq, (size_t)c.n_iq), (size_t)c.n_iq, l.h_params[i].salt);
                            std::memcpy(l.h_out + (size_t)i * 8, &v, 8);
                        }
            }
        l.pending.clear();
        return 0;
    }
    int oom_error() const { return 2; }
    const char* error_string(int) const { return "fake backend error"; }
};

typedef gc_l1_batcher_t<FakeBackend> Batcher;

static std::atomic<long> wrong{0};

// one synchronous call, like gc_correlator_carrier_wipeoff_multicorrelator_resampler: returns after the result is in *out
static void call(Batcher& b, const char* win, size_t bytes, uint32_t salt, int n_corr, char* own_pinned)
{
    Batcher::Request rq;
    rq.chan.iq = nullptr;
    rq.chan.n_iq = bytes;
    rq.params.salt = salt;
    rq.n_corr = n_corr;
    rq.host_sig = win;
    rq.sig_bytes = bytes;
    uint64_t out = 0;
    rq.out_host = &out;
    rq.out_bytes = 8;
    const int st = b.submit(&rq, own_pinned, own_pinned);
    const uint64_t want = FakeBackend::checksum(win, bytes, salt);
    if (st != 0 || out != want) wrong++;
}

static void do_register(Batcher& b, FakeBackend& be, const char* base, size_t bytes)
{
    be.host_register(base, bytes);
    b.add_region(base, bytes, reinterpret_cast<const void*>((uintptr_t)base + VIEW));
}

int main(int argc, char** argv)
{
    const int n_threads = argc > 1 ? std::atoi(argv[1]) : 64;
    const int n_calls = argc > 2 ? std::atoi(argv[2]) : 1000;
    FakeBackend be;
    Batcher b(&be, 64);
    if (!b.ok()) return 2;
    const size_t S = 1 << 20;
    std::vector<char> stream(S), regbuf(S);
    std::mt19937 g(7);
    for (auto& c : stream) c = (char)g();
    for (auto& c : regbuf) c = (char)g();

    // ---- 1. ADVICE round 2: register -> unregister -> register -> unregister of ONE array, followed by a call ----
    {
        std::vector<char> own(8192);
        do_register(b, be, regbuf.data(), S);
        call(b, regbuf.data() + 64, 4000, 1, 3, own.data());
        if (!b.remove_region(regbuf.data())) return 3;
        do_register(b, be, regbuf.data(), S);
        call(b, regbuf.data() + 128, 4000, 2, 3, own.data());
        if (!b.remove_region(regbuf.data())) return 4;   // must hit the LIVE second registration, not the dead first slot
        if (b.remove_region(regbuf.data())) return 5;    // nothing is registered any more
        call(b, regbuf.data() + 256, 4000, 3, 3, own.data());  // staged like any unregistered window
        if (be.unregisters != 2 || b.live_regions() != 0 || b.region_slots() != 1 || be.stale_reads != 0 || wrong != 0)
            {
                std::printf("FAIL sequence: unregisters %ld live %zu slots %zu stale %ld wrong %ld\n", be.unregisters.load(), b.live_regions(), b.region_slots(),
                    be.stale_reads.load(), wrong.load());
                return 6;
            }
    }

    // ---- 2. many threads, random overlaps, a buffer registered and unregistered while calls into it are in flight ----
    std::atomic<bool> stop{false};
    std::thread control([&] {
        std::mt19937 r(99);
        while (!stop.load())
            {
                do_register(b, be, regbuf.data(), S);
                std::this_thread::sleep_for(std::chrono::microseconds(200 + (int)(r() % 800)));
                if (!b.remove_region(regbuf.data())) std::abort();
                std::this_thread::sleep_for(std::chrono::microseconds(50 + (int)(r() % 300)));
            }
    });
    const long launches0 = be.launches;
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&, t] {
            std::mt19937 r(1000 + t);
            std::vector<char> own(16384), priv(65536);
            for (auto& c : priv) c = (char)r();
            for (int k = 0; k < n_calls; k++)
                {
                    const size_t bytes = 8 * (size_t)(13 + r() % 1500);  // 104 .. 12 KB, whole samples
                    const uint32_t salt = (uint32_t)r();
                    const int n_corr = (r() % 8 == 0) ? 5 : 3;  // two kernel shapes: never mixed in one batch
                    const char* win;
                    switch (r() % 4)
                        {
                        case 0:  // channels at neighbouring read positions of one stream buffer (8-byte granularity: odd 16-byte phases too)
                            win = stream.data() + 8 * ((size_t)(k * 97) % 1000 + r() % 600);
                            break;
                        case 1:  // everybody hands in the same pointer (the reference's own timing test)
                            win = stream.data() + 4096;
                            break;
                        case 2:  // a buffer per thread: nothing overlaps
                            win = priv.data() + 8 * (r() % 1024);
                            break;
                        default:  // inside the buffer the control thread keeps registering and unregistering
                            win = regbuf.data() + 8 * ((size_t)(k * 31) % 2000 + r() % 600);
                            break;
                        }
                    call(b, win, bytes, salt, n_corr, own.data());
                    if (bytes == 0) std::abort();
                }
        });
    for (auto& x : th) x.join();
    stop = true;
    control.join();
    const Batcher::Stats st = b.stats();
    const long launches = be.launches - launches0;
    const long total = (long)n_threads * n_calls;
    std::printf("%ld calls from %d threads: %ld launches, largest batch %d, %llu windows shared, %ld copies, %ld register/unregister cycles, %ld stale, %ld wrong\n",
        total, n_threads, launches, st.max_batch, st.n_shared, be.copies.load(), be.unregisters.load() - 2, be.stale_reads.load(), wrong.load());
    if (wrong != 0 || be.stale_reads != 0) return 7;
    if ((long)st.n_requests != total + 3) return 8;
    if (n_threads >= 16 && !(launches < total)) return 9;  // served N calls in fewer than N launches
    if (b.live_regions() != 0 || b.region_slots() > 2) return 10;
    std::printf("OK\n");
    return 0;
}
