// gc_stream.hip -- gc_stream_*: the IQ block of one RF stream, pushed once per GPU and shared by every
// channel / acquisition that reads that stream (all channels of a GNSS-SDR flowgraph read one stream,
// src/core/receiver/gnss_flowgraph.cc:496-499; SURVEY.md section 8b/8e "shared IQ ring per RF stream").
//
// Layout: a ring of `capacity` samples in HBM followed by a mirror of its first `max_window` samples, so
// that any window of <= max_window samples is contiguous wherever it starts: sample a lives at a % capacity
// and, when that is < max_window, also at capacity + a % capacity.  Kernels index with absolute sample
// numbers (TrkChan::ring_len).  Pushes run on the stream's own HIP stream through pinned staging slots and
// overlap with compute; a push waits only for kernels that may still read the samples it evicts.
#include "gc_stream.h"
#include <algorithm>
#include <cstring>

static void stream_release(gc_stream* s)
{
    if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
    (void)hipFree(s->d_ring);
    for (int i = 0; i < gc_stream::kSlots; i++)
        {
            if (s->h_slot[i]) (void)hipHostFree(s->h_slot[i]);
            if (s->slot_done[i]) (void)hipEventDestroy(s->slot_done[i]);
        }
    for (hipEvent_t e : s->reader_events)
        if (e) (void)hipEventDestroy(e);
    if (s->pushed) (void)hipEventDestroy(s->pushed);
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
}

void gc_stream_keep(gc_stream* s) { s->refs.fetch_add(1); }

void gc_stream_drop(gc_stream* s)
{
    if (s->refs.fetch_sub(1) != 1) return;
    gc_device_guard g(s->ctx->device);
    {
        std::unique_lock<std::mutex> lk(s->mtx);
        s->readers.drain(lk);
    }
    stream_release(s);
    delete s;
}

gc_status gc_stream_begin_read(gc_stream* s, hipStream_t compute, uint64_t min_index, gc_stream_ticket* t)
{
    std::unique_lock<std::mutex> lk(s->mtx);
    const int slot = s->readers.reserve(lk, min_index, [s]() { return gc_stream_oldest(s); });
    if (slot < 0)
        return gc_fail(GC_ERR_STATE, "the reader fell behind the stream ring: it needs sample %llu, the oldest resident sample is %llu",
            (unsigned long long)min_index, (unsigned long long)gc_stream_oldest(s));
    // the range as of the reservation (reserve() returns with the lock held): nothing at or above the floor is evicted from here on
    t->slot = slot;
    t->oldest = gc_stream_oldest(s);
    t->head = s->head;
    if (s->has_pushed)
        {
            hipError_t e = hipStreamWaitEvent(compute, s->pushed, 0);
            if (e != hipSuccess)
                {
                    s->readers.cancel(slot);
                    t->slot = -1;
                    return gc_fail(GC_ERR_HIP, "hipStreamWaitEvent failed: %s", hipGetErrorString(e));
                }
        }
    return GC_OK;
}

gc_status gc_stream_end_read(gc_stream* s, hipStream_t compute, const gc_stream_ticket& t)
{
    if (t.slot < 0) return GC_OK;
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->readers.commit(t.slot, compute)) return gc_fail(GC_ERR_HIP, "hipEventRecord failed behind a launch that reads the stream ring");
    return GC_OK;
}

void gc_stream_cancel_read(gc_stream* s, const gc_stream_ticket& t)
{
    if (t.slot < 0) return;
    std::lock_guard<std::mutex> lk(s->mtx);
    s->readers.cancel(t.slot);
}

extern "C" {

gc_status gc_stream_create(gc_ctx* ctx, int iq_format, uint64_t capacity_samples, uint32_t max_window_samples, gc_stream** out)
{
    GC_REQUIRE(ctx && out, "gc_stream_create: NULL argument");
    *out = nullptr;
    GC_REQUIRE(iq_format == GC_IQ_F32 || iq_format == GC_IQ_I16 || iq_format == GC_IQ_I8, "gc_stream_create: unknown format %d", iq_format);
    GC_REQUIRE(max_window_samples > 0 && capacity_samples >= 2ull * max_window_samples,
        "gc_stream_create: capacity must be at least twice the longest window");
    GC_REQUIRE(capacity_samples <= 0x7fffffffull, "gc_stream_create: capacity must be below 2^31 samples");
    gc_device_guard g(ctx->device);
    gc_stream* s = new gc_stream();
    s->ctx = ctx;
    s->ctx_ref.bind(ctx);
    s->iq_format = iq_format;
    s->elem = iq_format == GC_IQ_F32 ? 8 : iq_format == GC_IQ_I16 ? 4 : 2;
    s->capacity = capacity_samples;
    s->mirror = max_window_samples;
    s->slot_bytes = (size_t)4 << 20;
    hipError_t e = hipMalloc(&s->d_ring, (size_t)(s->capacity + s->mirror + 2) * s->elem);
    if (e == hipSuccess) e = hipMemset(s->d_ring, 0, (size_t)(s->capacity + s->mirror + 2) * s->elem);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->pushed, hipEventDisableTiming);
    for (int i = 0; i < gc_stream::kSlots && e == hipSuccess; i++)
        {
            e = hipHostMalloc(reinterpret_cast<void**>(&s->h_slot[i]), s->slot_bytes, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->slot_done[i], hipEventDisableTiming);
        }
    s->reader_events.assign(8, nullptr);
    for (auto& ev : s->reader_events)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) s->readers.init(s->reader_events);
    if (e != hipSuccess)
        {
            stream_release(s);
            delete s;
            return gc_fail(GC_ERR_HIP, "gc_stream_create: %s", hipGetErrorString(e));
        }
    *out = s;
    return GC_OK;
}

gc_status gc_stream_destroy(gc_stream* s)
{
    if (!s) return GC_OK;
    gc_stream_drop(s);  // batches that read the ring keep it alive until they are destroyed or re-pointed
    return GC_OK;
}

static gc_status stream_push(gc_stream* s, const void* host_iq, uint64_t n_samples, uint64_t* first_index, bool pinned);

gc_status gc_stream_push(gc_stream* s, const void* host_iq, uint64_t n_samples, uint64_t* first_index)
{
    return stream_push(s, host_iq, n_samples, first_index, false);
}

gc_status gc_stream_push_pinned(gc_stream* s, const void* pinned_host_iq, uint64_t n_samples, uint64_t* first_index)
{
    return stream_push(s, pinned_host_iq, n_samples, first_index, true);
}

gc_status gc_stream_broadcast_pinned(gc_stream* const* rings, int n_rings, const void* pinned_host_iq, uint64_t n_samples)
{
    GC_REQUIRE(rings && n_rings > 0 && pinned_host_iq, "gc_stream_broadcast_pinned: bad argument");
    // every ring has its own copy stream (and, on another GPU, its own DMA engines and link): the copies are enqueued back to
    // back and run concurrently; nothing is exchanged between the GPUs
    for (int i = 0; i < n_rings; i++)
        {
            GC_REQUIRE(rings[i], "gc_stream_broadcast_pinned: ring %d is NULL", i);
            GC_REQUIRE(rings[i]->iq_format == rings[0]->iq_format, "gc_stream_broadcast_pinned: ring %d has another sample format", i);
            gc_status st = stream_push(rings[i], pinned_host_iq, n_samples, nullptr, true);
            if (st != GC_OK) return st;
        }
    return GC_OK;
}

}  // extern "C"

static gc_status stream_push(gc_stream* s, const void* host_iq, uint64_t n_samples, uint64_t* first_index, bool pinned)
{
    GC_REQUIRE(s && host_iq, "gc_stream_push: NULL argument");
    GC_REQUIRE(n_samples <= s->capacity, "gc_stream_push: at most capacity = %llu samples per push", (unsigned long long)s->capacity);
    gc_device_guard g(s->ctx->device);
    std::lock_guard<std::mutex> one_push(s->push_mtx);
    std::unique_lock<std::mutex> lk(s->mtx);
    if (first_index) *first_index = s->head;
    if (n_samples == 0) return GC_OK;
    const uint64_t new_head = s->head + n_samples;
    const uint64_t new_oldest = new_head > s->capacity ? new_head - s->capacity : 0;
    // Launches that may still read samples this push evicts must finish first -- including launches whose reader slot is
    // reserved but which have not been enqueued yet (gc_reader_table.h).  From here on new readers see the post-push range.
    // The wait is done by the calling (producer) thread, not by the copy stream: a DMA queue that has to wait for a compute
    // signal takes the runtime's slow path (measured 316 us instead of 80 us per 3.2 MB push), and blocking here is also the
    // back-pressure that keeps the producer at most one ring ahead of the consumers.
    if (new_oldest > s->evicting_below) s->evicting_below = new_oldest;
    s->readers.wait_evictable(lk, new_oldest);
    const char* src = static_cast<const char*>(host_iq);
    uint64_t idx = s->head, left = n_samples;
    while (left > 0)
        {
            const uint64_t pos = idx % s->capacity;
            uint64_t len = std::min<uint64_t>(left, s->capacity - pos);
            int k = -1;
            if (pinned)
                {
                    // page-locked caller memory: DMA straight from it (the caller keeps it until gc_stream_synchronize)
                    GC_HIP(hipMemcpyAsync(s->d_ring + pos * s->elem, src, (size_t)len * s->elem, hipMemcpyHostToDevice, s->copy_stream));
                }
            else
                {
                    len = std::min<uint64_t>(len, s->slot_bytes / s->elem);
                    k = s->next_slot;
                    s->next_slot = (k + 1) % gc_stream::kSlots;
                    if (s->slot_busy[k]) GC_HIP(hipEventSynchronize(s->slot_done[k]));
                    std::memcpy(s->h_slot[k], src, (size_t)len * s->elem);
                    GC_HIP(hipMemcpyAsync(s->d_ring + pos * s->elem, s->h_slot[k], (size_t)len * s->elem, hipMemcpyHostToDevice, s->copy_stream));
                }
            if (pos < s->mirror)
                {
                    // the part that lands in the first max_window samples is repeated behind the ring (HBM to HBM)
                    const uint64_t mlen = std::min<uint64_t>(len, s->mirror - pos);
                    GC_HIP(hipMemcpyAsync(s->d_ring + (s->capacity + pos) * s->elem, s->d_ring + pos * s->elem, (size_t)mlen * s->elem,
                        hipMemcpyDeviceToDevice, s->copy_stream));
                }
            if (k >= 0)
                {
                    GC_HIP(hipEventRecord(s->slot_done[k], s->copy_stream));
                    s->slot_busy[k] = true;
                }
            src += (size_t)len * s->elem;
            idx += len;
            left -= len;
        }
    GC_HIP(hipEventRecord(s->pushed, s->copy_stream));
    s->has_pushed = true;
    s->head = new_head;
    return GC_OK;
}

extern "C" {

gc_status gc_stream_info(gc_stream* s, uint64_t* oldest_index, uint64_t* head_index, uint64_t* capacity_samples)
{
    GC_REQUIRE(s, "gc_stream_info: NULL handle");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (oldest_index) *oldest_index = gc_stream_oldest(s);
    if (head_index) *head_index = s->head;
    if (capacity_samples) *capacity_samples = s->capacity;
    return GC_OK;
}

gc_status gc_stream_synchronize(gc_stream* s)
{
    GC_REQUIRE(s, "gc_stream_synchronize: NULL handle");
    gc_device_guard g(s->ctx->device);
    GC_HIP(hipStreamSynchronize(s->copy_stream));
    return GC_OK;
}

}  // extern "C"
