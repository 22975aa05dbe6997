// gc_stream.h -- HBM ring of one RF stream (internal view; C ABI in gnsscorr.h).
#ifndef GC_STREAM_H
#define GC_STREAM_H
#include "gc_internal.h"
#include "gc_reader_table.h"
#include <vector>

struct gc_hip_event_policy
{
    typedef hipEvent_t event_t;
    typedef hipStream_t stream_t;
    static bool query(hipEvent_t e) { return hipEventQuery(e) == hipSuccess; }
    static void synchronize(hipEvent_t e) { (void)hipEventSynchronize(e); }
    static bool record(hipEvent_t e, hipStream_t st) { return hipEventRecord(e, st) == hipSuccess; }
};

struct gc_stream
{
    gc_ctx* ctx = nullptr;
    gc_ctx_ref ctx_ref;
    int iq_format = GC_IQ_F32;
    size_t elem = 8;         // bytes per complex sample
    uint64_t capacity = 0;   // samples in the ring
    uint64_t mirror = 0;     // samples repeated behind the ring so that a window never has to wrap
    char* d_ring = nullptr;  // (capacity + mirror) * elem bytes
    uint64_t head = 0;       // absolute index one past the newest sample
    hipStream_t copy_stream = nullptr;
    hipEvent_t pushed = nullptr;  // completion of the newest push
    bool has_pushed = false;
    // pinned staging for pageable caller buffers
    static const int kSlots = 4;
    size_t slot_bytes = 0;
    char* h_slot[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t slot_done[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool slot_busy[kSlots] = {false, false, false, false};
    int next_slot = 0;
    // launches that read the ring: reserved before their residency check, committed behind their enqueue (gc_reader_table.h)
    gc_reader_table<gc_hip_event_policy> readers;
    std::vector<hipEvent_t> reader_events;
    uint64_t evicting_below = 0;  // a push in progress is about to overwrite everything below this index
    std::mutex mtx;
    std::mutex push_mtx;  // one push at a time (a push may release mtx while it waits for readers)
    std::atomic<int> refs{1};  // the creator's reference + one per batch channel that reads the ring
};

void gc_stream_keep(gc_stream* s);
void gc_stream_drop(gc_stream* s);

// oldest absolute index still resident (a push in progress counts as done)
static inline uint64_t gc_stream_oldest(const gc_stream* s)
{
    const uint64_t o = s->head > s->capacity ? s->head - s->capacity : 0;
    return o > s->evicting_below ? o : s->evicting_below;
}

// A reserved read of the ring: slot of the reader table + the resident range [oldest, head) at the moment of the reservation.
struct gc_stream_ticket
{
    int slot = -1;
    uint64_t oldest = 0, head = 0;
};
static const uint64_t GC_STREAM_FLOOR_OLDEST = ~0ull;
// Call BEFORE checking residency and enqueueing a kernel that reads the ring: reserves a reader slot with floor `min_index`
// (GC_STREAM_FLOOR_OLDEST: the oldest resident sample), so that no push evicts samples at or above the floor until the launch
// registered by gc_stream_end_read has finished; makes `compute` wait for the newest push; reports the resident range, which
// stays valid for [floor, head) until end_read / cancel_read.  GC_ERR_STATE (nothing reserved) if the floor is no longer resident.
gc_status gc_stream_begin_read(gc_stream* s, hipStream_t compute, uint64_t min_index, gc_stream_ticket* t);
// The kernel(s) of the ticket have been enqueued on `compute`.
gc_status gc_stream_end_read(gc_stream* s, hipStream_t compute, const gc_stream_ticket& t);
// Nothing was enqueued after all (validation or launch failure).
void gc_stream_cancel_read(gc_stream* s, const gc_stream_ticket& t);

#endif
