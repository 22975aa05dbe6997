// gc_stream.h -- HBM ring of one RF stream (internal view; C ABI in gnsscorr.h).
#ifndef GC_STREAM_H
#define GC_STREAM_H
#include "gc_internal.h"
#include <vector>

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
    // kernels in flight that read the ring: the oldest absolute index each may touch + its completion
    struct Reader
    {
        uint64_t min_index;
        hipEvent_t done;
        bool active;
    };
    std::vector<Reader> readers;
    std::mutex mtx;
    std::atomic<int> refs{1};  // the creator's reference + one per batch channel that reads the ring
};

void gc_stream_keep(gc_stream* s);
void gc_stream_drop(gc_stream* s);

// oldest absolute index still resident
static inline uint64_t gc_stream_oldest(const gc_stream* s) { return s->head > s->capacity ? s->head - s->capacity : 0; }
// Makes `compute` wait for the newest push (call before enqueueing a kernel that reads the ring).
gc_status gc_stream_begin_read(gc_stream* s, hipStream_t compute);
// Registers the kernel(s) just enqueued on `compute`: they read absolute indices >= min_index.
gc_status gc_stream_end_read(gc_stream* s, hipStream_t compute, uint64_t min_index);

#endif
