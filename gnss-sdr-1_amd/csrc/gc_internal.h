// gc_internal.h -- shared internals of libgnsscorr.so (context, error handling).
#ifndef GC_INTERNAL_H
#define GC_INTERNAL_H

#include "gnsscorr.h"
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <atomic>
#include <cstdio>
#include <mutex>

// Experiment switches (measured-slower kernels and their tuning variables) exist only in builds made with
// `make EXTRA=-DGNSSCORR_EXPERIMENTS`: the product library reads no tuning variable from the environment and carries none of those
// kernels.  gc_build_has_experiments() (gnsscorr.h) tells a test which build it runs on.
#include <cstdlib>
#ifdef GNSSCORR_EXPERIMENTS
static inline const char* gc_exp_env(const char* name) { return std::getenv(name); }
#else
static inline const char* gc_exp_env(const char*) { return nullptr; }
#endif

// thread-local last-error text (gc_last_error)
void gc_set_error(const char* fmt, ...);
gc_status gc_fail(gc_status st, const char* fmt, ...);

#define GC_HIP(call)                                                                          \
    do                                                                                        \
        {                                                                                     \
            hipError_t e_ = (call);                                                           \
            if (e_ != hipSuccess)                                                             \
                return gc_fail(GC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                      \
        }                                                                                     \
    while (0)

#define GC_REQUIRE(cond, ...)                              \
    do                                                     \
        {                                                  \
            if (!(cond)) return gc_fail(GC_ERR_INVALID, __VA_ARGS__); \
        }                                                  \
    while (0)

struct gc_ctx
{
    int device = -1;
    hipStream_t stream = nullptr;
    int n_cus = 0;
    size_t lds_max = 0;
    std::mutex mtx;  // serialises entry points that share this context's scratch
    // one reference for the creator (dropped by gc_ctx_destroy) + one per live handle created on the context:
    // the stream and the struct go away with the last of them, so handles may be destroyed in any order
    std::atomic<int> refs{1};
    // level-1 epoch batcher (gc_tracking.hip), created on first use, released with the context
    std::atomic<void*> l1_batcher{nullptr};
    void (*l1_batcher_free)(void*) = nullptr;
};

void gc_ctx_retain(gc_ctx* ctx);
void gc_ctx_release(gc_ctx* ctx);

// member of every handle struct: keeps the context alive for the handle's lifetime
struct gc_ctx_ref
{
    gc_ctx* p = nullptr;
    void bind(gc_ctx* c)
    {
        p = c;
        gc_ctx_retain(c);
    }
    ~gc_ctx_ref()
    {
        if (p) gc_ctx_release(p);
    }
};

// RAII device selection for entry points (contexts may live on different GPUs)
struct gc_device_guard
{
    int prev = -1;
    bool ok = true;
    explicit gc_device_guard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~gc_device_guard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

static inline hipStream_t gc_pick_stream(gc_ctx* ctx, void* stream)
{
    return stream ? reinterpret_cast<hipStream_t>(stream) : ctx->stream;
}

#endif
