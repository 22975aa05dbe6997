// gc_internal.h -- shared internals of libgnsscorr.so (context, error handling).
#ifndef GC_INTERNAL_H
#define GC_INTERNAL_H

#include "gnsscorr.h"
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <mutex>

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
