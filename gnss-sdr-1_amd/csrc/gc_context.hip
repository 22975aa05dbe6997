// gc_context.hip -- context, error reporting, version (libgnsscorr.so).
#include "gc_internal.h"
#include <cstring>

static thread_local char g_err[512] = "";

void gc_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

gc_status gc_fail(gc_status st, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return st;
}

void gc_ctx_retain(gc_ctx* ctx) { ctx->refs.fetch_add(1); }

void gc_ctx_release(gc_ctx* ctx)
{
    if (ctx->refs.fetch_sub(1) != 1) return;
    gc_device_guard g(ctx->device);
    if (void* b = ctx->l1_batcher.load())
        if (ctx->l1_batcher_free) ctx->l1_batcher_free(b);
    if (ctx->stream)
        {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamDestroy(ctx->stream);
        }
    delete ctx;
}

extern "C" {

const char* gc_last_error(void) { return g_err; }

const char* gc_version(void) { return "gnsscorr 0.2 (gfx950)"; }

gc_status gc_abi_check(size_t sizeof_epoch_params, size_t sizeof_loop_conf, size_t sizeof_loop_record, size_t sizeof_loop_sync_conf, size_t sizeof_acq_conf,
    size_t sizeof_acq_result)
{
    const struct
    {
        const char* name;
        size_t theirs, ours;
    } t[] = {{"gc_epoch_params", sizeof_epoch_params, sizeof(gc_epoch_params)}, {"gc_loop_conf", sizeof_loop_conf, sizeof(gc_loop_conf)},
        {"gc_loop_record", sizeof_loop_record, sizeof(gc_loop_record)}, {"gc_loop_sync_conf", sizeof_loop_sync_conf, sizeof(gc_loop_sync_conf)},
        {"gc_acq_conf", sizeof_acq_conf, sizeof(gc_acq_conf)}, {"gc_acq_result", sizeof_acq_result, sizeof(gc_acq_result)}};
    for (const auto& e : t)
        if (e.theirs != e.ours)
            return gc_fail(GC_ERR_INVALID, "gc_abi_check: %s is %zu bytes in the caller's binding, %zu in this library (%s): header and library differ", e.name,
                e.theirs, e.ours, gc_version());
    return GC_OK;
}

int gc_build_has_experiments(void)
{
#ifdef GNSSCORR_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

int gc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

gc_status gc_ctx_create(int device, gc_ctx** out)
{
    GC_REQUIRE(out != nullptr, "gc_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return gc_fail(GC_ERR_NO_DEVICE, "gc_ctx_create: no HIP device is visible; libgnsscorr has no CPU fallback");
    if (device < 0 || device >= n)
        return gc_fail(GC_ERR_NO_DEVICE, "gc_ctx_create: device %d out of range (%d visible)", device, n);
    gc_device_guard g(device);
    if (!g.ok) return gc_fail(GC_ERR_HIP, "gc_ctx_create: hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    GC_HIP(hipGetDeviceProperties(&prop, device));
    gc_ctx* c = new gc_ctx();
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    c->lds_max = prop.sharedMemPerBlock;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess)
        {
            delete c;
            return gc_fail(GC_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
    *out = c;
    return GC_OK;
}

gc_status gc_ctx_destroy(gc_ctx* ctx)
{
    if (!ctx) return GC_OK;
    gc_ctx_release(ctx);  // the context lives on while handles created on it exist
    return GC_OK;
}

gc_status gc_ctx_synchronize(gc_ctx* ctx)
{
    GC_REQUIRE(ctx != nullptr, "gc_ctx_synchronize: ctx is NULL");
    gc_device_guard g(ctx->device);
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GC_OK;
}

}  // extern "C"
