// gc_tracking.hip -- host side of the tracking path of libgnsscorr.so:
//   * gc_trk_batch_*   : batched, HBM-resident multi-channel / multi-epoch engine
//   * gc_correlator_*  : one object per channel with the method set of the
//                        reference's Cpu_Multicorrelator_Real_Codes
//                        (src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.h:45-69)
//                        and, with complex chips, of Cpu_Multicorrelator
//                        (src/algorithms/tracking/libs/cpu_multicorrelator.h:46-64)
#include "gc_internal.h"
#include "gc_l1_batcher.h"
#include "gc_stream.h"
#include "trk_kernels.h"
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>

// -----------------------------------------------------------------------------
// parameter marshalling: same float arithmetic as the reference's call site
// -----------------------------------------------------------------------------
extern "C" void gc_epoch_params_fill(gc_epoch_params* p, uint64_t sample_offset,
    float rem_carrier_phase_in_rad, float phase_step_rad, float phase_rate_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    int signal_length_samples)
{
    // cpu_multicorrelator_real_codes.cc:141  lv_cmake(std::cos(rem), -std::sin(rem))
    p->sample_offset = sample_offset;
    p->phase0_re = std::cos(rem_carrier_phase_in_rad);
    p->phase0_im = -std::sin(rem_carrier_phase_in_rad);
    // :145/:149  std::exp(lv_32fc_t(0.0, -phase_step_rad)), std::exp(lv_32fc_t(0.0, -phase_rate_step_rad))
    const std::complex<float> inc = std::exp(std::complex<float>(0.0f, -phase_step_rad));
    const std::complex<float> rate = std::exp(std::complex<float>(0.0f, -phase_rate_step_rad));
    p->phase_inc_re = inc.real();
    p->phase_inc_im = inc.imag();
    p->phase_rate_re = rate.real();
    p->phase_rate_im = rate.imag();
    p->rem_code_phase_chips = rem_code_phase_chips;
    p->code_phase_step_chips = code_phase_step_chips;
    p->code_phase_rate_step_chips = code_phase_rate_step_chips;
    p->n_samples = signal_length_samples;
}

// -----------------------------------------------------------------------------
// batched engine
// -----------------------------------------------------------------------------
struct gc_trk_batch
{
    gc_ctx* ctx = nullptr;
    gc_ctx_ref ctx_ref;
    int n_channels = 0, n_taps = 0, max_code_len = 0, mode = TRK_MODE_PLAIN;
    int lds_table_floats = 0;
    bool lds_sizing = true;  // size a launch's LDS by what a slice touches (gc_trk_batch_set_slices(b, -1) turns it off: A/B timing)
    int nominal_len = 0;
    int iq_format = GC_IQ_F32;
    int forced_slices = 0;
    bool complex_codes = false;
    bool sc16 = false;  // Cpu_Multicorrelator_16sc arithmetic: int16 IQ, int16 complex chips, int16 results
    std::vector<gc_stream*> streams;  // per channel: the ring it reads (gc_trk_batch_set_input_stream) or NULL
    bool has_read_floor = false;
    uint64_t read_floor = 0;  // oldest absolute index run_dev launches may read (gc_trk_batch_set_read_floor)
    std::vector<TrkChan> h_chans;
    bool chans_dirty = true;
    TrkChan* d_chans = nullptr;
    float* d_codes = nullptr;
    float2* d_partial = nullptr;
    size_t partial_cap = 0;
    gc_epoch_params* d_params = nullptr;
    size_t params_cap = 0;
    float2* d_out = nullptr;
    size_t out_cap = 0;
};

static const int kMaxLdsTableFloats = 16000;  // 64 KB default dynamic-LDS limit minus the header

static int pick_slices(const gc_trk_batch* b, int n_epochs, int max_len)
{
    if (b->forced_slices > 0) return b->forced_slices;
    // aim at >= 8 workgroups per CU; never cut an epoch below 4 chunks (2048 samples) per slice
    const long long jobs = (long long)b->n_channels * n_epochs;
    const long long want = 8LL * (b->ctx->n_cus > 0 ? b->ctx->n_cus : 256);
    if (jobs >= want || max_len <= 0) return 1;
    const int chunks = (max_len + 511) / 512;
    int s = (int)((want + jobs - 1) / jobs);
    int smax = chunks / 4;
    if (smax < 1) smax = 1;
    if (s > smax) s = smax;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

static gc_status batch_sync_chans(gc_trk_batch* b, hipStream_t st)
{
    if (!b->chans_dirty) return GC_OK;
    GC_HIP(hipMemcpyAsync(b->d_chans, b->h_chans.data(), sizeof(TrkChan) * b->n_channels, hipMemcpyHostToDevice, st));
    // the host vector may be edited again right after this call returns
    GC_HIP(hipStreamSynchronize(st));
    b->chans_dirty = false;
    return GC_OK;
}

extern "C" {

gc_status gc_trk_batch_create(gc_ctx* ctx, int n_channels, int n_taps, int max_code_length, int high_dyn,
    gc_trk_batch** out)
{
    GC_REQUIRE(ctx && out, "gc_trk_batch_create: NULL argument");
    *out = nullptr;
    GC_REQUIRE(n_channels > 0, "gc_trk_batch_create: n_channels must be > 0");
    GC_REQUIRE(n_taps >= 1 && n_taps <= GC_MAX_TAPS, "gc_trk_batch_create: n_taps must be in 1..%d", GC_MAX_TAPS);
    GC_REQUIRE(max_code_length > 0 && max_code_length + 64 <= kMaxLdsTableFloats,
        "gc_trk_batch_create: max_code_length must be in 1..%d", kMaxLdsTableFloats - 64);
    gc_device_guard g(ctx->device);
    gc_trk_batch* b = new gc_trk_batch();
    b->ctx = ctx;
    b->ctx_ref.bind(ctx);
    b->n_channels = n_channels;
    b->n_taps = n_taps;
    b->max_code_len = max_code_length;
    b->mode = high_dyn ? TRK_MODE_HD_FULL : TRK_MODE_PLAIN;
    b->lds_table_floats = max_code_length + 64;
    b->h_chans.assign(n_channels, TrkChan{});
    b->streams.assign(n_channels, nullptr);
    hipError_t e1 = hipMalloc(&b->d_chans, sizeof(TrkChan) * n_channels);
    hipError_t e2 = hipMalloc(&b->d_codes, sizeof(float) * (size_t)n_channels * max_code_length);
    if (e1 != hipSuccess || e2 != hipSuccess)
        {
            (void)hipFree(b->d_chans);
            (void)hipFree(b->d_codes);
            delete b;
            return gc_fail(GC_ERR_HIP, "gc_trk_batch_create: hipMalloc failed");
        }
    (void)hipMemset(b->d_codes, 0, sizeof(float) * (size_t)n_channels * max_code_length);
    for (int i = 0; i < n_channels; i++)
        {
            b->h_chans[i].code = b->d_codes + (size_t)i * max_code_length;
            b->h_chans[i].code_len = 0;
        }
    *out = b;
    return GC_OK;
}

gc_status gc_trk_batch_destroy(gc_trk_batch* b)
{
    if (!b) return GC_OK;
    gc_device_guard g(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    (void)hipFree(b->d_chans);
    (void)hipFree(b->d_codes);
    (void)hipFree(b->d_partial);
    (void)hipFree(b->d_params);
    (void)hipFree(b->d_out);
    for (gc_stream* r : b->streams)
        if (r) gc_stream_drop(r);
    delete b;
    return GC_OK;
}

gc_status gc_trk_batch_set_shifts(gc_trk_batch* b, int ch, const float* shifts_chips)
{
    GC_REQUIRE(b && shifts_chips, "gc_trk_batch_set_shifts: NULL argument");
    GC_REQUIRE(ch >= 0 && ch < b->n_channels, "gc_trk_batch_set_shifts: channel %d out of range", ch);
    for (int t = 0; t < b->n_taps; t++) b->h_chans[ch].shifts[t] = shifts_chips[t];
    b->chans_dirty = true;
    return GC_OK;
}

gc_status gc_trk_batch_set_code(gc_trk_batch* b, int ch, const float* code, int code_length, const float* shifts_chips)
{
    GC_REQUIRE(b && code && shifts_chips, "gc_trk_batch_set_code: NULL argument");
    GC_REQUIRE(ch >= 0 && ch < b->n_channels, "gc_trk_batch_set_code: channel %d out of range", ch);
    GC_REQUIRE(code_length > 0 && code_length <= b->max_code_len, "gc_trk_batch_set_code: code_length %d not in 1..%d",
        code_length, b->max_code_len);
    if (b->complex_codes) return gc_fail(GC_ERR_STATE, "gc_trk_batch_set_code: the batch holds complex codes (gc_trk_batch_set_code_complex)");
    if (b->sc16) return gc_fail(GC_ERR_STATE, "gc_trk_batch_set_code: the batch is in 16-bit mode (gc_trk_batch_set_code_16sc)");
    gc_device_guard g(b->ctx->device);
    std::lock_guard<std::mutex> lk(b->ctx->mtx);
    GC_HIP(hipStreamSynchronize(b->ctx->stream));  // launches on the context's stream may still read the old table
    GC_HIP(hipMemcpy(b->d_codes + (size_t)ch * b->max_code_len, code, sizeof(float) * code_length, hipMemcpyHostToDevice));
    b->h_chans[ch].code_len = code_length;
    return gc_trk_batch_set_shifts(b, ch, shifts_chips);
}

gc_status gc_trk_batch_set_complex_codes(gc_trk_batch* b, int on)
{
    GC_REQUIRE(b, "gc_trk_batch_set_complex_codes: NULL argument");
    const bool want = on != 0;
    if (want == b->complex_codes) return GC_OK;
    if (want && b->sc16) return gc_fail(GC_ERR_STATE, "gc_trk_batch_set_complex_codes: the batch is in 16-bit mode");
    if (want && b->mode != TRK_MODE_PLAIN)
        return gc_fail(GC_ERR_STATE, "gc_trk_batch_set_complex_codes: the complex-code correlator has no high-dynamics variant");
    const int per_chip = want ? 2 : 1;
    GC_REQUIRE(per_chip * (b->max_code_len + 64) <= kMaxLdsTableFloats,
        "gc_trk_batch_set_complex_codes: max_code_length %d too long for complex chips (max %d)", b->max_code_len, kMaxLdsTableFloats / 2 - 64);
    gc_device_guard g(b->ctx->device);
    std::lock_guard<std::mutex> lk(b->ctx->mtx);
    GC_HIP(hipStreamSynchronize(b->ctx->stream));
    float* d_new = nullptr;
    const size_t stride = (size_t)per_chip * b->max_code_len;
    GC_HIP(hipMalloc(&d_new, sizeof(float) * (size_t)b->n_channels * stride));
    (void)hipMemset(d_new, 0, sizeof(float) * (size_t)b->n_channels * stride);
    (void)hipFree(b->d_codes);
    b->d_codes = d_new;
    for (int i = 0; i < b->n_channels; i++)
        {
            b->h_chans[i].code = b->d_codes + (size_t)i * stride;
            b->h_chans[i].code_len = 0;  // every channel needs its code again
        }
    b->complex_codes = want;
    b->mode = want ? TRK_MODE_COMPLEX_CODE : TRK_MODE_PLAIN;
    b->lds_table_floats = per_chip * (b->max_code_len + 64);
    b->chans_dirty = true;
    return GC_OK;
}

gc_status gc_trk_batch_set_code_complex(gc_trk_batch* b, int ch, const float* code_iq, int code_length, const float* shifts_chips)
{
    GC_REQUIRE(b && code_iq && shifts_chips, "gc_trk_batch_set_code_complex: NULL argument");
    GC_REQUIRE(ch >= 0 && ch < b->n_channels, "gc_trk_batch_set_code_complex: channel %d out of range", ch);
    GC_REQUIRE(code_length > 0 && code_length <= b->max_code_len, "gc_trk_batch_set_code_complex: code_length %d not in 1..%d",
        code_length, b->max_code_len);
    if (!b->complex_codes) return gc_fail(GC_ERR_STATE, "gc_trk_batch_set_code_complex: call gc_trk_batch_set_complex_codes(batch, 1) first");
    gc_device_guard g(b->ctx->device);
    std::lock_guard<std::mutex> lk(b->ctx->mtx);
    GC_HIP(hipStreamSynchronize(b->ctx->stream));
    GC_HIP(hipMemcpy(b->d_codes + (size_t)ch * 2 * b->max_code_len, code_iq, sizeof(float) * 2 * code_length, hipMemcpyHostToDevice));
    b->h_chans[ch].code_len = code_length;
    return gc_trk_batch_set_shifts(b, ch, shifts_chips);
}

gc_status gc_trk_batch_set_16sc(gc_trk_batch* b, int on)
{
    GC_REQUIRE(b, "gc_trk_batch_set_16sc: NULL argument");
    const bool want = on != 0;
    if (want == b->sc16) return GC_OK;
    if (want && b->mode != TRK_MODE_PLAIN)
        return gc_fail(GC_ERR_STATE, "gc_trk_batch_set_16sc: not available with high dynamics or complex float codes");
    b->sc16 = false;  // the format setter refuses anything but cshort while the flag is up
    gc_status s = gc_trk_batch_set_input_format(b, want ? GC_IQ_I16 : GC_IQ_F32);  // drops the registered inputs
    if (s != GC_OK) return s;
    for (auto& c : b->h_chans) c.code_len = 0;  // every channel needs its code again (4 bytes per chip either way)
    b->sc16 = want;
    b->mode = want ? TRK_MODE_SC16 : TRK_MODE_PLAIN;
    b->chans_dirty = true;
    return GC_OK;
}

gc_status gc_trk_batch_set_code_16sc(gc_trk_batch* b, int ch, const int16_t* code_iq, int code_length, const float* shifts_chips)
{
    GC_REQUIRE(b && code_iq && shifts_chips, "gc_trk_batch_set_code_16sc: NULL argument");
    GC_REQUIRE(ch >= 0 && ch < b->n_channels, "gc_trk_batch_set_code_16sc: channel %d out of range", ch);
    GC_REQUIRE(code_length > 0 && code_length <= b->max_code_len, "gc_trk_batch_set_code_16sc: code_length %d not in 1..%d",
        code_length, b->max_code_len);
    if (!b->sc16) return gc_fail(GC_ERR_STATE, "gc_trk_batch_set_code_16sc: call gc_trk_batch_set_16sc(batch, 1) first");
    gc_device_guard g(b->ctx->device);
    std::lock_guard<std::mutex> lk(b->ctx->mtx);
    GC_HIP(hipStreamSynchronize(b->ctx->stream));
    GC_HIP(hipMemcpy(b->d_codes + (size_t)ch * b->max_code_len, code_iq, sizeof(int16_t) * 2 * code_length, hipMemcpyHostToDevice));
    b->h_chans[ch].code_len = code_length;
    return gc_trk_batch_set_shifts(b, ch, shifts_chips);
}

gc_status gc_trk_batch_set_input_dev(gc_trk_batch* b, int ch, const void* dev_iq, uint64_t n_samples)
{
    GC_REQUIRE(b && dev_iq, "gc_trk_batch_set_input_dev: NULL argument");
    GC_REQUIRE(ch >= 0 && ch < b->n_channels, "gc_trk_batch_set_input_dev: channel %d out of range", ch);
    const uintptr_t es = b->iq_format == GC_IQ_F32 ? 8 : b->iq_format == GC_IQ_I16 ? 4 : 2;
    GC_REQUIRE((reinterpret_cast<uintptr_t>(dev_iq) % es) == 0, "gc_trk_batch_set_input_dev: IQ pointer must be aligned to one sample (%d bytes)", (int)es);
    b->h_chans[ch].iq = dev_iq;
    b->h_chans[ch].n_iq = n_samples;
    b->h_chans[ch].ring_len = 0;
    if (b->streams[ch]) gc_stream_drop(b->streams[ch]);
    b->streams[ch] = nullptr;
    b->chans_dirty = true;
    return GC_OK;
}

gc_status gc_trk_batch_set_input_stream(gc_trk_batch* b, int ch, gc_stream* s)
{
    GC_REQUIRE(b && s, "gc_trk_batch_set_input_stream: NULL argument");
    GC_REQUIRE(ch >= 0 && ch < b->n_channels, "gc_trk_batch_set_input_stream: channel %d out of range", ch);
    GC_REQUIRE(s->ctx->device == b->ctx->device, "gc_trk_batch_set_input_stream: the stream lives on another GPU");
    GC_REQUIRE(s->iq_format == b->iq_format, "gc_trk_batch_set_input_stream: stream format %d, batch format %d", s->iq_format, b->iq_format);
    gc_stream_keep(s);
    if (b->streams[ch]) gc_stream_drop(b->streams[ch]);
    b->h_chans[ch].iq = s->d_ring;
    b->h_chans[ch].n_iq = ~0ull;
    b->h_chans[ch].ring_len = (unsigned)s->capacity;
    b->streams[ch] = s;
    b->chans_dirty = true;
    return GC_OK;
}

gc_status gc_trk_batch_set_read_floor(gc_trk_batch* b, uint64_t oldest_index_read)
{
    GC_REQUIRE(b, "gc_trk_batch_set_read_floor: NULL argument");
    b->has_read_floor = true;
    b->read_floor = oldest_index_read;
    return GC_OK;
}

gc_status gc_trk_batch_set_input_format(gc_trk_batch* b, int iq_format)
{
    GC_REQUIRE(b, "gc_trk_batch_set_input_format: NULL argument");
    GC_REQUIRE(iq_format == GC_IQ_F32 || iq_format == GC_IQ_I16 || iq_format == GC_IQ_I8, "gc_trk_batch_set_input_format: unknown format %d", iq_format);
    if (b->sc16 && iq_format != GC_IQ_I16) return gc_fail(GC_ERR_STATE, "gc_trk_batch_set_input_format: 16-bit mode correlates lv_16sc_t input only");
    if (iq_format != b->iq_format)
        {
            // pointers registered so far were checked against the old sample size
            for (auto& c : b->h_chans)
                {
                    c.iq = nullptr;
                    c.n_iq = 0;
                    c.ring_len = 0;
                }
            for (gc_stream*& r : b->streams)
                {
                    if (r) gc_stream_drop(r);
                    r = nullptr;
                }
            b->chans_dirty = true;
        }
    b->iq_format = iq_format;
    return GC_OK;
}

// engine tuning knobs (not part of the reference surface)
gc_status gc_trk_batch_set_nominal_length(gc_trk_batch* b, int n_samples)
{
    GC_REQUIRE(b, "gc_trk_batch_set_nominal_length: NULL argument");
    b->nominal_len = n_samples;
    return GC_OK;
}

gc_status gc_trk_batch_set_slices(gc_trk_batch* b, int n_slices)
{
    GC_REQUIRE(b && n_slices >= -1 && n_slices <= 1024, "gc_trk_batch_set_slices: bad argument");
    b->lds_sizing = n_slices >= 0;
    b->forced_slices = n_slices > 0 ? n_slices : 0;
    return GC_OK;
}

// distinct rings the batch reads
static std::vector<gc_stream*> batch_streams(const gc_trk_batch* b)
{
    std::vector<gc_stream*> v;
    for (gc_stream* s : b->streams)
        if (s && std::find(v.begin(), v.end(), s) == v.end()) v.push_back(s);
    return v;
}

// floors / ends: oldest absolute index the launch reads from each ring of batch_streams(b) and one past the newest (NULL: unknown).
// The reader slots are reserved BEFORE the residency check and the enqueue, so a push on the producer's thread cannot evict
// what this launch was validated for (gc_reader_table.h).
static gc_status batch_launch(gc_trk_batch* b, int n_epochs, const gc_epoch_params* dev_params, void* dev_out,
    hipStream_t st, int max_len, const std::vector<uint64_t>* floors = nullptr, const std::vector<uint64_t>* ends = nullptr, float max_step = 0.0f)
{
    for (int i = 0; i < b->n_channels; i++)
        {
            GC_REQUIRE(b->h_chans[i].code_len > 0, "gc_trk_batch_run: channel %d has no code (gc_trk_batch_set_code)", i);
            GC_REQUIRE(b->h_chans[i].iq != nullptr, "gc_trk_batch_run: channel %d has no input (gc_trk_batch_set_input_dev)", i);
        }
    gc_status s = batch_sync_chans(b, st);
    if (s != GC_OK) return s;
    int n_slices = pick_slices(b, n_epochs, max_len > 0 ? max_len : b->nominal_len);
    // LDS of the launch: what one slice of a nominal epoch touches, not the capacity of the longest code (Galileo E1: 33 KB per
    // workgroup = 4 waves per SIMD; half a period per slice: 17 KB = 8).  Known on the host: the nominal window length, the code
    // lengths, the tap shifts; a code period per nominal window gives the chips per sample.  A record that exceeds the bound (a code
    // step far off nominal, a longer window) is served by the kernel from the whole table in LDS when it fits the launch's LDS, from
    // global memory otherwise (trk_epoch, GTAB): the bound decides speed, never results.
    int lds_floats = b->lds_table_floats;
    const int len = max_len > 0 ? max_len : b->nominal_len;
    // (the high-dynamics resampler's window is the whole epoch whatever the slice: those modes keep the whole table)
    if (len >= 4096 && b->lds_sizing && (b->mode == TRK_MODE_PLAIN || b->mode == TRK_MODE_COMPLEX_CODE) && trk_small_window_ok(b->mode, b->iq_format))
        {
            int l_max = 0;
            float spread = 0.0f;
            for (int i = 0; i < b->n_channels; i++)
                {
                    l_max = std::max(l_max, b->h_chans[i].code_len);
                    float lo = b->h_chans[i].shifts[0], hi = lo;
                    for (int t = 1; t < b->n_taps; t++)
                        {
                            lo = std::min(lo, b->h_chans[i].shifts[t]);
                            hi = std::max(hi, b->h_chans[i].shifts[t]);
                        }
                    spread = std::max(spread, hi - lo);
                }
            const int per_chip = b->complex_codes ? 2 : 1;
            const int window_target = 4096;  // floats of LDS per workgroup that still leave 8 waves per SIMD (9 workgroups of ~17 KB per CU)
            if (b->forced_slices == 0 && l_max + 64 > window_target + 128 && b->mode == TRK_MODE_PLAIN)
                n_slices = std::max(n_slices, (l_max + window_target - 1) / window_target);
            if (n_slices > 1)
                {
                    const int chunks = (len + 16 + 511) / 512;  // + the alignment lead-in of at most 8 sample pairs
                    const int cps = (chunks + n_slices - 1) / n_slices;
                    // chips per sample: the largest |code step| of the records when the caller's parameter array is on the host
                    // (gc_trk_batch_run), else one code period per nominal window (device-resident records: a batch whose windows
                    // span several code periods should state so with gc_trk_batch_set_slices(b, -1))
                    const double chips_per_sample = max_step > 0.0f ? (double)max_step : (double)l_max / (double)len;
                    const int need = per_chip * ((int)std::ceil((double)cps * 512.0 * chips_per_sample * 1.002 + (double)spread) + 64);
                    // (a table that fits the 8-waves budget whole is left whole: nothing to gain, and windows of several code periods keep
                    // the LDS modulo path)
                    if (need < lds_floats && lds_floats > window_target + 128) lds_floats = std::max((need + 63) & ~63, 1024);
                }
        }
    if (n_slices > 1)
        {
            size_t need = (size_t)b->n_channels * n_epochs * n_slices * b->n_taps;
            if (need > b->partial_cap)
                {
                    GC_HIP(hipStreamSynchronize(st));
                    (void)hipFree(b->d_partial);
                    b->d_partial = nullptr;
                    b->partial_cap = 0;
                    GC_HIP(hipMalloc(&b->d_partial, need * sizeof(float2)));
                    b->partial_cap = need;
                }
        }
    const std::vector<gc_stream*> rings = batch_streams(b);
    std::vector<gc_stream_ticket> tickets(rings.size());
    auto cancel_all = [&]() {
        for (size_t i = 0; i < rings.size(); i++) gc_stream_cancel_read(rings[i], tickets[i]);
    };
    for (size_t i = 0; i < rings.size(); i++)
        {
            // later pushes may evict everything below the floor while this launch is still running; the newest push must have landed
            const uint64_t floor = (floors && (*floors)[i] != ~0ull) ? (*floors)[i] : b->has_read_floor ? b->read_floor : GC_STREAM_FLOOR_OLDEST;
            gc_status rs = gc_stream_begin_read(rings[i], st, floor, &tickets[i]);
            if (rs == GC_OK && ends && (*ends)[i] > tickets[i].head)
                rs = gc_fail(GC_ERR_INVALID, "gc_trk_batch_run: a window ends at sample %llu, the stream's resident samples are [%llu, %llu)",
                    (unsigned long long)(*ends)[i], (unsigned long long)tickets[i].oldest, (unsigned long long)tickets[i].head);
            if (rs != GC_OK)
                {
                    cancel_all();
                    return rs;
                }
        }
    hipError_t e = trk_launch(b->n_taps, b->mode, b->iq_format, st, b->d_chans, dev_params, static_cast<float2*>(dev_out), b->d_partial,
        b->n_channels, n_epochs, n_slices, lds_floats, true);
    if (e != hipSuccess)
        {
            cancel_all();
            return gc_fail(GC_ERR_HIP, "tracking kernel launch failed: %s", hipGetErrorString(e));
        }
    gc_status out = GC_OK;
    for (size_t i = 0; i < rings.size(); i++)
        {
            gc_status rs = gc_stream_end_read(rings[i], st, tickets[i]);
            if (rs != GC_OK) out = rs;
        }
    return out;
}

gc_status gc_trk_batch_run_dev(gc_trk_batch* b, int n_epochs, const gc_epoch_params* dev_params, void* dev_out, void* stream)
{
    GC_REQUIRE(b && dev_params && dev_out, "gc_trk_batch_run_dev: NULL argument");
    GC_REQUIRE(n_epochs > 0, "gc_trk_batch_run_dev: n_epochs must be > 0");
    gc_device_guard g(b->ctx->device);
    std::lock_guard<std::mutex> lk(b->ctx->mtx);
    return batch_launch(b, n_epochs, dev_params, dev_out, gc_pick_stream(b->ctx, stream), 0);
}

gc_status gc_trk_batch_run(gc_trk_batch* b, int n_epochs, const gc_epoch_params* host_params, float* host_out)
{
    GC_REQUIRE(b && host_params && host_out, "gc_trk_batch_run: NULL argument");
    GC_REQUIRE(n_epochs > 0, "gc_trk_batch_run: n_epochs must be > 0");
    gc_device_guard g(b->ctx->device);
    std::lock_guard<std::mutex> lk(b->ctx->mtx);
    const size_t jobs = (size_t)b->n_channels * n_epochs;
    int max_len = 0;
    float max_step = 0.0f;
    const std::vector<gc_stream*> rings = batch_streams(b);
    std::vector<uint64_t> floors(rings.size(), ~0ull), ends(rings.size(), 0);
    for (size_t j = 0; j < jobs; j++)
        {
            const gc_epoch_params& p = host_params[j];
            const TrkChan& c = b->h_chans[j / n_epochs];
            GC_REQUIRE(p.n_samples >= 0, "gc_trk_batch_run: job %zu has negative n_samples", j);
            if (gc_stream* r = b->streams[j / n_epochs])
                {
                    uint64_t oldest = 0, head = 0;
                    gc_stream_info(r, &oldest, &head, nullptr);
                    GC_REQUIRE((uint64_t)p.n_samples <= r->mirror, "gc_trk_batch_run: job %zu is longer than the stream's max_window %llu", j,
                        (unsigned long long)r->mirror);
                    GC_REQUIRE(p.sample_offset >= oldest && p.sample_offset + (uint64_t)p.n_samples <= head,
                        "gc_trk_batch_run: job %zu window [%llu, +%d) is not inside the stream's resident samples [%llu, %llu)", j,
                        (unsigned long long)p.sample_offset, p.n_samples, (unsigned long long)oldest, (unsigned long long)head);
                    const size_t ri = std::find(rings.begin(), rings.end(), r) - rings.begin();
                    floors[ri] = std::min(floors[ri], p.sample_offset);
                    ends[ri] = std::max(ends[ri], p.sample_offset + (uint64_t)p.n_samples);
                }
            else
                {
                    GC_REQUIRE(p.sample_offset + (uint64_t)p.n_samples <= c.n_iq,
                        "gc_trk_batch_run: job %zu window [%llu, +%d) exceeds the channel's %llu samples", j,
                        (unsigned long long)p.sample_offset, p.n_samples, (unsigned long long)c.n_iq);
                }
            if (p.n_samples > max_len) max_len = p.n_samples;
            if (p.n_samples > 0) max_step = std::max(max_step, std::fabs(p.code_phase_step_chips) + std::fabs(p.code_phase_rate_step_chips) * (float)p.n_samples);
        }
    hipStream_t st = b->ctx->stream;
    if (jobs > b->params_cap)
        {
            (void)hipFree(b->d_params);
            b->d_params = nullptr;
            b->params_cap = 0;
            GC_HIP(hipMalloc(&b->d_params, jobs * sizeof(gc_epoch_params)));
            b->params_cap = jobs;
        }
    if (jobs * b->n_taps > b->out_cap)
        {
            (void)hipFree(b->d_out);
            b->d_out = nullptr;
            b->out_cap = 0;
            GC_HIP(hipMalloc(&b->d_out, jobs * b->n_taps * sizeof(float2)));
            b->out_cap = jobs * b->n_taps;
        }
    GC_HIP(hipMemcpyAsync(b->d_params, host_params, jobs * sizeof(gc_epoch_params), hipMemcpyHostToDevice, st));
    gc_status s = batch_launch(b, n_epochs, b->d_params, b->d_out, st, max_len, &floors, &ends, max_step);
    if (s != GC_OK) return s;
    // 16-bit mode: n_taps lv_16sc_t (4 bytes) per job instead of n_taps complex floats
    const size_t out_elem = b->sc16 ? sizeof(short2) : sizeof(float2);
    GC_HIP(hipMemcpyAsync(host_out, b->d_out, jobs * b->n_taps * out_elem, hipMemcpyDeviceToHost, st));
    GC_HIP(hipStreamSynchronize(st));
    return GC_OK;
}

}  // extern "C"

// -----------------------------------------------------------------------------
// Level 1: Cpu_Multicorrelator_Real_Codes image
// -----------------------------------------------------------------------------
struct gc_correlator
{
    gc_ctx* ctx = nullptr;
    gc_ctx_ref ctx_ref;
    bool use_high_dynamics_resampler = true;  // reference ctor default (cpu_multicorrelator_real_codes.cc:49)
    bool inited = false;
    int max_len = 0, n_corr = 0;
    // retained caller pointers (the reference stores pointers, .cc:79-98)
    const float* local_code_in = nullptr;
    float* shifts_chips = nullptr;
    int code_length_chips = 0;
    // what local_code_in points at: float chips (Cpu_Multicorrelator_Real_Codes), (re, im) float pairs
    // (Cpu_Multicorrelator), or (re, im) int16 pairs with int16 input/output (Cpu_Multicorrelator_16sc)
    enum CodeKind
    {
        CODE_REAL,
        CODE_COMPLEX,
        CODE_SC16
    };
    CodeKind code_kind = CODE_REAL;
    float* corr_out = nullptr;      // n_corr complex floats, or n_corr lv_16sc_t when io_16sc
    const float* sig_in = nullptr;  // complex floats, or lv_16sc_t when io_16sc
    bool io_16sc = false;
    // device side
    float2* d_sig = nullptr;
    float* d_code = nullptr;
    int d_code_cap = 0;
    std::vector<float> code_shadow;  // last uploaded code contents
    struct Staging
    {
        TrkChan chan;
        gc_epoch_params params;
    };
    Staging* h_stage = nullptr;  // pinned
    Staging* d_stage = nullptr;
    float2* d_out = nullptr;
    float2* h_out = nullptr;  // pinned
    float2* d_partial = nullptr;
    int partial_slices = 0;
    // zero-copy path: the window is copied into page-locked, device-mapped host memory and the kernel reads it (and the
    // descriptor) over PCIe and writes the result back the same way -- one launch and one synchronisation per call
    // instead of three copies around the launch
    bool zero_copy = true;
    char* h_sig = nullptr;            // pinned, mapped
    const void* dv_sig = nullptr;     // device view of h_sig
    Staging* dv_stage = nullptr;      // device view of h_stage
    float2* dv_out = nullptr;         // device view of h_out
};

static void correlator_release(gc_correlator* c)
{
    (void)hipFree(c->d_sig);
    (void)hipFree(c->d_code);
    (void)hipFree(c->d_stage);
    (void)hipFree(c->d_out);
    (void)hipFree(c->d_partial);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->h_sig) (void)hipHostFree(c->h_sig);
    c->h_sig = nullptr;
    c->dv_sig = nullptr;
    c->dv_stage = nullptr;
    c->dv_out = nullptr;
    c->d_sig = nullptr;
    c->d_code = nullptr;
    c->d_stage = nullptr;
    c->d_out = nullptr;
    c->d_partial = nullptr;
    c->h_stage = nullptr;
    c->h_out = nullptr;
    c->d_code_cap = 0;
    c->code_shadow.clear();
    c->inited = false;
}

// -----------------------------------------------------------------------------
// Level-1 epoch batcher: queue / lanes / waiters / window sharing / host-buffer registry live in gc_l1_batcher.h (header-only,
// no HIP types, exercised on the CPU under ThreadSanitizer by tests/l1_batcher_selftest.cpp); this is its HIP backend: a lane
// is a HIP stream with page-locked, device-mapped descriptor / result arrays, a launch is ONE trk_launch over the batch.
// -----------------------------------------------------------------------------
// host (mapped / registered) -> HBM copy of a shared window on the lane's stream: a kernel launch costs the calling thread
// less than a hipMemcpyAsync, and the tracking kernel behind it needs no other ordering
__global__ void l1_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, int tail_bytes)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail_bytes)  // never reads past the end of the caller's memory
        reinterpret_cast<char*>(dst + n16)[threadIdx.x] = reinterpret_cast<const char*>(src + n16)[threadIdx.x];
}

struct L1HipBackend
{
    typedef TrkChan Chan;
    typedef gc_epoch_params Params;
    static constexpr int MAX_SLICES = 64;
    struct Lane
    {
        hipStream_t stream = nullptr;
        TrkChan* h_chans = nullptr;  // pinned, mapped
        gc_epoch_params* h_params = nullptr;
        char* h_out = nullptr;
        TrkChan* dv_chans = nullptr;
        gc_epoch_params* dv_params = nullptr;
        float2* dv_out = nullptr;
        float2* d_partial = nullptr;
        char* d_span = nullptr;  // windows shared by several requests of a batch, copied once
        size_t span_cap = 0;
        char* h_span = nullptr;         // page-locked, mapped: where the leader stages unions of overlapping unregistered windows
        const char* dv_span = nullptr;  // its device view
        size_t hspan_cap = 0;
    };
    int device = 0;
    struct Guard
    {
        gc_device_guard g;
        explicit Guard(L1HipBackend& b) : g(b.device) {}
    };
    bool lane_init(Lane& l, int maxb, int max_out_bytes)
    {
        void* dv = nullptr;
        bool ok = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&l.h_chans), sizeof(TrkChan) * maxb, hipHostMallocMapped) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&l.h_params), sizeof(gc_epoch_params) * maxb, hipHostMallocMapped) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&l.h_out), (size_t)max_out_bytes * maxb, hipHostMallocMapped) == hipSuccess;
        ok = ok && hipHostGetDevicePointer(&dv, l.h_chans, 0) == hipSuccess;
        l.dv_chans = static_cast<TrkChan*>(dv);
        ok = ok && hipHostGetDevicePointer(&dv, l.h_params, 0) == hipSuccess;
        l.dv_params = static_cast<gc_epoch_params*>(dv);
        ok = ok && hipHostGetDevicePointer(&dv, l.h_out, 0) == hipSuccess;
        l.dv_out = static_cast<float2*>(dv);
        ok = ok && hipMalloc(&l.d_partial, sizeof(float2) * GC_MAX_TAPS * MAX_SLICES * maxb) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        return ok;
    }
    void lane_free(Lane& l)
    {
        if (l.stream)
            {
                (void)hipStreamSynchronize(l.stream);
                (void)hipStreamDestroy(l.stream);
            }
        if (l.h_chans) (void)hipHostFree(l.h_chans);
        if (l.h_params) (void)hipHostFree(l.h_params);
        if (l.h_out) (void)hipHostFree(l.h_out);
        (void)hipFree(l.d_partial);
        (void)hipFree(l.d_span);
        if (l.h_span) (void)hipHostFree(l.h_span);
        l = Lane();
    }
    // grows the lane's shared-window buffer (the lane is idle: nothing reads the old one)
    bool span_reserve(Lane& lane, size_t need)
    {
        if (need <= lane.span_cap) return true;
        (void)hipFree(lane.d_span);
        lane.d_span = nullptr;
        lane.span_cap = 0;
        need = (need + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
        if (hipMalloc(&lane.d_span, need) != hipSuccess)
            {
                (void)hipGetLastError();
                return false;
            }
        lane.span_cap = need;
        return true;
    }
    bool hspan_reserve(Lane& lane, size_t need)
    {
        if (need <= lane.hspan_cap) return true;
        if (lane.h_span) (void)hipHostFree(lane.h_span);
        lane.h_span = nullptr;
        lane.hspan_cap = 0;
        need = (need + ((size_t)4 << 20)) & ~(((size_t)1 << 20) - 1);  // page-locking is slow: grow in large steps
        void* dv = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&lane.h_span), need, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dv, lane.h_span, 0) != hipSuccess)
            {
                (void)hipGetLastError();
                if (lane.h_span) (void)hipHostFree(lane.h_span);
                lane.h_span = nullptr;
                return false;
            }
        lane.dv_span = static_cast<const char*>(dv);
        lane.hspan_cap = need;
        return true;
    }
    // src_dev_view and dst are 16-byte aligned
    int copy(Lane& lane, const void* src_dev_view, void* dst, size_t bytes)
    {
        const size_t n16 = bytes / 16;
        const unsigned blocks = (unsigned)std::min<size_t>(256, (n16 + 255) / 256);
        hipLaunchKernelGGL(l1_copy_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, lane.stream, static_cast<const uint4*>(src_dev_view), static_cast<uint4*>(dst), n16,
            (int)(bytes - n16 * 16));
        return (int)hipGetLastError();
    }
    template <class Rq>
    int launch(Lane& lane, const Rq& k, int B, int lds_floats)
    {
        return (int)trk_launch(k.n_corr, k.mode, k.fmt, lane.stream, lane.dv_chans, lane.dv_params, lane.dv_out, lane.d_partial, B, 1, k.n_slices, lds_floats, false);
    }
    int wait(Lane& lane) { return (int)hipStreamSynchronize(lane.stream); }
    int oom_error() const { return (int)hipErrorOutOfMemory; }
    const char* error_string(int e) const { return hipGetErrorString((hipError_t)e); }
    void host_unregister(const void* base) { (void)hipHostUnregister(const_cast<void*>(base)); }
};

typedef gc_l1_batcher_t<L1HipBackend> gc_l1_batcher;
typedef gc_l1_batcher::Request L1Request;

// the backend lives beside the batcher that points at it
struct L1Holder
{
    L1HipBackend be;
    gc_l1_batcher* bat = nullptr;
    ~L1Holder() { delete bat; }
};

static void l1_batcher_free(void* p)
{
    L1Holder* h = static_cast<L1Holder*>(p);
    gc_device_guard g(h->be.device);
    delete h;
}

// the context's batcher (created by the first caller; NULL if its buffers cannot be set up: callers then use the direct path)
static gc_l1_batcher* l1_batcher_get(gc_ctx* ctx)
{
    auto usable = [](void* p) -> gc_l1_batcher* {
        L1Holder* h = static_cast<L1Holder*>(p);
        return h->bat->ok() ? h->bat : nullptr;
    };
    if (void* p = ctx->l1_batcher.load(std::memory_order_acquire)) return usable(p);
    std::lock_guard<std::mutex> lk(ctx->mtx);
    if (void* p = ctx->l1_batcher.load(std::memory_order_acquire)) return usable(p);
    L1Holder* h = new L1Holder();
    h->be.device = ctx->device;
    h->bat = new gc_l1_batcher(&h->be, (int)(sizeof(float2) * GC_MAX_TAPS));
    if (const char* e = gc_exp_env("GNSSCORR_L1_SECOND_LANE")) h->bat->min_second_lane = std::max(1, std::atoi(e));
    ctx->l1_batcher_free = &l1_batcher_free;
    ctx->l1_batcher.store(h, std::memory_order_release);
    return usable(h);
}

static gc_status l1_submit(gc_l1_batcher* b, L1Request* rq, void* own_pinned, const void* own_pinned_dev)
{
    if (b->submit(rq, own_pinned, own_pinned_dev) != gc_l1_batcher::ST_OK) return gc_fail(GC_ERR_HIP, "%s", rq->err);
    return GC_OK;
}

static gc_status correlator_run(gc_correlator* c, int mode, float rem_carr, float phase_step, float phase_rate_step,
    float rem_code, float code_step, float code_rate_step, int N)
{
    GC_REQUIRE(c, "correlator: NULL handle");
    if (!c->inited) return gc_fail(GC_ERR_STATE, "correlator: init() has not been called");
    if (!c->local_code_in || !c->shifts_chips) return gc_fail(GC_ERR_STATE, "correlator: set_local_code_and_taps() has not been called");
    if (!c->corr_out || !c->sig_in) return gc_fail(GC_ERR_STATE, "correlator: set_input_output_vectors() has not been called");
    GC_REQUIRE(N >= 0 && N <= c->max_len, "correlator: signal_length_samples %d exceeds init() capacity %d", N, c->max_len);
    const int per_chip = (mode == TRK_MODE_COMPLEX_CODE) ? 2 : 1;  // floats (4-byte words) per chip
    const gc_correlator::CodeKind want_kind = (mode == TRK_MODE_COMPLEX_CODE) ? gc_correlator::CODE_COMPLEX
                                              : (mode == TRK_MODE_SC16)       ? gc_correlator::CODE_SC16
                                                                              : gc_correlator::CODE_REAL;
    if (want_kind != c->code_kind)
        return gc_fail(GC_ERR_STATE, "correlator: the local code is %s; use the matching setters and Carrier_wipeoff overload",
            c->code_kind == gc_correlator::CODE_COMPLEX ? "complex float" : c->code_kind == gc_correlator::CODE_SC16 ? "16-bit complex" : "real");
    const bool sc16 = (mode == TRK_MODE_SC16);
    if (sc16 != c->io_16sc) return gc_fail(GC_ERR_STATE, "correlator: input/output vectors and local code must both be 16-bit or both float");
    GC_REQUIRE(c->code_length_chips > 0 && per_chip * (c->code_length_chips + 64) <= kMaxLdsTableFloats,
        "correlator: code_length_chips %d not supported (max %d)", c->code_length_chips, kMaxLdsTableFloats / per_chip - 64);
    gc_device_guard g(c->ctx->device);
    static const bool batching = [] {
        const char* e = gc_exp_env("GNSSCORR_L1_BATCH");  // 0: every call is its own launch under the context mutex (round-1 behaviour)
        return !(e && e[0] == '0');
    }();
    gc_l1_batcher* bat = (c->zero_copy && batching) ? l1_batcher_get(c->ctx) : nullptr;
    // The object's own buffers (code table, page-locked window, staging) belong to the calling thread -- one correlator per
    // channel thread, like the reference's -- so with the batcher only the rare code upload takes the context mutex.
    std::unique_lock<std::mutex> lk(c->ctx->mtx, std::defer_lock);
    if (!bat) lk.lock();
    hipStream_t st = c->ctx->stream;
    const int L = c->code_length_chips;
    const int LF = per_chip * L;  // floats in the code table
    // code table: the caller's buffer is re-read on every call like the reference
    // does; it is re-uploaded only when its contents changed
    if (LF > c->d_code_cap || (int)c->code_shadow.size() != LF || std::memcmp(c->code_shadow.data(), c->local_code_in, sizeof(float) * LF) != 0)
        {
            if (bat) lk.lock();
            if (LF > c->d_code_cap)
                {
                    (void)hipFree(c->d_code);
                    c->d_code = nullptr;
                    c->d_code_cap = 0;
                    GC_HIP(hipMalloc(&c->d_code, sizeof(float) * LF));
                    c->d_code_cap = LF;
                }
            // the previous upload may still be in flight from the pageable shadow buffer
            GC_HIP(hipStreamSynchronize(st));
            c->code_shadow.assign(c->local_code_in, c->local_code_in + LF);
            GC_HIP(hipMemcpyAsync(c->d_code, c->code_shadow.data(), sizeof(float) * LF, hipMemcpyHostToDevice, st));
            if (bat)
                {
                    GC_HIP(hipStreamSynchronize(st));  // batches run on the lanes' streams: the table must have landed
                    lk.unlock();
                }
        }
    const size_t sig_bytes = (sc16 ? sizeof(short2) : sizeof(float2)) * (size_t)N;
    const bool zc = c->zero_copy;
    if (N > 0 && !bat)
        {
            if (zc)
                std::memcpy(c->h_sig, c->sig_in, sig_bytes);
            else
                GC_HIP(hipMemcpyAsync(c->d_sig, c->sig_in, sig_bytes, hipMemcpyHostToDevice, st));
        }
    // one epoch only: cut it in slices so that the launch covers many CUs.  The slice count depends on the window length alone,
    // so a call gives the same sums whether it runs alone or inside a batch.
    int chunks = (N + 1 + 511) / 512;
    int n_slices = chunks / 2;
    static const int one_wg_max = [] {
        const char* e = gc_exp_env("GNSSCORR_L1_ONE_WG_MAX");  // tuning knob: windows up to this length run in one workgroup
        return e ? std::atoi(e) : 2048;  // measured: beyond ~2000 samples several workgroups + the partial-sum launch win
    }();
    if (N <= one_wg_max) n_slices = 1;
    if (n_slices < 1) n_slices = 1;
    if (n_slices > c->partial_slices) n_slices = c->partial_slices;
    const size_t out_bytes = (sc16 ? sizeof(short2) : sizeof(float2)) * c->n_corr;
    if (bat)
        {
            L1Request rq;
            std::memset(&rq.chan, 0, sizeof rq.chan);
            rq.chan.iq = nullptr;  // set by the batcher: registered memory, this call's page-locked copy, or a shared HBM copy
            rq.chan.n_iq = (unsigned long long)N;
            rq.chan.code = c->d_code;
            rq.chan.code_len = L;
            for (int t = 0; t < c->n_corr; t++) rq.chan.shifts[t] = c->shifts_chips[t];
            gc_epoch_params_fill(&rq.params, 0, rem_carr, phase_step, phase_rate_step, rem_code, code_step, code_rate_step, N);
            rq.n_corr = c->n_corr;
            rq.mode = mode;
            rq.fmt = sc16 ? GC_IQ_I16 : GC_IQ_F32;
            rq.n_slices = n_slices;
            rq.lds_floats = per_chip * (L + 64);
            rq.host_sig = reinterpret_cast<const char*>(c->sig_in);
            rq.sig_bytes = sig_bytes;
            rq.out_host = c->corr_out;
            rq.out_bytes = out_bytes;
            return l1_submit(bat, &rq, c->h_sig, c->dv_sig);
        }
    gc_correlator::Staging* s = c->h_stage;
    std::memset(&s->chan, 0, sizeof s->chan);
    s->chan.iq = zc ? c->dv_sig : c->d_sig;
    s->chan.n_iq = (unsigned long long)N;
    s->chan.code = c->d_code;
    s->chan.code_len = L;
    for (int t = 0; t < c->n_corr; t++) s->chan.shifts[t] = c->shifts_chips[t];
    gc_epoch_params_fill(&s->params, 0, rem_carr, phase_step, phase_rate_step, rem_code, code_step, code_rate_step, N);
    if (!zc) GC_HIP(hipMemcpyAsync(c->d_stage, s, sizeof *s, hipMemcpyHostToDevice, st));
    gc_correlator::Staging* stage_dev = zc ? c->dv_stage : c->d_stage;
    hipError_t e = trk_launch(c->n_corr, mode, sc16 ? GC_IQ_I16 : GC_IQ_F32, st, &stage_dev->chan, &stage_dev->params, zc ? c->dv_out : c->d_out, c->d_partial, 1, 1,
        n_slices, per_chip * (L + 64), false);
    if (e != hipSuccess) return gc_fail(GC_ERR_HIP, "tracking kernel launch failed: %s", hipGetErrorString(e));
    if (!zc) GC_HIP(hipMemcpyAsync(c->h_out, c->d_out, out_bytes, hipMemcpyDeviceToHost, st));
    GC_HIP(hipStreamSynchronize(st));
    std::memcpy(c->corr_out, c->h_out, out_bytes);
    return GC_OK;
}

extern "C" {

gc_status gc_correlator_create(gc_ctx* ctx, gc_correlator** out)
{
    GC_REQUIRE(ctx && out, "gc_correlator_create: NULL argument");
    gc_correlator* c = new gc_correlator();
    c->ctx = ctx;
    c->ctx_ref.bind(ctx);
    *out = c;
    return GC_OK;
}

gc_status gc_correlator_destroy(gc_correlator* c)
{
    if (!c) return GC_OK;
    gc_device_guard g(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    correlator_release(c);
    delete c;
    return GC_OK;
}

gc_status gc_correlator_set_high_dynamics_resampler(gc_correlator* c, int use_high_dynamics_resampler)
{
    GC_REQUIRE(c, "gc_correlator_set_high_dynamics_resampler: NULL handle");
    c->use_high_dynamics_resampler = use_high_dynamics_resampler != 0;
    return GC_OK;
}

gc_status gc_correlator_init(gc_correlator* c, int max_signal_length_samples, int n_correlators)
{
    GC_REQUIRE(c, "gc_correlator_init: NULL handle");
    GC_REQUIRE(max_signal_length_samples > 0, "gc_correlator_init: max_signal_length_samples must be > 0");
    GC_REQUIRE(n_correlators >= 1 && n_correlators <= GC_MAX_TAPS, "gc_correlator_init: n_correlators must be in 1..%d", GC_MAX_TAPS);
    gc_device_guard g(c->ctx->device);
    std::lock_guard<std::mutex> lk(c->ctx->mtx);
    (void)hipStreamSynchronize(c->ctx->stream);
    correlator_release(c);
    c->max_len = max_signal_length_samples;
    c->n_corr = n_correlators;
    c->partial_slices = 64;
    GC_HIP(hipMalloc(&c->d_sig, sizeof(float2) * ((size_t)max_signal_length_samples + 2)));
    GC_HIP(hipMalloc(&c->d_stage, sizeof(gc_correlator::Staging)));
    GC_HIP(hipMalloc(&c->d_out, sizeof(float2) * GC_MAX_TAPS));
    GC_HIP(hipMalloc(&c->d_partial, sizeof(float2) * GC_MAX_TAPS * c->partial_slices));
    GC_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), sizeof(gc_correlator::Staging), hipHostMallocMapped));
    GC_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_out), sizeof(float2) * GC_MAX_TAPS, hipHostMallocMapped));
    {
        const char* e = gc_exp_env("GNSSCORR_L1_COPY");  // 1: stage the window in HBM with explicit copies instead
        c->zero_copy = !(e && e[0] == '1');
    }
    if (c->zero_copy)
        {
            void* dv = nullptr;
            hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->h_sig), sizeof(float2) * ((size_t)max_signal_length_samples + 2), hipHostMallocMapped);
            if (e == hipSuccess) e = hipHostGetDevicePointer(&dv, c->h_sig, 0);
            c->dv_sig = dv;
            if (e == hipSuccess) e = hipHostGetDevicePointer(&dv, c->h_stage, 0);
            c->dv_stage = static_cast<gc_correlator::Staging*>(dv);
            if (e == hipSuccess) e = hipHostGetDevicePointer(&dv, c->h_out, 0);
            c->dv_out = static_cast<float2*>(dv);
            if (e != hipSuccess)
                {
                    (void)hipGetLastError();
                    c->zero_copy = false;  // the copy path needs none of these
                }
        }
    c->inited = true;
    return GC_OK;
}

gc_status gc_correlator_set_local_code_and_taps(gc_correlator* c, int code_length_chips, const float* local_code_in,
    float* shifts_chips)
{
    GC_REQUIRE(c, "gc_correlator_set_local_code_and_taps: NULL handle");
    c->local_code_in = local_code_in;
    c->shifts_chips = shifts_chips;
    c->code_length_chips = code_length_chips;
    c->code_kind = gc_correlator::CODE_REAL;
    return GC_OK;
}

gc_status gc_correlator_set_local_code_and_taps_complex(gc_correlator* c, int code_length_chips, const float* local_code_in_iq,
    float* shifts_chips)
{
    GC_REQUIRE(c, "gc_correlator_set_local_code_and_taps_complex: NULL handle");
    c->local_code_in = local_code_in_iq;
    c->shifts_chips = shifts_chips;
    c->code_length_chips = code_length_chips;
    c->code_kind = gc_correlator::CODE_COMPLEX;
    return GC_OK;
}

gc_status gc_correlator_set_local_code_and_taps_16sc(gc_correlator* c, int code_length_chips, const int16_t* local_code_in_iq,
    float* shifts_chips)
{
    GC_REQUIRE(c, "gc_correlator_set_local_code_and_taps_16sc: NULL handle");
    c->local_code_in = reinterpret_cast<const float*>(local_code_in_iq);  // 4 bytes per chip, compared and uploaded as words
    c->shifts_chips = shifts_chips;
    c->code_length_chips = code_length_chips;
    c->code_kind = gc_correlator::CODE_SC16;
    return GC_OK;
}

gc_status gc_correlator_set_input_output_vectors_16sc(gc_correlator* c, int16_t* corr_out, const int16_t* sig_in)
{
    GC_REQUIRE(c, "gc_correlator_set_input_output_vectors_16sc: NULL handle");
    c->sig_in = reinterpret_cast<const float*>(sig_in);
    c->corr_out = reinterpret_cast<float*>(corr_out);
    c->io_16sc = true;
    return GC_OK;
}

gc_status gc_correlator_set_input_output_vectors(gc_correlator* c, float* corr_out, const float* sig_in)
{
    GC_REQUIRE(c, "gc_correlator_set_input_output_vectors: NULL handle");
    c->sig_in = sig_in;
    c->corr_out = corr_out;
    c->io_16sc = false;
    return GC_OK;
}

gc_status gc_correlator_carrier_wipeoff_multicorrelator_resampler(gc_correlator* c,
    float rem_carrier_phase_in_rad, float phase_step_rad, float phase_rate_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    int signal_length_samples)
{
    GC_REQUIRE(c, "gc_correlator_carrier_wipeoff_multicorrelator_resampler: NULL handle");
    const int mode = c->use_high_dynamics_resampler ? TRK_MODE_HD_FULL : TRK_MODE_PLAIN;
    return correlator_run(c, mode, rem_carrier_phase_in_rad, phase_step_rad, phase_rate_step_rad, rem_code_phase_chips,
        code_phase_step_chips, code_phase_rate_step_chips, signal_length_samples);
}

gc_status gc_correlator_carrier_wipeoff_multicorrelator_resampler_6(gc_correlator* c,
    float rem_carrier_phase_in_rad, float phase_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    int signal_length_samples)
{
    GC_REQUIRE(c, "gc_correlator_carrier_wipeoff_multicorrelator_resampler_6: NULL handle");
    const int mode = c->use_high_dynamics_resampler ? TRK_MODE_HD_RESAMPLER : TRK_MODE_PLAIN;
    return correlator_run(c, mode, rem_carrier_phase_in_rad, phase_step_rad, 0.0f, rem_code_phase_chips,
        code_phase_step_chips, code_phase_rate_step_chips, signal_length_samples);
}

gc_status gc_correlator_carrier_wipeoff_multicorrelator_resampler_5(gc_correlator* c,
    float rem_carrier_phase_in_rad, float phase_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips,
    int signal_length_samples)
{
    GC_REQUIRE(c, "gc_correlator_carrier_wipeoff_multicorrelator_resampler_5: NULL handle");
    return correlator_run(c, c->code_kind == gc_correlator::CODE_SC16 ? TRK_MODE_SC16 : TRK_MODE_COMPLEX_CODE, rem_carrier_phase_in_rad, phase_step_rad, 0.0f, rem_code_phase_chips,
        code_phase_step_chips, 0.0f, signal_length_samples);
}

gc_status gc_ctx_register_host_buffer(gc_ctx* ctx, const void* base, size_t bytes)
{
    GC_REQUIRE(ctx && base && bytes > 0, "gc_ctx_register_host_buffer: bad argument");
    gc_device_guard g(ctx->device);
    gc_l1_batcher* b = l1_batcher_get(ctx);
    if (!b) return gc_fail(GC_ERR_HIP, "gc_ctx_register_host_buffer: the batcher's buffers could not be set up");
    void* dv = nullptr;
    hipError_t e = hipHostRegister(const_cast<void*>(base), bytes, hipHostRegisterMapped);
    if (e == hipSuccess)
        {
            e = hipHostGetDevicePointer(&dv, const_cast<void*>(base), 0);
            if (e != hipSuccess) (void)hipHostUnregister(const_cast<void*>(base));
        }
    if (e != hipSuccess)
        {
            (void)hipGetLastError();
            return gc_fail(GC_ERR_HIP, "gc_ctx_register_host_buffer: %s (the correlators keep staging their windows)", hipGetErrorString(e));
        }
    b->add_region(base, bytes, dv);
    return GC_OK;
}

gc_status gc_ctx_unregister_host_buffer(gc_ctx* ctx, const void* base)
{
    GC_REQUIRE(ctx && base, "gc_ctx_unregister_host_buffer: bad argument");
    gc_device_guard g(ctx->device);
    void* p = ctx->l1_batcher.load(std::memory_order_acquire);
    if (!p) return gc_fail(GC_ERR_STATE, "gc_ctx_unregister_host_buffer: nothing is registered");
    // live registrations only; returns once no queued or running call reads through the buffer's device view and it is unpinned
    if (!static_cast<L1Holder*>(p)->bat->remove_region(base)) return gc_fail(GC_ERR_STATE, "gc_ctx_unregister_host_buffer: %p is not registered", base);
    return GC_OK;
}

gc_status gc_correlator_batch_stats(gc_ctx* ctx, uint64_t* n_batches, uint64_t* n_requests, uint64_t* n_shared_windows, int* max_batch)
{
    GC_REQUIRE(ctx, "gc_correlator_batch_stats: NULL context");
    gc_l1_batcher::Stats st;
    if (void* p = ctx->l1_batcher.load(std::memory_order_acquire))
        {
            st = static_cast<L1Holder*>(p)->bat->stats();
            if (const char* e = gc_exp_env("GNSSCORR_L1_TRACE"))
                if (e[0] == '1' && st.n_batches)
                    std::fprintf(stderr, "gnsscorr level-1 batcher: %llu batches, %llu calls; per batch: prepare %.1f us, launch %.1f us, wait %.1f us, scatter %.1f us; per call in the batcher %.1f us\n",
                        st.n_batches, st.n_requests, st.t_prep / st.n_batches, st.t_launch / st.n_batches, st.t_sync / st.n_batches, st.t_scatter / st.n_batches,
                        st.t_queue / (st.n_requests ? st.n_requests : 1));
        }
    if (n_batches) *n_batches = st.n_batches;
    if (n_requests) *n_requests = st.n_requests;
    if (n_shared_windows) *n_shared_windows = st.n_shared;
    if (max_batch) *max_batch = st.max_batch;
    return GC_OK;
}

gc_status gc_correlator_free(gc_correlator* c)
{
    GC_REQUIRE(c, "gc_correlator_free: NULL handle");
    gc_device_guard g(c->ctx->device);
    std::lock_guard<std::mutex> lk(c->ctx->mtx);
    (void)hipStreamSynchronize(c->ctx->stream);
    correlator_release(c);
    return GC_OK;
}

}  // extern "C"
