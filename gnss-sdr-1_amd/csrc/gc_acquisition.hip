// gc_acquisition.hip -- host side of the PCPS acquisition path of libgnsscorr.so.
// Mirrors pcps_acquisition (src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.cc):
// constructor sizes (:63-190), set_local_code (:239-274), init (:313-368),
// acquisition_core (:668-770) -- batched over n_sats satellites that search the
// same input block.
#include "acq_kernels.h"
#include "gc_internal.h"
#include "gc_stream.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

struct gc_acq
{
    gc_ctx* ctx = nullptr;
    gc_ctx_ref ctx_ref;
    gc_acq_conf conf{};
    int n_sats = 0;
    uint32_t fft_size = 0, consumed = 0, eff = 0, n_bins = 0;  // n_bins: bins of the ACTIVE grid
    uint32_t n_bins_main = 0, n_bins_alloc = 0;
    bool step_two = false;
    float center_step_two = 0.0f;
    float2* d_wipe_main = nullptr;
    float2* d_wipe2 = nullptr;
    bool use_cfar = false;
    uint32_t max_dwells = 1;
    uint32_t dwell_counter = 0;
    AcqFftPlan plan{};
    int n_blocks = 0;
    int sats_per_batch = 1;
    float2* d_wN = nullptr;
    float2* d_wN2 = nullptr;
    float2* d_wipe = nullptr;
    float2* d_codes = nullptr;
    float2* d_xw = nullptr;
    float2* d_X = nullptr;
    float2* d_Q = nullptr;
    float* d_grid = nullptr;
    float* d_tmp = nullptr;
    float* d_blkv = nullptr;
    unsigned* d_blki = nullptr;
    float* d_power = nullptr;
    float2* d_in = nullptr;
    float2* d_cvt = nullptr;  // converted input block (integer sample formats)
    int iq_format = GC_IQ_F32;
    gc_acq_result* d_results = nullptr;
    gc_acq_result* h_results = nullptr;  // pinned
    float* d_part_val = nullptr;      // acq_final_kernel: second-peak candidates per row piece
    unsigned* d_part_cnt = nullptr;   // and its ticket counters
    std::vector<char> code_set;
    int64_t freq_offset_hz = 0;  // d_old_freq: intermediate frequency / GLONASS FDMA channel offset
    AcqPhaseSeg* d_segs = nullptr;  // closed-form segments of the rows' running phases (acq_phase_segments.h) [seg_cap]
    int* d_seg_off = nullptr;       // [n_bins_alloc + 1]
    size_t seg_cap = 0;
    // the wipe-off tables hold what freq_offset_hz / the step-two centre say: cleared before a rebuild starts, set when it has
    // completed; a search on tables in an unknown state is refused (GC_ERR_STATE) instead of run
    bool wipe_valid = false, wipe2_valid = false;
    bool grid_logically_zero = true;  // gc_acq_reset() since the last dwell: the grid reads as zeros
    // The statistics of a dwell (acq_final_kernel) are computed when somebody can see them -- a fetch -- or when the next dwell
    // would not reproduce their side effect, not after every dwell: the reference evaluates the statistic after each dwell
    // (pcps_acquisition.cc:747-755), but a caller that enqueues the dwells of a search back to back and fetches once only ever
    // reads the last one.  The one side effect, the scratch image of d_tmp_buffer, is overwritten completely by the column pass
    // of an ACCUMULATING dwell (last-bin magnitudes, :737), so a pending evaluation may be dropped in front of such a dwell and must
    // run in front of any other.
    bool final_pending = false;
    AcqFinalArgs final_args{};
    hipStream_t final_stream = nullptr;
    // Of a dwell that more dwells are expected to follow (dwell counter < max_dwells, plain statistic) only the input block is taken
    // at once (row-permuted into slot 0 of d_xw); its transforms are held back: if the next call is an accumulating dwell on the same
    // stream, BOTH blocks go through one forward row / column pass (2 * n_bins spectra into d_X), one inverse row pass over
    // 2 * n_bins spectra per satellite and one column pass that adds the two |.|^2 and writes the grid once
    // (ACQ_EPI_MAG2) -- the grid read-modify-write of the second dwell and the first dwell's grid write never happen.  Anything
    // else (a fetch, a grid read, a change of codes or of the Doppler grid) first runs the held-back passes alone, so every caller
    // sees what per-dwell processing would have produced; gc_acq_reset() drops them with the grid.
    bool inv_pending = false;
    bool inv_accumulate = false;
    int inv_n_bins = 0;
    hipStream_t inv_stream = nullptr;
    bool fuse_dwells = true;  // $GNSSCORR_ACQ_FUSE=0: every dwell on its own
    // Experiment, off by default ($GNSSCORR_ACQ_OVERLAP=1): row pass (instruction- and LDS-bound, 3.8 TB/s) and column pass (bandwidth-
    // bound, 5.6 TB/s) of DIFFERENT satellite batches side by side -- rows on the caller's stream, columns on a stream of the engine,
    // d_Q double-buffered, events both ways.  Measured 0.42 instead of 0.355 ms per search: the row pass holds all the LDS of every CU
    // (4 x 40 KB), so the column workgroups do not become resident beside it and the eight cross-queue hand-overs only add latency
    bool overlap = false;
    // Experiment, off by default ($GNSSCORR_ACQ_ROLES=1): with more than one satellite batch of dwell pairs, the row pass of batch b and
    // the column pass of batch b - 1 share ONE launch (acq_rows3_cols_kernel: of every four workgroups of a CU, 4 - ACQ_ROLE_COLS do rows
    // and ACQ_ROLE_COLS columns), d_Q double-buffered, no events.  Measured 0.44 (one column workgroup of four) and 0.33 (two) against
    // 0.31 ms per search: a column workgroup needs its CU-mates' loads in flight to hide its two load phases, and the row pass slows down
    // with fewer than three workgroups
    bool roles = false;
    // Experiment, off by default ($GNSSCORR_ACQ_ONCHIP=1, experiments build): N = 25 x 1000 plans run the inverse transform of a cell
    // entirely on its CU (acq_inv_fused_kernel, no inter-pass buffer).  Measured 0.525 ms per cfg4 search against 0.311 ms for the
    // two-pass form: DESIGN.md section 3.2
    bool onchip = false;
    hipStream_t side = nullptr;
    hipEvent_t ev_rows[2] = {nullptr, nullptr}, ev_cols[2] = {nullptr, nullptr};
    size_t q_stride = 0;      // float2 elements between the two halves of d_Q (0: single buffer)
};

static hipError_t acq_flush_inverse(gc_acq* a, hipStream_t st);

// runs the pending statistics kernel, if any, on `st`
static hipError_t acq_flush_final(gc_acq* a, hipStream_t st)
{
    if (!a->final_pending) return hipSuccess;
    a->final_pending = false;
    // the dwell it belongs to was enqueued on final_stream: same stream in every sensible use; otherwise order the two
    if (a->final_stream != st)
        {
            hipEvent_t ev;
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(ev, a->final_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, ev, 0);
            if (e == hipSuccess) (void)hipEventDestroy(ev);
            if (e != hipSuccess) return e;
        }
    return acq_launch_final(st, a->final_args, a->n_sats);
}

static void acq_release(gc_acq* a)
{
    (void)hipFree(a->d_wN);
    (void)hipFree(a->d_wN2);
    (void)hipFree(a->d_wipe_main);
    (void)hipFree(a->d_wipe2);
    (void)hipFree(a->d_segs);
    (void)hipFree(a->d_seg_off);
    (void)hipFree(a->d_codes);
    (void)hipFree(a->d_xw);
    (void)hipFree(a->d_X);
    (void)hipFree(a->d_Q);
    (void)hipFree(a->d_grid);
    (void)hipFree(a->d_tmp);
    (void)hipFree(a->d_blkv);
    (void)hipFree(a->d_blki);
    (void)hipFree(a->d_power);
    (void)hipFree(a->d_in);
    (void)hipFree(a->d_cvt);
    (void)hipFree(a->d_results);
    (void)hipFree(a->d_part_val);
    (void)hipFree(a->d_part_cnt);
    if (a->h_results) (void)hipHostFree(a->h_results);
    for (int i = 0; i < 2; i++)
        {
            if (a->ev_rows[i]) (void)hipEventDestroy(a->ev_rows[i]);
            if (a->ev_cols[i]) (void)hipEventDestroy(a->ev_cols[i]);
        }
    if (a->side) (void)hipStreamDestroy(a->side);
}

// The wipe-off tables are kept in the row-permuted layout the forward row pass reads (P[bin][a][b] = wipe[bin][a + N1 * b]): the
// product x * wipeoff[bin] (pcps_acquisition.cc:717) is then the two-operand load of that pass -- A = the table, B = the permuted
// input block, shared by every bin -- instead of a kernel of its own that writes and re-reads n_bins x N products per dwell.
// A table is built out of place by ONE kernel in which every sample is independent: the float32 running phase of each row in closed form
// (acq_phase_segments.h, ~30 arithmetic-progression segments per row, computed here on the host; 25000 dependent additions on one
// lane per bin took 0.56 ms), (cos, sin) stored at the permuted position.  No scratch array, no device-to-device copy, no allocation
// in a set-up call unless a pathological increment needs more segments than the buffer holds.  The caller holds the context mutex.
static hipError_t acq_build_wipeoffs(gc_acq* a, const std::vector<float>& inc, float2* table, hipStream_t st)
{
    const int n = (int)inc.size();
    std::vector<AcqPhaseSeg> segs;
    std::vector<int> off(n + 1, 0);
    for (int d = 0; d < n; d++)
        {
            acq_phase_segments(inc[d], (int)a->fft_size, segs);
            off[d + 1] = (int)segs.size();
        }
    hipError_t e = hipSuccess;
    if (segs.size() > a->seg_cap)
        {
            e = hipStreamSynchronize(st);
            (void)hipFree(a->d_segs);
            a->d_segs = nullptr;
            a->seg_cap = 0;
            if (e == hipSuccess) e = hipMalloc(&a->d_segs, segs.size() * sizeof(AcqPhaseSeg));
            if (e != hipSuccess) return e;
            a->seg_cap = segs.size();
        }
    e = hipMemcpyAsync(a->d_segs, segs.data(), sizeof(AcqPhaseSeg) * segs.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(a->d_seg_off, off.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = acq_launch_wipeoff_segments(st, a->d_segs, a->d_seg_off, table, n, a->plan);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // the host vectors go out of scope
    return e;
}

// d_grid_doppler_wipeoffs of the coarse grid: exp(-j*2*pi*(d_old_freq + doppler)/fs * n) with the reference's float32
// running phase (update_grid_doppler_wipeoffs :371-380, update_local_carrier :296-310); d_old_freq = freq_offset_hz
static hipError_t acq_build_main_wipeoffs(gc_acq* a, hipStream_t st)
{
    std::vector<float> inc(a->n_bins_main);
    for (uint32_t d = 0; d < a->n_bins_main; d++)
        {
            const int32_t doppler = -(int32_t)a->conf.doppler_max + (int32_t)a->conf.doppler_step * (int32_t)d;
            const float freq = (float)(a->freq_offset_hz + (int64_t)doppler);
            const float phase_step_rad = (float)(6.283185307179586 * freq / (float)a->conf.fs_in);
            inc[d] = -phase_step_rad;
        }
    a->wipe_valid = false;
    const hipError_t e = acq_build_wipeoffs(a, inc, a->d_wipe_main, st);
    a->wipe_valid = (e == hipSuccess);
    return e;
}

#define ACQ_TRY(call)                                                                                  \
    do                                                                                                 \
        {                                                                                              \
            hipError_t e_ = (call);                                                                    \
            if (e_ != hipSuccess)                                                                      \
                {                                                                                      \
                    acq_release(a);                                                                    \
                    delete a;                                                                          \
                    return gc_fail(GC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));         \
                }                                                                                      \
        }                                                                                              \
    while (0)

extern "C" {

gc_status gc_acq_create(gc_ctx* ctx, const gc_acq_conf* conf, int n_sats, gc_acq** out)
{
    GC_REQUIRE(ctx && conf && out, "gc_acq_create: NULL argument");
    *out = nullptr;
    GC_REQUIRE(n_sats > 0, "gc_acq_create: n_sats must be > 0");
    GC_REQUIRE(conf->sampled_ms > 0 && conf->samples_per_ms > 0.0f, "gc_acq_create: bad sizes");
    GC_REQUIRE(conf->doppler_step > 0, "gc_acq_create: doppler_step must be > 0");
    gc_device_guard g(ctx->device);
    gc_acq* a = new gc_acq();
    a->ctx = ctx;
    a->ctx_ref.bind(ctx);
    a->conf = *conf;
    a->n_sats = n_sats;
    // pcps_acquisition.cc:77-85, :113-117
    const bool bt = conf->bit_transition_flag != 0;
    a->consumed = (uint32_t)(conf->sampled_ms * conf->samples_per_ms * (bt ? 2 : 1));
    a->fft_size = (conf->sampled_ms == conf->ms_per_code) ? a->consumed : a->consumed * 2;
    a->max_dwells = conf->max_dwells;
    if (bt)
        {
            a->fft_size = a->consumed * 2;
            a->max_dwells = 1;
        }
    a->eff = bt ? a->fft_size / 2 : a->fft_size;
    // :152-159 CFAR statistic only for a single dwell
    a->use_cfar = (a->max_dwells == 1) ? (conf->use_CFAR_algorithm_flag != 0) : false;
    // :326
    a->n_bins = (uint32_t)std::ceil((double)((int32_t)conf->doppler_max - (int32_t)(-(int32_t)conf->doppler_max)) / (double)conf->doppler_step);
    if (conf->num_doppler_bins_override > 0) a->n_bins = conf->num_doppler_bins_override;
    a->n_bins_main = a->n_bins;
    a->n_bins_alloc = a->n_bins;
    if (conf->make_2_steps && conf->num_doppler_bins_step2 > a->n_bins_alloc) a->n_bins_alloc = conf->num_doppler_bins_step2;
    if (a->n_bins == 0 || a->fft_size == 0)
        {
            delete a;
            return gc_fail(GC_ERR_INVALID, "gc_acq_create: empty search grid");
        }
    const size_t lds_limit = 160 * 1024;
    if (!acq_plan_make(&a->plan, (int)a->fft_size, lds_limit))
        {
            delete a;
            return gc_fail(GC_ERR_INVALID, "gc_acq_create: fft_size %u has no supported factorisation", a->fft_size);
        }
    const size_t N = a->fft_size;
    a->n_blocks = acq_cols_blocks(a->plan);
    // scratch Q: a batch of satellites stays within ~140 MB (it lives in the 256 MB Infinity Cache between the two passes), and the
    // batches are equal: a launch's workgroups run in rounds of (CUs x workgroups per CU), so the time of a pass steps with the
    // round count -- 11 + 11 + 10 satellites x 41 bins cost 3 + 3 + 2 rounds of the row pass, 16 + 16 cost 4 + 4 with fewer, larger
    // launches (measured 0.53 -> 0.50 ms per search)
    size_t per_sat = (size_t)a->n_bins_alloc * N * sizeof(float2);
    size_t q_budget = (size_t)140 << 20;
    if (const char* e = std::getenv("GNSSCORR_ACQ_Q_MB")) q_budget = (size_t)std::max(1, std::atoi(e)) << 20;
    a->sats_per_batch = (int)(q_budget / per_sat);
    if (a->sats_per_batch < 1) a->sats_per_batch = 1;
    if (a->sats_per_batch > n_sats) a->sats_per_batch = n_sats;
    {
        const int n_batches = (n_sats + a->sats_per_batch - 1) / a->sats_per_batch;
        a->sats_per_batch = (n_sats + n_batches - 1) / n_batches;
    }
    size_t q_cells = (size_t)a->sats_per_batch * a->n_bins_alloc;

    ACQ_TRY(hipMalloc(&a->d_wN, N * sizeof(float2)));
    ACQ_TRY(hipMalloc(&a->d_wN2, (size_t)2 * a->plan.N2 * sizeof(float2)));
    ACQ_TRY(hipMalloc(&a->d_wipe_main, (size_t)a->n_bins_main * N * sizeof(float2)));
    a->d_wipe = a->d_wipe_main;
    if (conf->make_2_steps && conf->num_doppler_bins_step2 > 0)
        ACQ_TRY(hipMalloc(&a->d_wipe2, (size_t)conf->num_doppler_bins_step2 * N * sizeof(float2)));
    ACQ_TRY(hipMalloc(&a->d_codes, (size_t)n_sats * N * sizeof(float2)));
    ACQ_TRY(hipMalloc(&a->d_xw, (size_t)a->n_bins_alloc * N * sizeof(float2)));
    ACQ_TRY(hipMalloc(&a->d_X, (size_t)2 * a->n_bins_alloc * N * sizeof(float2)));  // two dwells' spectra (see inv_pending)
    if (const char* e = gc_exp_env("GNSSCORR_ACQ_FUSE")) a->fuse_dwells = std::atoi(e) != 0;
    if (const char* e = gc_exp_env("GNSSCORR_ACQ_OVERLAP")) a->overlap = std::atoi(e) != 0;
    if (const char* e = gc_exp_env("GNSSCORR_ACQ_ROLES")) a->roles = std::atoi(e) != 0;
    if (const char* e = gc_exp_env("GNSSCORR_ACQ_ONCHIP")) a->onchip = std::atoi(e) != 0;
#ifdef GNSSCORR_EXPERIMENTS
    a->roles = a->roles && a->fuse_dwells && acq_rows_cols_fusable(a->plan);
#else
    a->roles = a->overlap = a->onchip = false;
#endif
    if (a->roles && n_sats > 1) a->q_stride = q_cells * N;  // second inter-pass buffer
    if (a->overlap && a->sats_per_batch < n_sats)
        {
            // more than one batch: a second buffer, so that the rows of batch b + 1 run beside the columns of batch b
            a->q_stride = q_cells * N;
            ACQ_TRY(hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking));
            for (int i = 0; i < 2; i++)
                {
                    ACQ_TRY(hipEventCreateWithFlags(&a->ev_rows[i], hipEventDisableTiming));
                    ACQ_TRY(hipEventCreateWithFlags(&a->ev_cols[i], hipEventDisableTiming));
                }
        }
    ACQ_TRY(hipMalloc(&a->d_Q, (a->q_stride ? 2 : 1) * q_cells * N * sizeof(float2)));
    a->seg_cap = (size_t)a->n_bins_alloc * 128;
    ACQ_TRY(hipMalloc(&a->d_segs, a->seg_cap * sizeof(AcqPhaseSeg)));
    ACQ_TRY(hipMalloc(&a->d_seg_off, ((size_t)a->n_bins_alloc + 1) * sizeof(int)));
    ACQ_TRY(hipMalloc(&a->d_grid, (size_t)n_sats * a->n_bins_alloc * N * sizeof(float)));
    ACQ_TRY(hipMalloc(&a->d_tmp, (size_t)n_sats * N * sizeof(float)));
    ACQ_TRY(hipMalloc(&a->d_blkv, (size_t)n_sats * a->n_bins_alloc * a->n_blocks * sizeof(float)));
    ACQ_TRY(hipMalloc(&a->d_blki, (size_t)n_sats * a->n_bins_alloc * a->n_blocks * sizeof(unsigned)));
    ACQ_TRY(hipMalloc(&a->d_power, sizeof(float)));
    ACQ_TRY(hipMalloc(&a->d_in, N * sizeof(float2)));
    ACQ_TRY(hipMalloc(&a->d_cvt, N * sizeof(float2)));
    ACQ_TRY(hipMalloc(&a->d_results, (size_t)n_sats * sizeof(gc_acq_result)));
    ACQ_TRY(hipMalloc(&a->d_part_val, (size_t)n_sats * ACQ_FINAL_PIECES * 2 * sizeof(float)));
    ACQ_TRY(hipMalloc(&a->d_part_cnt, (size_t)n_sats * sizeof(unsigned)));
    ACQ_TRY(hipMemset(a->d_part_cnt, 0, (size_t)n_sats * sizeof(unsigned)));
    ACQ_TRY(hipHostMalloc(reinterpret_cast<void**>(&a->h_results), (size_t)n_sats * sizeof(gc_acq_result), hipHostMallocDefault));
    a->code_set.assign(n_sats, 0);

    hipStream_t st = ctx->stream;
    // twiddle tables, rounded from float64
    {
        std::vector<float2> w(N);
        // inter-pass twiddles as the matrix T[k1][n2] = exp(-2*pi*j*k1*n2/N): unit-stride reads per row
        for (size_t k1 = 0; k1 < (size_t)a->plan.N1; k1++)
            for (size_t n2 = 0; n2 < (size_t)a->plan.N2; n2++)
                {
                    double ang = -2.0 * M_PI * (double)((k1 * n2) % N) / (double)N;
                    w[k1 * a->plan.N2 + n2] = make_float2((float)std::cos(ang), (float)std::sin(ang));
                }
        ACQ_TRY(hipMemcpy(a->d_wN, w.data(), N * sizeof(float2), hipMemcpyHostToDevice));
        // row-FFT twiddles: per-stage tables + plain table (acq_stage_twiddles)
        std::vector<float2> w2((size_t)2 * a->plan.N2);
        acq_stage_twiddles(a->plan, w2.data());
        ACQ_TRY(hipMemcpy(a->d_wN2, w2.data(), w2.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    // Doppler wipe-off grid: init() (:340-357) + update_local_carrier (:296-310)
    {
        hipError_t e = acq_build_main_wipeoffs(a, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        ACQ_TRY(e);
    }
    ACQ_TRY(hipMemsetAsync(a->d_grid, 0, (size_t)n_sats * a->n_bins_alloc * N * sizeof(float), st));
    ACQ_TRY(hipMemsetAsync(a->d_tmp, 0, (size_t)n_sats * N * sizeof(float), st));
    ACQ_TRY(hipMemsetAsync(a->d_codes, 0, (size_t)n_sats * N * sizeof(float2), st));
    ACQ_TRY(hipMemsetAsync(a->d_power, 0, sizeof(float), st));
    ACQ_TRY(hipStreamSynchronize(st));
    *out = a;
    return GC_OK;
}

gc_status gc_acq_destroy(gc_acq* a)
{
    if (!a) return GC_OK;
    gc_device_guard g(a->ctx->device);
    (void)hipStreamSynchronize(a->ctx->stream);
    if (a->side) (void)hipStreamSynchronize(a->side);
    acq_release(a);
    delete a;
    return GC_OK;
}

gc_status gc_acq_fft_size(const gc_acq* a, uint32_t* fft_size, uint32_t* consumed_samples, uint32_t* num_doppler_bins)
{
    GC_REQUIRE(a, "gc_acq_fft_size: NULL handle");
    if (fft_size) *fft_size = a->fft_size;
    if (consumed_samples) *consumed_samples = a->consumed;
    if (num_doppler_bins) *num_doppler_bins = a->n_bins;
    return GC_OK;
}

gc_status gc_acq_set_local_code(gc_acq* a, int sat, const float* code)
{
    GC_REQUIRE(a && code, "gc_acq_set_local_code: NULL argument");
    GC_REQUIRE(sat >= 0 && sat < a->n_sats, "gc_acq_set_local_code: satellite slot %d out of range", sat);
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    hipStream_t st = a->ctx->stream;
    const size_t N = a->fft_size;
    GC_HIP(acq_flush_inverse(a, st));  // a held-back dwell was searched with the codes of its time (and d_Q is the staging buffer below)
    // [0 .. 0 c_0 .. c_L] layouts of set_local_code (:252-269)
    std::vector<float2> buf(N, make_float2(0.f, 0.f));
    const float2* c = reinterpret_cast<const float2*>(code);
    if (a->conf.bit_transition_flag)
        {
            size_t off = N / 2;
            std::memcpy(buf.data() + off, c, sizeof(float2) * off);
        }
    else if (a->fft_size == a->consumed)
        std::memcpy(buf.data(), c, sizeof(float2) * a->consumed);
    else
        std::memcpy(buf.data() + (N - a->consumed), c, sizeof(float2) * a->consumed);
    GC_HIP(hipMemcpyAsync(a->d_in, buf.data(), N * sizeof(float2), hipMemcpyHostToDevice, st));
    // FFT, conjugate (:272-273), kept in the row-permuted layout the inverse rows pass reads
    hipError_t e = acq_launch_permute(st, a->d_in, nullptr, a->d_xw, a->plan, (int)N, 1, 0, 0, 0);
    if (e == hipSuccess) e = acq_launch_rows(st, false, a->plan, 1, a->d_xw, AcqCellMap{1, 1}, nullptr, AcqCellMap{1, 1}, a->d_Q, a->d_wN2, a->d_wN);
    if (e == hipSuccess) e = acq_launch_cols(st, false, ACQ_EPI_COMPLEX_CONJ_PERM, a->plan, 1, a->d_Q, a->d_codes + (size_t)sat * N, nullptr);
    if (e != hipSuccess) return gc_fail(GC_ERR_HIP, "gc_acq_set_local_code: %s", hipGetErrorString(e));
    GC_HIP(hipStreamSynchronize(st));  // buf goes out of scope
    a->code_set[sat] = 1;
    return GC_OK;
}

gc_status gc_acq_set_frequency_offset(gc_acq* a, int64_t offset_hz)
{
    GC_REQUIRE(a, "gc_acq_set_frequency_offset: NULL handle");
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    if (offset_hz == a->freq_offset_hz && a->wipe_valid) return GC_OK;
    GC_HIP(acq_flush_inverse(a, a->ctx->stream));  // a held-back dwell was wiped off with the tables of its time
    GC_HIP(hipStreamSynchronize(a->ctx->stream));
    a->freq_offset_hz = offset_hz;
    // a failed rebuild leaves wipe_valid clear: searches are refused until a later call (with any offset) has rebuilt the tables
    const hipError_t e = acq_build_main_wipeoffs(a, a->ctx->stream);
    if (e != hipSuccess) return gc_fail(GC_ERR_HIP, "gc_acq_set_frequency_offset: %s", hipGetErrorString(e));
    return GC_OK;
}

gc_status gc_acq_reset(gc_acq* a)
{
    GC_REQUIRE(a, "gc_acq_reset: NULL handle");
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    // grid reset (:917-924) and d_num_noncoherent_integrations_counter = 0.  The first dwell after a reset
    // STORES |.|^2 into every kept cell (accumulate = 0), so nothing has to be written here: a 131 MB memset per
    // search at cfg4 sizes, and it would have to be ordered against dwells enqueued on caller streams.
    a->dwell_counter = 0;
    a->grid_logically_zero = true;
    a->inv_pending = false;  // a held-back dwell goes with the grid it would have been added to
    return GC_OK;
}

gc_status gc_acq_set_step_two(gc_acq* a, int enable, float doppler_center_hz)
{
    GC_REQUIRE(a, "gc_acq_set_step_two: NULL handle");
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    hipStream_t st = a->ctx->stream;
    GC_HIP(acq_flush_inverse(a, st));  // a held-back dwell belongs to the grid that is active now
    if (!enable)
        {
            a->step_two = false;
            a->n_bins = a->n_bins_main;
            a->d_wipe = a->d_wipe_main;
            a->dwell_counter = 0;
            return GC_OK;
        }
    if (!a->conf.make_2_steps || !a->d_wipe2) return gc_fail(GC_ERR_STATE, "gc_acq_set_step_two: the plan was created without make_2_steps");
    const uint32_t n2 = a->conf.num_doppler_bins_step2;
    // update_grid_doppler_wipeoffs_step2 (pcps_acquisition.cc:383-390) + update_local_carrier (:296-310)
    std::vector<float> inc(n2);
    for (uint32_t d = 0; d < n2; d++)
        {
            float doppler = (static_cast<float>(d) - static_cast<float>(std::floor(n2 / 2.0))) * a->conf.doppler_step2;
            float freq = doppler_center_hz + doppler;
            float phase_step_rad = (float)(6.283185307179586 * freq / (float)a->conf.fs_in);
            inc[d] = -phase_step_rad;
        }
    a->wipe2_valid = false;
    const hipError_t e = acq_build_wipeoffs(a, inc, a->d_wipe2, st);
    if (e != hipSuccess) return gc_fail(GC_ERR_HIP, "gc_acq_set_step_two: %s", hipGetErrorString(e));
    a->wipe2_valid = true;
    a->step_two = true;
    a->center_step_two = doppler_center_hz;
    a->n_bins = n2;
    a->d_wipe = a->d_wipe2;
    a->dwell_counter = 0;  // d_num_noncoherent_integrations_counter = 0 (:848)
    return GC_OK;
}

// per satellite and bin: * conj(FFT(code)) (pcps_acquisition.cc:724), IFFT (:727), |.|^2 (+=) (:730-739) of the spectra in d_X, in
// batches of satellites; `pair`: d_X holds two dwells (slot 0, then slot 1 at n_bins spectra), summed into the grid in one pass
static hipError_t acq_inverse(gc_acq* a, hipStream_t st, bool pair, bool accumulate)
{
    const size_t N = a->fft_size;
    const int n_bins = (int)a->n_bins;
    const bool bt = a->conf.bit_transition_flag != 0;
    const int spectra = pair ? 2 * n_bins : n_bins;  // per satellite
    const int q_cells = a->sats_per_batch * (int)a->n_bins_alloc;
    int per_batch = q_cells / spectra;
    {
        const int n_batches = (a->n_sats + per_batch - 1) / per_batch;
        per_batch = (a->n_sats + n_batches - 1) / n_batches;  // equal batches
    }
    hipError_t e = hipSuccess;
    auto mag_args = [&](int s0) {
        AcqMagArgs m;
        m.grid = a->d_grid + (size_t)s0 * n_bins * N;
        m.tmp = a->d_tmp + (size_t)s0 * N;
        m.blk_max_val = a->d_blkv + (size_t)s0 * n_bins * a->n_blocks;
        m.blk_max_idx = a->d_blki + (size_t)s0 * n_bins * a->n_blocks;
        m.offset = bt ? (int)a->eff : 0;
        m.eff = (int)a->eff;
        m.n_bins = n_bins;
        m.tmp_bin = n_bins - 1;
        return m;
    };
#ifdef GNSSCORR_EXPERIMENTS
    if (a->onchip && acq_inv_fusable(a->plan))
        {
            // N = 25 x 1000: the whole inverse transform of every cell in ONE launch that keeps a cell's values on the CU (no
            // inter-pass buffer, no satellite batches); a pair's two transforms run back to back on the cell's workgroup
            const AcqMagArgs m = mag_args(0);
            e = acq_launch_inv_fused(st, a->plan, a->n_sats, n_bins, pair ? 2 : 1, accumulate, a->d_X, a->d_codes, a->d_wN2, a->d_wN, m, a->ctx->n_cus);
        }
    else if (pair && a->roles && a->q_stride != 0 && per_batch < a->n_sats && !a->overlap)
        {
            // rows(0); rows(b) + columns(b - 1) in one launch, b = 1 ..; columns(last)
            const int epi = accumulate ? ACQ_EPI_MAG2_ACC : ACQ_EPI_MAG2;
            int b = 0, prev_s0 = 0, prev_ns = 0;
            for (int s0 = 0; s0 < a->n_sats && e == hipSuccess; s0 += per_batch, b++)
                {
                    const int ns = std::min(per_batch, a->n_sats - s0);
                    float2* Q = a->d_Q + (size_t)(b & 1) * a->q_stride;
                    if (b == 0)
                        e = acq_launch_rows(st, true, a->plan, ns * spectra, a->d_X, AcqCellMap{1, spectra}, a->d_codes + (size_t)s0 * N,
                            AcqCellMap{spectra, 1 << 30}, Q, a->d_wN2, a->d_wN);
                    else
                        e = acq_launch_rows_cols(st, a->plan, ns * spectra, a->d_X, AcqCellMap{1, spectra}, a->d_codes + (size_t)s0 * N,
                            AcqCellMap{spectra, 1 << 30}, Q, a->d_wN2, a->d_wN, epi, prev_ns * n_bins, a->d_Q + (size_t)((b - 1) & 1) * a->q_stride,
                            mag_args(prev_s0));
                    prev_s0 = s0;
                    prev_ns = ns;
                }
            if (e == hipSuccess)
                {
                    const AcqMagArgs m = mag_args(prev_s0);
                    e = acq_launch_cols(st, true, epi, a->plan, prev_ns * n_bins, a->d_Q + (size_t)((b - 1) & 1) * a->q_stride, nullptr, &m);
                }
        }
    else
#endif
    {
    const bool two_streams = a->overlap && a->q_stride != 0 && per_batch < a->n_sats;
    int b = 0;
    for (int s0 = 0; s0 < a->n_sats && e == hipSuccess; s0 += per_batch, b++)
        {
            const int ns = std::min(per_batch, a->n_sats - s0);
            float2* Q = a->d_Q + (two_streams ? (size_t)(b & 1) * a->q_stride : 0);
            hipStream_t cst = two_streams ? a->side : st;
            if (two_streams && b >= 2) e = hipStreamWaitEvent(st, a->ev_cols[b & 1], 0);  // the columns of batch b - 2 have read this half
            if (e == hipSuccess)
                e = acq_launch_rows(st, true, a->plan, ns * spectra, a->d_X, AcqCellMap{1, spectra}, a->d_codes + (size_t)s0 * N,
                    AcqCellMap{spectra, 1 << 30}, Q, a->d_wN2, a->d_wN);
            if (e == hipSuccess && two_streams) e = hipEventRecord(a->ev_rows[b & 1], st);
            if (e == hipSuccess && two_streams) e = hipStreamWaitEvent(cst, a->ev_rows[b & 1], 0);
            if (e != hipSuccess) break;
            const AcqMagArgs m = mag_args(s0);
            e = acq_launch_cols(cst, true, pair ? (accumulate ? ACQ_EPI_MAG2_ACC : ACQ_EPI_MAG2) : (accumulate ? ACQ_EPI_MAG_ACC : ACQ_EPI_MAG), a->plan, ns * n_bins, Q, nullptr, &m);
            if (e == hipSuccess && two_streams) e = hipEventRecord(a->ev_cols[b & 1], cst);
        }
    // everything behind this call on `st` sees the finished grid (the side stream runs its column passes in order: the last one's event covers all)
    if (e == hipSuccess && two_streams && b > 0) e = hipStreamWaitEvent(st, a->ev_cols[(b - 1) & 1], 0);
    }
    if (e == hipSuccess)
        {
            AcqFinalArgs f;
            f.grid = a->d_grid;
            f.tmp = a->d_tmp;
            f.blk_max_val = a->d_blkv;
            f.blk_max_idx = a->d_blki;
            f.input_power = (a->use_cfar || bt) ? a->d_power : nullptr;
            f.results = a->d_results;
            f.part_val = a->d_part_val;
            f.part_cnt = a->d_part_cnt;
            f.n_bins = n_bins;
            f.n_blocks = a->n_blocks;
            f.fft_size = (int)N;
            f.doppler_max = (int)a->conf.doppler_max;
            f.doppler_step = (int)a->conf.doppler_step;
            f.use_cfar = a->use_cfar ? 1 : 0;
            f.samples_per_chip = (int)a->conf.samples_per_chip;
            f.samples_per_code = a->conf.samples_per_code;
            f.step_two = a->step_two ? 1 : 0;
            f.center_step_two = a->center_step_two;
            f.doppler_step2 = a->conf.doppler_step2;
            f.n_bins_step2 = (int)a->conf.num_doppler_bins_step2;
            a->final_args = f;
            a->final_stream = st;
            a->final_pending = true;
        }
    return e;
}

// FFT(x * wipeoff[d]) of `n_blocks` input blocks parked (row-permuted) in d_xw -> d_X[block][bin]: x * wipeoff[d]
// (pcps_acquisition.cc:717) is the two-operand load of the forward row pass (:721), shared by every satellite
static hipError_t acq_forward(gc_acq* a, hipStream_t st, int n_blocks)
{
    const int n_bins = (int)a->n_bins;
    hipError_t e = acq_launch_rows(st, false, a->plan, n_blocks * n_bins, a->d_wipe, AcqCellMap{1, n_bins}, a->d_xw, AcqCellMap{n_bins, 1 << 30},
        a->d_Q, a->d_wN2, a->d_wN);
    if (e == hipSuccess) e = acq_launch_cols(st, false, ACQ_EPI_PERM, a->plan, n_blocks * n_bins, a->d_Q, a->d_X, nullptr);
    return e;
}

// makes `st` wait for what has been enqueued on `other` so far
static hipError_t acq_order_after(hipStream_t st, hipStream_t other)
{
    if (st == other) return hipSuccess;
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, other);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, ev, 0);
    if (e == hipSuccess) (void)hipEventDestroy(ev);
    return e;
}

// runs the held-back inverse passes of the last dwell, if any, on `st` (its statistics become the pending ones)
static hipError_t acq_flush_inverse(gc_acq* a, hipStream_t st)
{
    if (!a->inv_pending) return hipSuccess;
    a->inv_pending = false;
    hipError_t e = acq_order_after(st, a->inv_stream);
    // statistics of an earlier dwell still pending: the dwell held back here accumulates onto that grid (or it would have flushed them)
    if (e == hipSuccess && a->final_pending)
        {
            const bool reproduces = a->inv_accumulate && a->final_args.n_bins == a->inv_n_bins && a->final_stream == st;
            if (reproduces)
                a->final_pending = false;
            else
                e = acq_flush_final(a, st);
        }
    if (e == hipSuccess) e = acq_forward(a, st, 1);
    if (e == hipSuccess) e = acq_inverse(a, st, false, a->inv_accumulate);
    return e;
}

// one dwell of every satellite slot on `st`; the caller holds the context mutex
static gc_status acq_enqueue(gc_acq* a, const void* dev_iq_in, int iq_format, hipStream_t st)
{
    for (int s = 0; s < a->n_sats; s++)
        if (!a->code_set[s]) return gc_fail(GC_ERR_STATE, "gc_acq_dwell: satellite slot %d has no local code", s);
    if (!(a->step_two ? a->wipe2_valid : a->wipe_valid))
        return gc_fail(GC_ERR_STATE, "gc_acq_dwell: the Doppler wipe-off tables are not valid (a rebuild failed): call gc_acq_set_frequency_offset / gc_acq_set_step_two again");
    const float2* dev_iq = static_cast<const float2*>(dev_iq_in);
    if (iq_format != GC_IQ_F32)
        {
            // d_cshort path of acquisition_core (:676-679): convert the block, then the float search
            hipError_t ec = acq_launch_convert(st, iq_format, dev_iq_in, a->d_cvt, (int)a->consumed);
            if (ec != hipSuccess) return gc_fail(GC_ERR_HIP, "gc_acq_dwell: input conversion failed: %s", hipGetErrorString(ec));
            dev_iq = a->d_cvt;
        }
    const size_t N = a->fft_size;
    const int n_bins = (int)a->n_bins;
    a->dwell_counter++;
    hipError_t e = hipSuccess;
    const bool bt = a->conf.bit_transition_flag != 0;
    const bool accumulate = a->dwell_counter > 1;
    const bool plain = !a->use_cfar && !bt;
    // a dwell is held back: this one joins it if it adds to the same grid, otherwise the held-back passes run first
    bool pair = false;
    if (a->inv_pending)
        {
            pair = accumulate && plain && a->inv_n_bins == n_bins && a->inv_stream == st;
            if (!pair) e = acq_flush_inverse(a, st);
        }
    if (e == hipSuccess && a->final_pending)
        {
            // an accumulating dwell on the same grid overwrites the scratch the pending evaluation would have left: drop it
            const bool reproduces = accumulate && plain && a->final_args.n_bins == n_bins && a->final_stream == st;
            if (reproduces)
                a->final_pending = false;
            else
                e = acq_flush_final(a, st);
        }
    if (e == hipSuccess && (a->use_cfar || bt))
        e = acq_launch_input_power(st, dev_iq, (int)a->consumed, (int)N, a->d_power, a->d_tmp, a->n_sats, N);
    // the input block, zero padded to fft_size (:680-688), row-permuted once: the caller's buffer is free when this has run.  The
    // second block of a pair goes behind the first one's
    if (e == hipSuccess) e = acq_launch_permute(st, dev_iq, nullptr, a->d_xw + (pair ? N : 0), a->plan, (int)a->consumed, 1, 0, 0, 0);
    if (e == hipSuccess)
        {
            const int q_cells = a->sats_per_batch * (int)a->n_bins_alloc;
            if (pair)
                {
                    // both blocks' forward transforms in one pair of launches, then both dwells' inverse passes
                    a->inv_pending = false;
                    e = acq_forward(a, st, 2);
                    if (e == hipSuccess) e = acq_inverse(a, st, true, a->inv_accumulate);
                }
            else if (a->fuse_dwells && plain && a->dwell_counter < a->max_dwells && q_cells >= 2 * n_bins && a->n_bins_alloc >= 2)
                {
                    // more dwells of this search are expected: hold everything behind the input permutation back (see gc_acq::inv_pending)
                    a->inv_pending = true;
                    a->inv_accumulate = accumulate;
                    a->inv_n_bins = n_bins;
                    a->inv_stream = st;
                }
            else
                {
                    e = acq_forward(a, st, 1);
                    if (e == hipSuccess) e = acq_inverse(a, st, false, accumulate);
                }
        }
    if (e != hipSuccess) return gc_fail(GC_ERR_HIP, "gc_acq_dwell: kernel launch failed: %s", hipGetErrorString(e));
    a->grid_logically_zero = false;
    return GC_OK;
}

gc_status gc_acq_set_input_format(gc_acq* a, int iq_format)
{
    GC_REQUIRE(a, "gc_acq_set_input_format: NULL handle");
    GC_REQUIRE(iq_format == GC_IQ_F32 || iq_format == GC_IQ_I16 || iq_format == GC_IQ_I8, "gc_acq_set_input_format: unknown format %d", iq_format);
    a->iq_format = iq_format;
    return GC_OK;
}

gc_status gc_acq_dwell_enqueue(gc_acq* a, const void* dev_iq, void* stream)
{
    GC_REQUIRE(a && dev_iq, "gc_acq_dwell_enqueue: NULL argument");
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    return acq_enqueue(a, dev_iq, a->iq_format, gc_pick_stream(a->ctx, stream));
}

// results of the last dwell to the host; the caller holds the context mutex
static gc_status acq_fetch(gc_acq* a, gc_acq_result* host_results, hipStream_t st)
{
    GC_HIP(acq_flush_inverse(a, st));
    GC_HIP(acq_flush_final(a, st));
    GC_HIP(hipMemcpyAsync(a->h_results, a->d_results, sizeof(gc_acq_result) * a->n_sats, hipMemcpyDeviceToHost, st));
    GC_HIP(hipStreamSynchronize(st));
    std::memcpy(host_results, a->h_results, sizeof(gc_acq_result) * a->n_sats);
    return GC_OK;
}

gc_status gc_acq_fetch_results(gc_acq* a, gc_acq_result* host_results, void* stream)
{
    GC_REQUIRE(a && host_results, "gc_acq_fetch_results: NULL argument");
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    return acq_fetch(a, host_results, gc_pick_stream(a->ctx, stream));
}

gc_status gc_acq_flush(gc_acq* a, void* stream)
{
    GC_REQUIRE(a, "gc_acq_flush: NULL handle");
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    hipStream_t st = gc_pick_stream(a->ctx, stream);
    GC_HIP(acq_flush_inverse(a, st));
    GC_HIP(acq_flush_final(a, st));
    return GC_OK;
}

gc_status gc_acq_dwell_dev(gc_acq* a, const void* dev_iq, gc_acq_result* host_results, void* stream)
{
    GC_REQUIRE(a && dev_iq && host_results, "gc_acq_dwell_dev: NULL argument");
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    hipStream_t st = gc_pick_stream(a->ctx, stream);
    gc_status s = acq_enqueue(a, dev_iq, a->iq_format, st);
    if (s != GC_OK) return s;
    return acq_fetch(a, host_results, st);
}

gc_status gc_acq_dwell(gc_acq* a, const float* host_iq, gc_acq_result* host_results)
{
    GC_REQUIRE(a && host_iq && host_results, "gc_acq_dwell: NULL argument");
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    hipStream_t st = a->ctx->stream;
    // the host entry point takes gr_complex, whatever the device-side format is; the copy, the search and
    // the read-back stay under one lock because d_in is also set_local_code's staging buffer
    GC_HIP(hipMemcpyAsync(a->d_in, host_iq, sizeof(float2) * a->consumed, hipMemcpyHostToDevice, st));
    gc_status s = acq_enqueue(a, a->d_in, GC_IQ_F32, st);
    if (s != GC_OK) return s;
    return acq_fetch(a, host_results, st);
}

gc_status gc_acq_dwell_stream(gc_acq* a, gc_stream* s, uint64_t first_index, gc_acq_result* host_results)
{
    GC_REQUIRE(a && s && host_results, "gc_acq_dwell_stream: NULL argument");
    GC_REQUIRE(s->ctx->device == a->ctx->device, "gc_acq_dwell_stream: the stream lives on another GPU");
    GC_REQUIRE(s->iq_format == a->iq_format, "gc_acq_dwell_stream: stream format %d, acquisition format %d (gc_acq_set_input_format)",
        s->iq_format, a->iq_format);
    GC_REQUIRE(a->consumed <= s->mirror, "gc_acq_dwell_stream: the block of %u samples is longer than the stream's max_window", a->consumed);
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    hipStream_t st = a->ctx->stream;
    // the block is protected against eviction from here (reserved before the residency check and the launch)
    gc_stream_ticket t;
    gc_status rs = gc_stream_begin_read(s, st, first_index, &t);
    if (rs != GC_OK) return rs;
    if (first_index + a->consumed > t.head)
        {
            gc_stream_cancel_read(s, t);
            return gc_fail(GC_ERR_INVALID, "gc_acq_dwell_stream: block [%llu, +%u) is not inside the stream's resident samples [%llu, %llu)",
                (unsigned long long)first_index, a->consumed, (unsigned long long)t.oldest, (unsigned long long)t.head);
        }
    rs = acq_enqueue(a, s->d_ring + (first_index % s->capacity) * s->elem, a->iq_format, st);
    if (rs != GC_OK)
        {
            gc_stream_cancel_read(s, t);
            return rs;
        }
    rs = gc_stream_end_read(s, st, t);
    if (rs != GC_OK) return rs;
    return acq_fetch(a, host_results, st);
}

gc_status gc_acq_get_grid(gc_acq* a, int sat, float* host_grid)
{
    GC_REQUIRE(a && host_grid, "gc_acq_get_grid: NULL argument");
    GC_REQUIRE(sat >= 0 && sat < a->n_sats, "gc_acq_get_grid: satellite slot %d out of range", sat);
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    const size_t n = (size_t)a->n_bins * a->fft_size;
    if (a->grid_logically_zero)
        {
            std::memset(host_grid, 0, n * sizeof(float));
            return GC_OK;
        }
    GC_HIP(acq_flush_inverse(a, a->ctx->stream));
    GC_HIP(hipStreamSynchronize(a->ctx->stream));
    GC_HIP(hipMemcpy(host_grid, a->d_grid + (size_t)sat * n, n * sizeof(float), hipMemcpyDeviceToHost));
    return GC_OK;
}

// row-permuted P[a][b] = v[a + N1 b]  ->  natural order
static void acq_unpermute(const AcqFftPlan& plan, const float2* p, float* out)
{
    for (int a = 0; a < plan.N1; a++)
        for (int b = 0; b < plan.N2; b++)
            {
                const float2 v = p[(size_t)a * plan.N2 + b];
                out[2 * ((size_t)a + (size_t)plan.N1 * b)] = v.x;
                out[2 * ((size_t)a + (size_t)plan.N1 * b) + 1] = v.y;
            }
}

gc_status gc_acq_peek(gc_acq* a, int what, int index, float* host_out)
{
    GC_REQUIRE(a && host_out, "gc_acq_peek: NULL argument");
    gc_device_guard g(a->ctx->device);
    std::lock_guard<std::mutex> lk(a->ctx->mtx);
    hipStream_t st = a->ctx->stream;
    const size_t N = a->fft_size;
    GC_HIP(acq_flush_inverse(a, st));
    GC_HIP(hipStreamSynchronize(st));
    if (what == GC_ACQ_PEEK_ROW_MAX)
        {
            GC_REQUIRE(index >= 0 && index < a->n_sats, "gc_acq_peek: satellite slot %d out of range", index);
            const size_t n = (size_t)a->n_bins * a->n_blocks;
            std::vector<float> v(n);
            std::vector<unsigned> ix(n);
            GC_HIP(hipMemcpy(v.data(), a->d_blkv + (size_t)index * n, n * sizeof(float), hipMemcpyDeviceToHost));
            GC_HIP(hipMemcpy(ix.data(), a->d_blki + (size_t)index * n, n * sizeof(unsigned), hipMemcpyDeviceToHost));
            for (uint32_t d = 0; d < a->n_bins; d++)
                {
                    float best = -1.0f;
                    unsigned bi = 0xffffffffu;
                    for (int b = 0; b < a->n_blocks; b++)
                        {
                            const size_t e = (size_t)d * a->n_blocks + b;
                            if (ix[e] == 0xffffffffu) continue;
                            if (v[e] > best || (v[e] == best && ix[e] < bi))
                                {
                                    best = v[e];
                                    bi = ix[e];
                                }
                        }
                    host_out[2 * d] = best;
                    host_out[2 * d + 1] = (float)bi;
                }
            return GC_OK;
        }
    const float2* src = nullptr;
    if (what == GC_ACQ_PEEK_WIPEOFF || what == GC_ACQ_PEEK_SPECTRUM)
        {
            GC_REQUIRE(index >= 0 && (uint32_t)index < a->n_bins, "gc_acq_peek: Doppler bin %d out of range", index);
            src = (what == GC_ACQ_PEEK_WIPEOFF ? a->d_wipe : a->d_X) + (size_t)index * N;
        }
    else if (what == GC_ACQ_PEEK_CODE)
        {
            GC_REQUIRE(index >= 0 && index < a->n_sats, "gc_acq_peek: satellite slot %d out of range", index);
            src = a->d_codes + (size_t)index * N;
        }
    else
        return gc_fail(GC_ERR_INVALID, "gc_acq_peek: unknown item %d", what);
    std::vector<float2> p(N);
    GC_HIP(hipMemcpy(p.data(), src, N * sizeof(float2), hipMemcpyDeviceToHost));
    acq_unpermute(a->plan, p.data(), host_out);  // all three live in the row-permuted layout
    return GC_OK;
}

}  // extern "C"
