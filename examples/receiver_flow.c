/*
 * receiver_flow.c -- the C ABI of libgnsscorr.so used from plain C99, end to end:
 *   one synthetic RF stream (GPS L1 C/A, 4 Msps, four satellites in noise) is pushed block by block into the
 *   HBM ring; a PCPS search of eight PRNs runs on its first 4 ms; every detected satellite is handed over to the
 *   closed-loop DLL/PLL engine, which tracks it on the same ring a few code periods per launch.
 * Build:  gcc -std=c99 -O2 -Iinclude examples/receiver_flow.c -Lgnss-sdr-1_amd -lgnsscorr -Wl,-rpath,$PWD/gnss-sdr-1_amd -lm -o receiver_flow
 * Run on a machine with an MI355X: ./receiver_flow   (exit code 0 = every present satellite acquired and locked)
 */
#include "gnsscorr.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FS 4000000
#define N_MS 4000 /* samples per code period */
#define N_PRESENT 4
#define N_SEARCH 8
#define TOTAL_MS 300

#define CHECK(call)                                                              \
    do                                                                           \
        {                                                                        \
            gc_status st_ = (call);                                              \
            if (st_ != GC_OK)                                                    \
                {                                                                \
                    fprintf(stderr, "%s -> %d: %s\n", #call, st_, gc_last_error()); \
                    return 2;                                                    \
                }                                                                \
        }                                                                        \
    while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static double uniform01(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}
static double gauss(void)
{
    double u = uniform01(), v = uniform01();
    if (u < 1e-300) u = 1e-300;
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}

int main(void)
{
    static const int present[N_PRESENT] = {3, 8, 14, 22};
    static const int search[N_SEARCH] = {3, 8, 14, 22, 5, 11, 19, 30};
    const double doppler[N_PRESENT] = {-2210.0, 1375.0, 3120.0, -640.0};
    const double delay[N_PRESENT] = {1234.0, 77.0, 3001.0, 2500.0}; /* code phase of the first sample, in samples */
    const double cn0[N_PRESENT] = {47.0, 49.0, 46.0, 50.0};
    const size_t total = (size_t)TOTAL_MS * N_MS;
    float* x = (float*)calloc(2 * total, sizeof(float));
    float code[N_PRESENT][1023];
    int s, ch;
    size_t i;
    if (GC_ABI_CHECK() != GC_OK)
        {
            printf("%s\n", gc_last_error()); /* this program was compiled against another gnsscorr.h than the library */
            return 2;
        }
    if (gc_device_count() == 0)
        {
            printf("no GPU: libgnsscorr has no CPU fallback\n");
            return 3;
        }
    /* ---- synthetic stream ---- */
    for (s = 0; s < N_PRESENT; s++)
        {
            const double amp = sqrt(pow(10.0, cn0[s] / 10.0) / FS);
            const double rate = 1.023e6 * (1.0 + doppler[s] / 1575.42e6) / FS;
            const double tau0 = 1023.0 - delay[s] * 1.023e6 / FS;
            CHECK(gc_gps_l1_ca_code_gen_float(code[s], present[s], 0));
            for (i = 0; i < total; i++)
                {
                    const long chip = (long)floor(tau0 + (double)i * rate) % 1023;
                    const double ph = 6.283185307179586 * doppler[s] * (double)i / FS + 0.3 * s;
                    x[2 * i] += (float)(amp * code[s][chip] * cos(ph));
                    x[2 * i + 1] += (float)(amp * code[s][chip] * sin(ph));
                }
        }
    for (i = 0; i < 2 * total; i++) x[i] += (float)(gauss() * sqrt(0.5));

    gc_ctx* ctx = NULL;
    gc_stream* ring = NULL;
    CHECK(gc_ctx_create(0, &ctx));
    CHECK(gc_stream_create(ctx, GC_IQ_F32, 40 * N_MS, 4 * N_MS, &ring));
    CHECK(gc_stream_push(ring, x, 4 * N_MS, NULL));

    /* ---- acquisition: eight PRNs, 4 ms coherent, 50 Hz bins, straight from the ring ---- */
    gc_acq_conf ac;
    memset(&ac, 0, sizeof ac);
    ac.fs_in = FS;
    ac.sampled_ms = 4;
    ac.ms_per_code = 1;
    ac.samples_per_ms = (float)FS * 0.001f;
    ac.samples_per_code = 4000.0f;
    ac.samples_per_chip = 4;
    ac.doppler_max = 5000;
    ac.doppler_step = 50;
    ac.max_dwells = 1;
    ac.use_CFAR_algorithm_flag = 1;
    gc_acq* acq = NULL;
    CHECK(gc_acq_create(ctx, &ac, N_SEARCH, &acq));
    {
        float* sampled = (float*)malloc(sizeof(float) * 2 * 4 * N_MS);
        for (s = 0; s < N_SEARCH; s++)
            {
                int32_t n = 0;
                int rep;
                CHECK(gc_gps_l1_ca_code_gen_complex_sampled(sampled, (uint32_t)search[s], FS, 0, &n));
                for (rep = 1; rep < 4; rep++) memcpy(sampled + 2 * rep * N_MS, sampled, sizeof(float) * 2 * N_MS); /* tiled like gps_l1_ca_pcps_acquisition.cc:239-243 */
                CHECK(gc_acq_set_local_code(acq, s, sampled));
            }
        free(sampled);
    }
    gc_acq_result res[N_SEARCH];
    CHECK(gc_acq_dwell_stream(acq, ring, 0, res));
    float noise_stat = 0.0f;
    for (s = N_PRESENT; s < N_SEARCH; s++) noise_stat += res[s].test_statistics / (N_SEARCH - N_PRESENT);
    int detected[N_SEARCH], n_det = 0;
    for (s = 0; s < N_SEARCH; s++)
        {
            const int hit = res[s].test_statistics > 2.0f * noise_stat;
            printf("PRN %2d: statistic %.5f delay %7.1f samples Doppler %6.0f Hz %s\n", search[s], res[s].test_statistics, res[s].acq_delay_samples,
                res[s].acq_doppler_hz, hit ? "<- acquired" : "");
            if (hit) detected[n_det++] = s;
        }
    CHECK(gc_acq_destroy(acq));
    if (n_det != N_PRESENT)
        {
            printf("expected %d detections, got %d\n", N_PRESENT, n_det);
            return 1;
        }

    /* ---- hand-over and closed-loop tracking on the same ring ---- */
    gc_trk_loop* loop = NULL;
    CHECK(gc_trk_loop_create(ctx, n_det, 1023, &loop));
    for (ch = 0; ch < n_det; ch++)
        {
            gc_loop_conf lc;
            memset(&lc, 0, sizeof lc);
            lc.fs_in = FS;
            lc.signal_carrier_freq_hz = 1575.42e6;
            lc.code_chip_rate_hz = 1.023e6;
            lc.code_period_s = 0.001;
            lc.carrier_lock_th = 0.85;
            lc.acq_delay_samples = res[detected[ch]].acq_delay_samples;
            lc.acq_doppler_hz = res[detected[ch]].acq_doppler_hz;
            lc.acq_samplestamp_samples = 0;
            lc.sample_counter = 0;
            lc.code_length_chips = 1023;
            lc.code_samples_per_chip = 1;
            lc.vector_length = N_MS;
            lc.pull_in_time_s = 2;
            lc.pll_filter_order = 3;
            lc.dll_filter_order = 2;
            lc.cn0_samples = 20;
            lc.cn0_min = 25;
            lc.max_lock_fail = 50;
            lc.pll_bw_hz = 40.0f;
            lc.dll_bw_hz = 2.0f;
            lc.fll_bw_hz = 35.0f;
            lc.early_late_space_chips = 0.5f;
            CHECK(gc_trk_loop_set_input_stream(loop, ch, ring));
            CHECK(gc_trk_loop_start(loop, ch, &lc, code[detected[ch]], 1023));
        }
    {
        enum { PER_LAUNCH = 6 };
        gc_loop_record* recs = (gc_loop_record*)malloc(sizeof(gc_loop_record) * (size_t)n_det * PER_LAUNCH);
        double last_doppler[N_PRESENT] = {0, 0, 0, 0}, last_lock[N_PRESENT] = {0, 0, 0, 0};
        long tracked[N_PRESENT] = {0, 0, 0, 0};
        int ms, ok = 1;
        for (ms = 4; ms + 5 <= TOTAL_MS; ms += 5)
            {
                int e;
                CHECK(gc_stream_push(ring, x + 2 * (size_t)ms * N_MS, 5 * N_MS, NULL));
                CHECK(gc_trk_loop_run(loop, PER_LAUNCH, recs));
                for (ch = 0; ch < n_det; ch++)
                    for (e = 0; e < PER_LAUNCH; e++)
                        {
                            const gc_loop_record* r = &recs[ch * PER_LAUNCH + e];
                            if (!r->valid) continue;
                            tracked[ch]++;
                            last_doppler[ch] = r->carrier_doppler_hz;
                            last_lock[ch] = r->carrier_lock_test;
                        }
            }
        for (ch = 0; ch < n_det; ch++)
            {
                const int good = fabs(last_doppler[ch] - doppler[detected[ch]]) < 15.0 && last_lock[ch] > 0.8 && tracked[ch] >= TOTAL_MS - 10;
                printf("PRN %2d: %ld code periods tracked, Doppler %.1f Hz (truth %.1f), lock detector %.3f %s\n", present[detected[ch]], tracked[ch],
                    last_doppler[ch], doppler[detected[ch]], last_lock[ch], good ? "locked" : "NOT LOCKED");
                ok = ok && good;
            }
        free(recs);
        CHECK(gc_trk_loop_destroy(loop));
        CHECK(gc_stream_destroy(ring));
        CHECK(gc_ctx_destroy(ctx));
        free(x);
        printf(ok ? "receiver flow ok\n" : "receiver flow FAILED\n");
        return ok ? 0 : 1;
    }
}
