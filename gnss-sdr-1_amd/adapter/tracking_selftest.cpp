// tracking_selftest.cpp -- closed-loop run of the tracking drop-in layer on the GPU (BASELINE configs[0]
// shape: GPS L1 C/A, one channel, 4 Msps, PCPS acquisition followed by 3-tap DLL/PLL tracking), plus
// Galileo E1 (5 taps, 4 ms) and BeiDou B1I hand-overs.  The input is a synthetic noiseless + noisy signal
// with known Doppler / code phase; the loop must pull in and stay locked, and its Doppler / C/N0 estimates
// must converge to the truth.  Usage: tracking_selftest (needs a GPU).
#include "dll_pll_tracking_adapters.h"
#include "pcps_acquisition_adapters.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

static int g_fail = 0;
#define EXPECT(cond, ...)                                            \
    do                                                               \
        {                                                            \
            if (!(cond))                                             \
                {                                                    \
                    std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
                    std::printf(__VA_ARGS__);                        \
                    std::printf("\n");                               \
                    g_fail++;                                        \
                }                                                    \
        }                                                            \
    while (0)

// x[n] = A c(tau0 + n*rate/fs) exp(j(2 pi fd n / fs + phi)) + w[n]
static std::vector<gr_complex> synth(const std::vector<float>& code, double chip_rate_hz, double carrier_hz, double fs, size_t n, double fd,
    double tau0_chips, double cn0_dbhz, unsigned seed)
{
    std::mt19937 gen(seed);
    std::normal_distribution<float> nd(0.0f, std::sqrt(0.5f));
    const double amp = std::sqrt(std::pow(10.0, cn0_dbhz / 10.0) / fs);
    const double rate = chip_rate_hz * (1.0 + fd / carrier_hz) / fs;
    const size_t L = code.size();
    std::vector<gr_complex> x(n);
    for (size_t i = 0; i < n; i++)
        {
            const double ph = 2.0 * M_PI * fd * static_cast<double>(i) / fs + 0.7;
            const size_t chip = static_cast<size_t>(std::floor(tau0_chips + static_cast<double>(i) * rate)) % L;
            x[i] = gr_complex(static_cast<float>(amp * code[chip] * std::cos(ph)) + nd(gen), static_cast<float>(amp * code[chip] * std::sin(ph)) + nd(gen));
        }
    return x;
}

template <class Trk>
static void run_tracking(Trk& trk, const std::vector<gr_complex>& x, Gnss_Synchro& syn, double true_doppler, double doppler_tol, double cn0_true, const char* name,
    int min_epochs)
{
    auto blk = trk.block();
    size_t pos = 0;
    int epochs = 0;
    Gnss_Synchro out;
    double last_doppler = 0, last_cn0 = 0;
    while (pos + blk->required_input_items() <= x.size())
        {
            int produced = 0;
            int used = blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
            pos += used;
            if (produced)
                {
                    epochs++;
                    last_doppler = out.Carrier_Doppler_hz;
                    last_cn0 = out.CN0_dB_hz;
                }
            if (blk->state() == 0) break;
        }
    EXPECT(blk->last_status() == GC_OK, "%s: engine status %d (%s)", name, blk->last_status(), gc_last_error());
    EXPECT(blk->state() == 2, "%s: lost lock (state %d after %d epochs)", name, blk->state(), epochs);
    EXPECT(blk->events().empty(), "%s: loss-of-lock event", name);
    EXPECT(epochs >= min_epochs, "%s: only %d epochs", name, epochs);
    EXPECT(std::fabs(last_doppler - true_doppler) < doppler_tol, "%s: Doppler %.2f Hz, truth %.2f", name, last_doppler, true_doppler);
    EXPECT(std::fabs(last_cn0 - cn0_true) < 3.0, "%s: C/N0 %.1f dB-Hz, truth %.1f", name, last_cn0, cn0_true);
    EXPECT(blk->carrier_lock_test() > 0.85, "%s: carrier lock test %.3f", name, blk->carrier_lock_test());
    std::printf("%s: %d epochs, Doppler %.2f Hz (truth %.2f), C/N0 %.1f dB-Hz (truth %.1f), lock test %.3f, code rate %.3f chips/s\n", name, epochs, last_doppler,
        true_doppler, last_cn0, cn0_true, blk->carrier_lock_test(), blk->code_freq_chips());
}

static void test_gps_acq_then_track()
{
    const double fs = 4e6, fd = 1680.0, cn0 = 45.0;
    std::vector<float> code(1023);
    gc_gps_l1_ca_code_gen_float(code.data(), 1, 0);
    // 524 samples of delay at 4 samples/chip...: chip phase at sample 0 such that the code starts at sample 524
    const double tau0 = 1023.0 - 524.0 * 1.023e6 / fs;
    auto x = synth(code, 1.023e6, 1575.42e6, fs, 4000 * 1200, fd, tau0, cn0, 11);  // 1.2 s: a 1 ms search leaves up to ~100 Hz for the PLL to pull in
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "4000000");
    config.set_property("Acquisition_1C.doppler_max", "5000");
    config.set_property("Acquisition_1C.doppler_step", "20");
    config.set_property("Tracking_1C.pll_bw_hz", "40.0");
    config.set_property("Tracking_1C.dll_bw_hz", "2.0");
    config.set_property("Tracking_1C.early_late_space_chips", "0.5");
    if (const char* dump_dir = std::getenv("GNSSCORR_SELFTEST_DUMP_DIR"))
        {
            config.set_property("Tracking_1C.dump", "true");
            config.set_property("Tracking_1C.dump_filename", std::string(dump_dir) + "/track_ch");
        }
    Gnss_Synchro syn;
    syn.System = 'G';
    syn.Signal[0] = '1';
    syn.Signal[1] = 'C';
    syn.PRN = 1;
    // acquisition (CPU reference path of configs[0]: 1 ms PCPS)
    GpsL1CaPcpsAcquisitionHip acq(&config, "Acquisition_1C", 1, 0);
    acq.set_channel(0);
    acq.set_gnss_synchro(&syn);
    acq.set_threshold(0.005f);
    acq.set_doppler_max(5000);
    acq.set_doppler_step(20);
    acq.init();
    acq.set_local_code();
    acq.set_state(1);
    size_t pos = 0;
    auto ablk = acq.block();
    while (ablk->events().empty() && pos + 1000 <= 20000) pos += ablk->work(x.data() + pos, 1000);
    EXPECT(ablk->events().size() == 1 && ablk->events()[0] == 1, "GPS acquisition failed");
    EXPECT(std::fabs(syn.Acq_delay_samples - 524.0) <= 1.0 && std::fabs(syn.Acq_doppler_hz - fd) <= 150.0, "GPS acquisition result %g samples, %g Hz", syn.Acq_delay_samples, syn.Acq_doppler_hz);
    std::printf("GPS acquisition: delay %.0f samples, Doppler %.0f Hz, stamp %llu\n", syn.Acq_delay_samples, syn.Acq_doppler_hz, (unsigned long long)syn.Acq_samplestamp_samples);
    // hand-over (ChannelFsm::Event_valid_acquisition -> trk_->start_tracking(), channel_fsm.cc:104-115, 210-218)
    GpsL1CaDllPllTrackingHip trk(&config, "Tracking_1C", 1, 1);
    EXPECT(trk.implementation() == "GPS_L1_CA_DLL_PLL_Tracking_HIP", "implementation name");
    EXPECT(trk.conf().vector_length == 4000, "vector_length %u", trk.conf().vector_length);
    trk.set_channel(0);
    trk.set_gnss_synchro(&syn);
    trk.start_tracking();
    run_tracking(trk, x, syn, fd, 5.0, cn0, "GPS L1 C/A tracking", 1150);
}

static void test_galileo_track()
{
    const double fs = 25e6, fd = -1234.0, cn0 = 45.0;
    std::vector<float> code(8184);
    char sig[3] = "1B";
    gc_galileo_e1_code_gen_sinboc11_float(code.data(), sig, 11);
    const double delay_samples = 33333.0;
    const double tau0 = 8184.0 - delay_samples * 2.046e6 / fs;
    auto x = synth(code, 2.046e6, 1575.42e6, fs, 100000 * 100, fd, tau0, cn0, 12);
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "25000000");
    config.set_property("Tracking_1B.pll_bw_hz", "12.0");
    config.set_property("Tracking_1B.dll_bw_hz", "2.0");
    Gnss_Synchro syn;
    syn.System = 'E';
    syn.Signal[0] = '1';
    syn.Signal[1] = 'B';
    syn.PRN = 11;
    syn.Acq_delay_samples = delay_samples;
    syn.Acq_doppler_hz = -1230.0;
    syn.Acq_samplestamp_samples = 0;
    GalileoE1DllPllVemlTrackingHip trk(&config, "Tracking_1B", 1, 1);
    EXPECT(trk.conf().vector_length == 100000, "Galileo vector_length %u", trk.conf().vector_length);
    trk.set_gnss_synchro(&syn);
    trk.start_tracking();
    run_tracking(trk, x, syn, fd, 3.0, cn0, "Galileo E1 tracking", 90);
    EXPECT(trk.block()->correlator_outs().size() == 5, "5 taps");
}

static void test_beidou_track()
{
    const double fs = 25e6, fd = 2222.0, cn0 = 47.0;
    std::vector<float> code(2046);
    gc_beidou_b1i_code_gen_float(code.data(), 6, 0);
    const double delay_samples = 7777.0;
    const double tau0 = 2046.0 - delay_samples * 2.046e6 / fs;
    auto x = synth(code, 2.046e6, 1.561098e9, fs, 25000 * 300, fd, tau0, cn0, 13);
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "25000000");
    config.set_property("Tracking_B1.pll_bw_hz", "40.0");
    Gnss_Synchro syn;
    syn.System = 'C';
    syn.Signal[0] = 'B';
    syn.Signal[1] = '1';
    syn.PRN = 6;
    syn.Acq_delay_samples = delay_samples;
    syn.Acq_doppler_hz = 2225.0;
    syn.Acq_samplestamp_samples = 0;
    BeidouB1iDllPllTrackingHip trk(&config, "Tracking_B1", 1, 1);
    trk.set_gnss_synchro(&syn);
    trk.start_tracking();
    run_tracking(trk, x, syn, fd, 5.0, cn0, "BeiDou B1I tracking", 280);
}

static void test_loss_of_lock()
{
    // noise only: the lock detectors must raise message 3 and put the block in standby
    const double fs = 4e6;
    std::vector<float> code(1023);
    gc_gps_l1_ca_code_gen_float(code.data(), 3, 0);
    auto x = synth(code, 1.023e6, 1575.42e6, fs, 4000 * 1800, 0.0, 0.0, -100.0, 14);
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "4000000");
    config.set_property("Tracking_1C.pull_in_time_s", "0");
    config.set_property("Tracking_1C.max_lock_fail", "10");
    Gnss_Synchro syn;
    syn.System = 'G';
    syn.PRN = 3;
    GpsL1CaDllPllTrackingHip trk(&config, "Tracking_1C", 1, 1);
    trk.set_gnss_synchro(&syn);
    trk.start_tracking();
    auto blk = trk.block();
    size_t pos = 0;
    Gnss_Synchro out;
    while (pos + blk->required_input_items() <= x.size() && blk->state() != 0)
        {
            int produced = 0;
            pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
        }
    EXPECT(blk->state() == 0 && blk->events().size() == 1 && blk->events()[0] == 3, "loss of lock not declared (state %d, %zu events)", blk->state(), blk->events().size());
    std::printf("noise only: loss of lock declared after %llu samples\n", (unsigned long long)blk->sample_counter());
}

int main()
{
    if (gc_device_count() == 0)
        {
            std::printf("no GPU: libgnsscorr has no CPU fallback\n");
            return 3;
        }
    test_gps_acq_then_track();
    test_galileo_track();
    test_beidou_track();
    test_loss_of_lock();
    std::printf(g_fail ? "%d FAILURES\n" : "tracking self-test passed\n", g_fail);
    return g_fail ? 1 : 0;
}
