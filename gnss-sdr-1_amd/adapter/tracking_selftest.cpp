// tracking_selftest.cpp -- closed-loop run of the tracking drop-in layer on the GPU (BASELINE configs[0]
// shape: GPS L1 C/A, one channel, 4 Msps, PCPS acquisition followed by 3-tap DLL/PLL tracking), plus
// Galileo E1 (5 taps, 4 ms) and BeiDou B1I hand-overs, and the synchronisation / extended-integration states: Galileo E1
// pilot tracking (secondary-code lock, 4 x 4 ms integration, data symbols from the E1-B prompt), BeiDou NH20 lock, GPS
// telemetry-preamble bit synchronisation with 20 ms integration.  The input is a synthetic noiseless + noisy signal
// with known Doppler / code phase; the loop must pull in and stay locked, and its Doppler / C/N0 estimates
// must converge to the truth.  Usage: tracking_selftest (needs a GPU).
#include "dll_pll_tracking_adapters.h"
#include "hip_glonass_ca_dll_pll_tracking.h"
#include "hip_gps_l1_ca_dll_pll_c_aid_tracking.h"
#include "hip_acquisition_bank.h"
#include "hip_tracking_group.h"
#include "pcps_acquisition_adapters.h"
#include <chrono>
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

// several components on one carrier, each with one symbol (+-1) per code period: period p of the stream carries
// symbols[p % size]; period 0 is the partial one before the first code start
struct Component
{
    std::vector<float> code;
    std::vector<float> symbols;
    bool quadrature = false;  // the component rides on j * carrier (GPS L5 Q5, Galileo E5a-Q)
};
static std::vector<gr_complex> synth_symbols(const std::vector<Component>& comps, double chip_rate_hz, double carrier_hz, double fs, size_t n, double fd,
    double delay_samples, double cn0_dbhz, unsigned seed, double phi = 0.7)
{
    std::mt19937 gen(seed);
    std::normal_distribution<float> nd(0.0f, std::sqrt(0.5f));
    const double amp = std::sqrt(std::pow(10.0, cn0_dbhz / 10.0) / fs);
    const double rate = chip_rate_hz * (1.0 + fd / carrier_hz) / fs;
    const size_t L = comps[0].code.size();
    const double tau0 = static_cast<double>(L) - delay_samples * chip_rate_hz / fs;
    std::vector<gr_complex> x(n);
    for (size_t i = 0; i < n; i++)
        {
            const double ph = 2.0 * M_PI * fd * static_cast<double>(i) / fs + phi;
            const size_t k = static_cast<size_t>(std::floor(tau0 + static_cast<double>(i) * rate));
            const size_t chip = k % L, period = k / L;
            double vi = 0.0, vq = 0.0;
            for (const auto& c : comps) (c.quadrature ? vq : vi) += c.code[chip] * c.symbols[period % c.symbols.size()];
            const double cs = std::cos(ph), sn = std::sin(ph);
            x[i] = gr_complex(static_cast<float>(amp * (vi * cs - vq * sn)) + nd(gen), static_cast<float>(amp * (vi * sn + vq * cs)) + nd(gen));
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
    // a signal with one symbol per bit and no secondary code (Galileo E1-B alone) is handed to state 4 at once
    EXPECT(blk->state() == 2 || blk->state() == 4, "%s: lost lock (state %d after %d epochs)", name, blk->state(), epochs);
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

static void test_galileo_pilot_extended()
{
    // E1-C pilot (secondary code CS25) drives the loop, 4 x 4 ms coherent integration after the secondary lock; the E1-B
    // prompt (d_Prompt_Data) carries the data symbols to the telemetry decoder
    const double fs = 4e6, fd = 2210.0, cn0 = 47.0, delay_samples = 3456.0;
    Component b, c;
    b.code.resize(8184);
    c.code.resize(8184);
    char sb[3] = "1B", sc[3] = "1C";
    gc_galileo_e1_code_gen_sinboc11_float(b.code.data(), sb, 19);
    gc_galileo_e1_code_gen_sinboc11_float(c.code.data(), sc, 19);
    std::mt19937 gen(5);
    for (int i = 0; i < 500; i++) b.symbols.push_back((gen() & 1u) ? 1.0f : -1.0f);
    const std::string cs25 = "0011100000001010110110010";
    for (char ch : cs25) c.symbols.push_back(ch == '0' ? 1.0f : -1.0f);
    const int n_periods = 160;
    auto x = synth_symbols({b, c}, 2.046e6, 1575.42e6, fs, 16000 * static_cast<size_t>(n_periods), fd, delay_samples, cn0, 21);
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "4000000");
    config.set_property("Tracking_1B.track_pilot", "true");
    config.set_property("Tracking_1B.extend_correlation_symbols", "4");
    config.set_property("Tracking_1B.pll_bw_hz", "25.0");
    config.set_property("Tracking_1B.dll_bw_hz", "2.0");
    config.set_property("Tracking_1B.pll_bw_narrow_hz", "10.0");
    config.set_property("Tracking_1B.dll_bw_narrow_hz", "0.5");
    config.set_property("Tracking_1B.early_late_space_narrow_chips", "0.1");
    config.set_property("Tracking_1B.very_early_late_space_narrow_chips", "0.5");
    config.set_property("Tracking_1B.pull_in_time_s", "0");
    Gnss_Synchro syn;
    syn.System = 'E';
    syn.Signal[0] = '1';
    syn.Signal[1] = 'B';
    syn.PRN = 19;
    syn.Acq_delay_samples = delay_samples;
    syn.Acq_doppler_hz = fd - 1.0;
    syn.Acq_samplestamp_samples = 0;
    GalileoE1DllPllVemlTrackingHip trk(&config, "Tracking_1B", 1, 1);
    EXPECT(trk.conf().track_pilot && trk.conf().extend_correlation_symbols == 4, "pilot configuration not taken");
    trk.set_gnss_synchro(&syn);
    trk.start_tracking();
    auto blk = trk.block();
    size_t pos = 0;
    int epochs = 0, first_ext = -1, agree = 0, counted = 0, updates = 0;
    Gnss_Synchro out;
    while (pos + blk->required_input_items() <= x.size() && blk->state() != 0)
        {
            int produced = 0;
            const int st_in = blk->state();
            pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
            if (st_in < 2) continue;
            if (blk->state() >= 3 && first_ext < 0) first_ext = epochs;
            if (first_ext >= 0 && epochs > first_ext + 40 && produced)
                {
                    // tracked period k is stream period k + 2 (the pull-in skips the first complete one)
                    counted++;
                    if ((out.Prompt_I > 0) == (b.symbols[(epochs + 2) % b.symbols.size()] > 0)) agree++;
                }
            if (st_in == 4) updates++;
            epochs++;
        }
    EXPECT(blk->last_status() == GC_OK, "pilot: engine status %d (%s)", blk->last_status(), gc_last_error());
    EXPECT(blk->state() == 3 || blk->state() == 4, "pilot: state %d", blk->state());
    // CS25 occupies stream periods 0..24, 25..49, ...: the first complete one seen by the loop ends at period 49 = epoch 47
    EXPECT(first_ext == 47, "pilot: secondary code locked at epoch %d", first_ext);
    EXPECT(updates >= (epochs - first_ext) / 4 - 1, "pilot: %d loop updates in %d periods", updates, epochs - first_ext);
    EXPECT(counted > 50 && (agree == counted || agree == 0), "pilot: %d of %d data symbols agree", agree, counted);
    EXPECT(std::fabs(blk->carrier_doppler_hz() - fd) < 2.0, "pilot: Doppler %.2f", blk->carrier_doppler_hz());
    EXPECT(blk->events().empty(), "pilot: loss-of-lock event");
    std::printf("Galileo E1 pilot tracking: secondary code locked at epoch %d, %d loop updates, %d / %d data symbols, Doppler %.2f Hz, C/N0 %.1f dB-Hz\n", first_ext, updates,
        agree, counted, blk->carrier_doppler_hz(), blk->cn0_db_hz());
}

static void test_beidou_secondary_lock()
{
    const double fs = 4.092e6, fd = -900.0, cn0 = 47.0, delay_samples = 1500.0;
    Component d;
    d.code.resize(2046);
    gc_beidou_b1i_code_gen_float(d.code.data(), 8, 0);
    const std::string nh = "00000100110101001110";
    for (char ch : nh) d.symbols.push_back(ch == '0' ? 1.0f : -1.0f);
    auto x = synth_symbols({d}, 2.046e6, 1.561098e9, fs, 4092 * 300, fd, delay_samples, cn0, 22);
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "4092000");
    config.set_property("Tracking_B1.pll_bw_hz", "40.0");
    config.set_property("Tracking_B1.pull_in_time_s", "0");
    Gnss_Synchro syn;
    syn.System = 'C';
    syn.Signal[0] = 'B';
    syn.Signal[1] = '1';
    syn.PRN = 8;
    syn.Acq_delay_samples = delay_samples;
    syn.Acq_doppler_hz = fd + 3.0;
    syn.Acq_samplestamp_samples = 0;
    BeidouB1iDllPllTrackingHip trk(&config, "Tracking_B1", 1, 1);
    trk.set_gnss_synchro(&syn);
    trk.start_tracking();
    auto blk = trk.block();
    size_t pos = 0;
    int epochs = 0, first4 = -1;
    Gnss_Synchro out;
    while (pos + blk->required_input_items() <= x.size() && blk->state() != 0)
        {
            int produced = 0;
            const int st_in = blk->state();
            pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
            if (st_in < 2) continue;
            if (blk->state() == 4 && first4 < 0) first4 = epochs;
            epochs++;
        }
    // NH20 occupies stream periods 0..19, 20..39: the first complete one seen by the loop ends at period 39 = epoch 37
    EXPECT(first4 == 37 && blk->state() == 4, "BeiDou: NH20 locked at epoch %d, state %d", first4, blk->state());
    EXPECT(std::fabs(blk->carrier_doppler_hz() - fd) < 5.0 && blk->events().empty(), "BeiDou: Doppler %.2f", blk->carrier_doppler_hz());
    std::printf("BeiDou B1I: NH20 locked at epoch %d, state %d, Doppler %.2f Hz, C/N0 %.1f dB-Hz\n", first4, blk->state(), blk->carrier_doppler_hz(), blk->cn0_db_hz());
}

static void test_gps_bit_synchronisation()
{
    // telemetry preamble 10001011 inside random bits; the block waits bit_sync_min_time_s (10 s in the reference, shortened
    // here), finds the preamble on the prompt signs and integrates 20 ms aligned with the bit edges.  Only the upright preamble
    // counts, so the test tries both polarities of the stream: exactly one of them synchronises.
    const double fs = 4e6, fd = 1500.0, cn0 = 50.0, delay_samples = 100.0;
    Component d;
    d.code.resize(1023);
    gc_gps_l1_ca_code_gen_float(d.code.data(), 17, 0);
    std::mt19937 gen(41);
    std::vector<int> bits;
    for (int i = 0; i < 6; i++) bits.push_back(gen() & 1u);
    for (int b : {1, 0, 0, 0, 1, 0, 1, 1}) bits.push_back(b);
    for (int i = 0; i < 8; i++) bits.push_back(gen() & 1u);
    for (int b : bits)
        for (int j = 0; j < 20; j++) d.symbols.push_back(b ? 1.0f : -1.0f);
    int synced = 0;
    for (float polarity : {1.0f, -1.0f})
        {
            Component c = d;
            for (auto& v : c.symbols) v *= polarity;
            auto x = synth_symbols({c}, 1.023e6, 1575.42e6, fs, 4000 * (20 * bits.size() - 10), fd, delay_samples, cn0, 23);
            InMemoryConfiguration config;
            config.set_property("GNSS-SDR.internal_fs_sps", "4000000");
            config.set_property("Tracking_1C.extend_correlation_symbols", "20");
            config.set_property("Tracking_1C.pll_bw_hz", "40.0");
            config.set_property("Tracking_1C.pll_bw_narrow_hz", "10.0");
            config.set_property("Tracking_1C.dll_bw_narrow_hz", "1.0");
            config.set_property("Tracking_1C.pull_in_time_s", "0");
            Gnss_Synchro syn;
            syn.System = 'G';
            syn.Signal[0] = '1';
            syn.Signal[1] = 'C';
            syn.PRN = 17;
            syn.Acq_delay_samples = delay_samples;
            syn.Acq_doppler_hz = fd + 4.0;
            syn.Acq_samplestamp_samples = 0;
            GpsL1CaDllPllTrackingHip trk(&config, "Tracking_1C", 1, 1);
            trk.set_gnss_synchro(&syn);
            auto blk = trk.block();
            blk->set_bit_sync_min_time_s(0.03f);
            trk.start_tracking();
            size_t pos = 0;
            int epochs = 0, first3 = -1;
            double ratio = 0.0;
            Gnss_Synchro out;
            while (pos + blk->required_input_items() <= x.size() && blk->state() != 0)
                {
                    int produced = 0;
                    const int st_in = blk->state();
                    pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
                    if (st_in < 2) continue;
                    if (blk->state() == 3 && first3 < 0) first3 = epochs;
                    if (st_in == 4 && first3 >= 0) ratio = std::abs(blk->correlator_outs()[1]);
                    epochs++;
                }
            if (first3 >= 0)
                {
                    synced++;
                    // the preamble's last symbol is stream period 20 * 14 - 1 = epoch 20 * 14 - 1 - 2
                    EXPECT(first3 == 20 * 14 - 3, "GPS: bit synchronisation at epoch %d", first3);
                    EXPECT(blk->state() == 3 || blk->state() == 4, "GPS: state %d", blk->state());
                    EXPECT(std::fabs(blk->carrier_doppler_hz() - fd) < 3.0 && blk->events().empty(), "GPS: Doppler %.2f", blk->carrier_doppler_hz());
                    std::printf("GPS L1 C/A: preamble found at epoch %d (polarity %+.0f), 20 ms integration, Doppler %.2f Hz, C/N0 %.1f dB-Hz, |P| %.0f\n", first3, polarity,
                        blk->carrier_doppler_hz(), blk->cn0_db_hz(), ratio);
                }
            else
                EXPECT(blk->state() == 2, "GPS: state %d without synchronisation", blk->state());
        }
    EXPECT(synced == 1, "GPS: %d of 2 polarities synchronised", synced);
}

// Pilot tracking of a signal whose data and pilot components are in quadrature (GPS L5: I5 data x NH10, Q5 pilot x NH20;
// Galileo E5a: E5a-I data x CS20, E5a-Q pilot x CS100 of the PRN): secondary-code lock on the pilot, extended integration,
// data symbols from the interchanged data prompt (interchange_iq).
template <class Trk>
static void pilot_quadrature_case(const char* name, const char* role, char system, const char* signal, uint32_t prn, const std::vector<float>& data_code,
    const std::vector<float>& pilot_code, const std::string& data_secondary, const std::string& pilot_secondary, double carrier_hz, unsigned seed)
{
    const double fs = 25e6, fd = -3100.0, cn0 = 48.0, delay_samples = 7777.0;
    Component d, p;
    d.code = data_code;
    p.code = pilot_code;
    p.quadrature = true;
    std::mt19937 gen(seed);
    // one data bit per data-secondary-code period
    const size_t n_periods = pilot_secondary.size() * 2 + 70;
    float bit = 1.0f;
    for (size_t k = 0; k < n_periods + 4; k++)
        {
            if (k % data_secondary.size() == 0) bit = (gen() & 1u) ? 1.0f : -1.0f;
            d.symbols.push_back(bit * (data_secondary[k % data_secondary.size()] == '0' ? 1.0f : -1.0f));
        }
    for (char ch : pilot_secondary) p.symbols.push_back(ch == '0' ? 1.0f : -1.0f);
    auto x = synth_symbols({d, p}, 10.23e6, carrier_hz, fs, 25000 * n_periods, fd, delay_samples, cn0, seed + 100);
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "25000000");
    config.set_property(std::string(role) + ".track_pilot", "true");
    config.set_property(std::string(role) + ".extend_correlation_symbols", "5");
    config.set_property(std::string(role) + ".pll_bw_hz", "40.0");
    config.set_property(std::string(role) + ".dll_bw_hz", "2.0");
    config.set_property(std::string(role) + ".pll_bw_narrow_hz", "12.0");
    config.set_property(std::string(role) + ".dll_bw_narrow_hz", "1.0");
    config.set_property(std::string(role) + ".early_late_space_narrow_chips", "0.4");
    config.set_property(std::string(role) + ".pull_in_time_s", "0");
    Gnss_Synchro syn;
    syn.System = system;
    syn.Signal[0] = signal[0];
    syn.Signal[1] = signal[1];
    syn.PRN = prn;
    syn.Acq_delay_samples = delay_samples;
    syn.Acq_doppler_hz = fd + 2.0;
    syn.Acq_samplestamp_samples = 0;
    Trk trk(&config, role, 1, 1);
    EXPECT(trk.conf().track_pilot && trk.conf().extend_correlation_symbols == 5 && trk.conf().vector_length == 25000, "%s: configuration", name);
    trk.set_gnss_synchro(&syn);
    trk.start_tracking();
    auto blk = trk.block();
    size_t pos = 0;
    int epochs = 0, first_ext = -1, agree = 0, counted = 0;
    Gnss_Synchro out;
    while (pos + blk->required_input_items() <= x.size() && blk->state() != 0)
        {
            int produced = 0;
            const int st_in = blk->state();
            pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
            if (st_in < 2) continue;
            if (blk->state() >= 3 && first_ext < 0) first_ext = epochs;
            if (first_ext >= 0 && epochs > first_ext + 30 && produced)
                {
                    counted++;
                    if ((out.Prompt_I > 0) == (d.symbols[(epochs + 2) % d.symbols.size()] > 0)) agree++;
                }
            epochs++;
        }
    EXPECT(blk->last_status() == GC_OK, "%s: engine status %d (%s)", name, blk->last_status(), gc_last_error());
    EXPECT(blk->state() == 3 || blk->state() == 4, "%s: state %d", name, blk->state());
    // the first complete pilot secondary code the loop sees ends at stream period 2 * length - 1 = epoch 2 * length - 3
    EXPECT(first_ext == static_cast<int>(2 * pilot_secondary.size()) - 3, "%s: secondary code locked at epoch %d", name, first_ext);
    EXPECT(counted > 20 && (agree == counted || agree == 0), "%s: %d of %d data symbols agree", name, agree, counted);
    EXPECT(std::fabs(blk->carrier_doppler_hz() - fd) < 6.0 && blk->events().empty(), "%s: Doppler %.2f", name, blk->carrier_doppler_hz());
    std::printf("%s: secondary code (%zu symbols) locked at epoch %d, %d / %d data symbols on the interchanged prompt, Doppler %.2f Hz, C/N0 %.1f dB-Hz\n", name,
        pilot_secondary.size(), first_ext, agree, counted, blk->carrier_doppler_hz(), blk->cn0_db_hz());
}

static void test_gps_l5_pilot()
{
    std::vector<float> i5(10230), q5(10230);
    gc_gps_l5i_code_gen_float(i5.data(), 7);
    gc_gps_l5q_code_gen_float(q5.data(), 7);
    pilot_quadrature_case<GpsL5DllPllTrackingHip>("GPS L5 pilot tracking", "Tracking_L5", 'G', "L5", 7, i5, q5, "0000110101", "00000100110101001110", 1.17645e9, 31);
}

static void test_galileo_e5a_pilot()
{
    std::vector<float> x(2 * 10230), i5(10230), q5(10230);
    gc_galileo_e5_a_code_gen_complex_primary(x.data(), 3, "5X");
    for (int i = 0; i < 10230; i++)
        {
            i5[i] = x[2 * i];
            q5[i] = x[2 * i + 1];
        }
    char sec[128];
    int32_t len = 0;
    EXPECT(gc_secondary_code("5Q", 3, sec, sizeof sec, &len) == GC_OK && len == 100, "E5a-Q secondary code");
    pilot_quadrature_case<GalileoE5aDllPllTrackingHip>("Galileo E5a pilot tracking", "Tracking_5X", 'E', "5X", 3, i5, q5, "10000100001011101001", std::string(sec, 100),
        1.17645e9, 32);
}

static void test_beidou_b3i_and_gps_l2c()
{
    {
        // B3I: NH20 on the data component, 10.23 Mcps
        const double fs = 25e6, fd = 1234.0, delay_samples = 4321.0;
        Component d;
        d.code.resize(10230);
        gc_beidou_b3i_code_gen_float(d.code.data(), 20, 0);
        for (char ch : std::string("00000100110101001110")) d.symbols.push_back(ch == '0' ? 1.0f : -1.0f);
        auto x = synth_symbols({d}, 10.23e6, 1.268520e9, fs, 25000 * 120, fd, delay_samples, 47.0, 41);
        InMemoryConfiguration config;
        config.set_property("GNSS-SDR.internal_fs_sps", "25000000");
        config.set_property("Tracking_B3.pull_in_time_s", "0");
        Gnss_Synchro syn;
        syn.System = 'C';
        syn.Signal[0] = 'B';
        syn.Signal[1] = '3';
        syn.PRN = 20;
        syn.Acq_delay_samples = delay_samples;
        syn.Acq_doppler_hz = fd - 3.0;
        syn.Acq_samplestamp_samples = 0;
        BeidouB3iDllPllTrackingHip trk(&config, "Tracking_B3", 1, 1);
        EXPECT(trk.implementation() == "BEIDOU_B3I_DLL_PLL_Tracking_HIP" && trk.conf().vector_length == 25000, "B3I adapter");
        trk.set_gnss_synchro(&syn);
        trk.start_tracking();
        auto blk = trk.block();
        size_t pos = 0;
        int epochs = 0, first4 = -1;
        Gnss_Synchro out;
        while (pos + blk->required_input_items() <= x.size() && blk->state() != 0)
            {
                int produced = 0;
                const int st_in = blk->state();
                pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
                if (st_in < 2) continue;
                if (blk->state() == 4 && first4 < 0) first4 = epochs;
                epochs++;
            }
        EXPECT(first4 == 37 && blk->state() == 4, "B3I: NH20 locked at epoch %d, state %d", first4, blk->state());
        EXPECT(std::fabs(blk->carrier_doppler_hz() - fd) < 5.0 && blk->events().empty(), "B3I: Doppler %.2f", blk->carrier_doppler_hz());
        std::printf("BeiDou B3I: NH20 locked at epoch %d, Doppler %.2f Hz, C/N0 %.1f dB-Hz\n", first4, blk->carrier_doppler_hz(), blk->cn0_db_hz());
    }
    {
        // L2C(M): one 20 ms code per symbol -> narrow tracking (state 4) from the first loop update
        const double fs = 2.046e6, fd = -800.0, delay_samples = 12345.0;
        Component d;
        d.code.resize(10230);
        gc_gps_l2c_m_code_gen_float(d.code.data(), 12);
        d.symbols = {1.0f, -1.0f, -1.0f, 1.0f, -1.0f};
        auto x = synth_symbols({d}, 0.5115e6, 1.22760e9, fs, 40920 * 60, fd, delay_samples, 42.0, 42);
        InMemoryConfiguration config;
        config.set_property("GNSS-SDR.internal_fs_sps", "2046000");
        config.set_property("Tracking_2S.pull_in_time_s", "0");
        config.set_property("Tracking_2S.extend_correlation_symbols", "4");  // not allowed on L2C(M): forced to 1
        Gnss_Synchro syn;
        syn.System = 'G';
        syn.Signal[0] = '2';
        syn.Signal[1] = 'S';
        syn.PRN = 12;
        syn.Acq_delay_samples = delay_samples;
        syn.Acq_doppler_hz = fd + 1.0;
        syn.Acq_samplestamp_samples = 0;
        GpsL2MDllPllTrackingHip trk(&config, "Tracking_2S", 1, 1);
        EXPECT(trk.conf().vector_length == 40920 && trk.conf().extend_correlation_symbols == 1, "L2C adapter: %u samples, extension %d", trk.conf().vector_length,
            trk.conf().extend_correlation_symbols);
        trk.set_gnss_synchro(&syn);
        trk.start_tracking();
        auto blk = trk.block();
        size_t pos = 0;
        int epochs = 0, agree = 0;
        Gnss_Synchro out;
        while (pos + blk->required_input_items() <= x.size() && blk->state() != 0)
            {
                int produced = 0;
                const int st_in = blk->state();
                pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
                if (st_in < 2) continue;
                if (produced && epochs > 20 && (out.Prompt_I > 0) == (d.symbols[(epochs + 2) % d.symbols.size()] > 0)) agree++;
                epochs++;
            }
        EXPECT(blk->state() == 4 && blk->events().empty(), "L2C: state %d", blk->state());
        EXPECT(std::fabs(blk->carrier_doppler_hz() - fd) < 1.0, "L2C: Doppler %.2f", blk->carrier_doppler_hz());
        EXPECT(agree == epochs - 21 || agree == 0, "L2C: %d of %d symbols", agree, epochs - 21);
        std::printf("GPS L2C(M): %d 20 ms periods in state 4, Doppler %.2f Hz, C/N0 %.1f dB-Hz, symbols %d / %d\n", epochs, blk->carrier_doppler_hz(), blk->cn0_db_hz(),
            agree, epochs - 21);
    }
}

static void test_glonass_fdma_tracking()
{
    // GLONASS L1 C/A, slot 22 = frequency channel -3: the signal sits 3 x 562.5 kHz below the band centre; the block's carrier NCO
    // carries that offset, its Doppler bookkeeping does not (glonass_l1_ca_dll_pll_tracking_cc.cc:173-214, :604-639)
    const double fs = 6.625e6, fd = 2100.0, cn0 = 46.0, delay_samples = 1343.0;
    const double f_channel = -3 * 562500.0;
    std::vector<float> code(511);
    gc_glonass_l1_ca_code_gen_float(code.data(), 0);
    // the code Doppler follows the slot's own carrier (1602 MHz - 1.6875 MHz)
    auto x = synth(code, 0.511e6, 1.602e9 + f_channel, fs, 6625 * 900, fd, 511.0 - delay_samples * 0.511e6 / fs, cn0, 51);
    // shift the whole signal (not the noise statistics) down to the channel's centre
    {
        std::mt19937 gen(52);
        for (size_t i = 0; i < x.size(); i++)
            {
                const double ph = 2.0 * M_PI * std::fmod(f_channel * static_cast<double>(i) / fs, 1.0);
                x[i] *= gr_complex(static_cast<float>(std::cos(ph)), static_cast<float>(std::sin(ph)));
            }
    }
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "6625000");
    config.set_property("Tracking_1G.pll_bw_hz", "40.0");
    config.set_property("Tracking_1G.dll_bw_hz", "3.0");
    Gnss_Synchro syn;
    syn.System = 'R';
    syn.Signal[0] = '1';
    syn.Signal[1] = 'G';
    syn.PRN = 22;
    syn.Acq_delay_samples = delay_samples;
    syn.Acq_doppler_hz = fd + 40.0;
    syn.Acq_samplestamp_samples = 0;
    GlonassL1CaDllPllTrackingHip trk(&config, "Tracking_1G", 1, 1);
    EXPECT(trk.implementation() == "GLONASS_L1_CA_DLL_PLL_Tracking_HIP" && trk.vector_length() == 6625, "GLONASS adapter: %u samples", trk.vector_length());
    trk.set_gnss_synchro(&syn);
    trk.start_tracking();
    auto blk = trk.block();
    size_t pos = 0;
    int epochs = 0, averaged = 0;
    double mean_doppler = 0.0, mean_nco = 0.0;
    Gnss_Synchro out;
    while (pos + blk->required_input_items() <= x.size() && blk->tracking_enabled())
        {
            int produced = 0;
            pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
            if (epochs >= 400)
                {
                    // the block's frequency estimate is the running sum of the loop filter output: noisy period by period
                    // (some +-15 Hz at 40 Hz of bandwidth), unbiased on average
                    mean_doppler += blk->carrier_doppler_hz();
                    mean_nco += blk->carrier_frequency_hz();
                    averaged++;
                }
            if (std::getenv("GNSSCORR_SELFTEST_VERBOSE") && epochs % 50 == 0)
                std::printf("  %d: Doppler %.3f, P = (%.0f, %.0f), counter %llu\n", epochs, blk->carrier_doppler_hz(), blk->correlator_outs()[1].real(),
                    blk->correlator_outs()[1].imag(), (unsigned long long)blk->sample_counter());
            epochs++;
        }
    EXPECT(blk->last_status() == GC_OK, "GLONASS: engine status %d (%s)", blk->last_status(), gc_last_error());
    EXPECT(blk->tracking_enabled() && blk->events().empty(), "GLONASS: lost lock after %d periods", epochs);
    mean_doppler /= std::max(1, averaged);
    mean_nco /= std::max(1, averaged);
    EXPECT(averaged > 400 && std::fabs(mean_doppler - fd) < 3.0, "GLONASS: mean Doppler %.2f Hz over %d periods, truth %.2f", mean_doppler, averaged, fd);
    EXPECT(std::fabs(mean_nco - (fd + f_channel)) < 3.0, "GLONASS: mean NCO frequency %.2f Hz", mean_nco);
    EXPECT(std::fabs(blk->cn0_db_hz() - cn0) < 3.0 && blk->carrier_lock_test() > 0.85, "GLONASS: C/N0 %.1f, lock test %.3f", blk->cn0_db_hz(), blk->carrier_lock_test());
    const auto& c = blk->correlator_outs();
    EXPECT(std::abs(c[1]) > std::abs(c[0]) && std::abs(c[1]) > std::abs(c[2]), "GLONASS: prompt is not the largest tap");
    std::printf("GLONASS L1 C/A (slot 22, channel -3): %d periods, mean Doppler %.2f Hz (truth %.2f), mean NCO %.2f Hz, C/N0 %.1f dB-Hz, lock test %.3f\n", epochs,
        mean_doppler, fd, mean_nco, blk->cn0_db_hz(), blk->carrier_lock_test());
}

// gps_l1_ca_dll_pll_c_aid_tracking_cc / _sc: carrier-aided DLL; the telemetry decoder's preamble time stamp switches the block to
// extend_correlation_ms-long coherent sums of its correlator history and to the narrow bandwidths
template <class Item, class Block>
static void c_aid_case(const char* name, std::shared_ptr<Block> blk, GpsL1CaDllPllCAidTrackingHip& trk, const std::vector<Item>& x, double fd, double cn0, double fs)
{
    Gnss_Synchro syn;
    syn.System = 'G';
    syn.Signal[0] = '1';
    syn.Signal[1] = 'C';
    syn.PRN = 9;
    syn.Acq_delay_samples = 2001.0;
    syn.Acq_doppler_hz = fd + 25.0;
    syn.Acq_samplestamp_samples = 0;
    trk.set_gnss_synchro(&syn);
    trk.start_tracking();
    size_t pos = 0;
    int epochs = 0, updates_after = 0, valid_after = 0, len5 = 0;
    Gnss_Synchro out;
    double mean_doppler = 0.0;
    int averaged = 0;
    while (pos + blk->required_input_items() <= x.size() && blk->tracking_enabled())
        {
            int produced = 0;
            pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
            if (epochs == 400) blk->set_preamble_timestamp_s(static_cast<double>(blk->sample_counter()) / fs);  // "a preamble started here"
            if (epochs > 420)
                {
                    updates_after++;
                    if (out.Flag_valid_symbol_output)
                        {
                            valid_after++;
                            if (out.correlation_length_ms == 5) len5++;
                        }
                }
            if (epochs >= 600)
                {
                    mean_doppler += blk->carrier_doppler_hz();
                    averaged++;
                }
            epochs++;
        }
    mean_doppler /= std::max(1, averaged);
    EXPECT(blk->last_status() == GC_OK, "%s: engine status %d (%s)", name, blk->last_status(), gc_last_error());
    EXPECT(blk->tracking_enabled() && blk->events().empty() && epochs > 1100, "%s: lost lock after %d periods", name, epochs);
    EXPECT(blk->preamble_synchronized(), "%s: extended integration never started", name);
    // one loop update per 5 code periods once the preamble stamp is in
    EXPECT(valid_after > 0 && len5 == valid_after && std::abs(5 * valid_after - updates_after) <= 6, "%s: %d loop updates in %d periods, %d of 5 ms", name, valid_after,
        updates_after, len5);
    EXPECT(averaged > 300 && std::fabs(mean_doppler - fd) < 2.0, "%s: mean Doppler %.2f Hz, truth %.2f", name, mean_doppler, fd);
    EXPECT(std::fabs(blk->cn0_db_hz() - (cn0 + 7.0)) < 4.0 && blk->carrier_lock_test() > 0.85, "%s: C/N0 %.1f dB-Hz, lock test %.3f", name, blk->cn0_db_hz(),
        blk->carrier_lock_test());
    std::printf("%s: %d periods, %d loop updates of 5 ms after the preamble stamp, mean Doppler %.2f Hz (truth %.2f), C/N0 reading %.1f dB-Hz, lock test %.3f\n", name,
        epochs, valid_after, mean_doppler, fd, blk->cn0_db_hz(), blk->carrier_lock_test());
}

static void test_gps_c_aid_tracking()
{
    const double fs = 4e6, fd = -1500.0, cn0 = 45.0;
    std::vector<float> code(1023);
    gc_gps_l1_ca_code_gen_float(code.data(), 9, 0);
    auto x = synth(code, 1.023e6, 1575.42e6, fs, 4000 * 1200, fd, 1023.0 - 2001.0 * 1.023e6 / fs, cn0, 61);
    {
        InMemoryConfiguration config;
        config.set_property("GNSS-SDR.internal_fs_sps", "4000000");
        config.set_property("Tracking_1C.extend_correlation_ms", "5");
        config.set_property("Tracking_1C.pll_bw_hz", "35.0");
        config.set_property("Tracking_1C.pll_bw_narrow_hz", "15.0");
        GpsL1CaDllPllCAidTrackingHip trk(&config, "Tracking_1C", 1, 1);
        EXPECT(trk.implementation() == "GPS_L1_CA_DLL_PLL_C_Aid_Tracking_HIP" && trk.item_type() == "gr_complex" && trk.vector_length() == 4000, "C-aid adapter");
        c_aid_case<gr_complex>("GPS L1 C/A C-aid tracking (gr_complex)", trk.block_gr_complex(), trk, x, fd, cn0, fs);
    }
    {
        // cshort items: the 16-bit correlator (Cpu_Multicorrelator_16sc arithmetic); the stream is scaled so that five summed prompts stay inside int16, as the block sums them in lv_16sc_t
        std::vector<std::complex<int16_t>> xs(x.size());
        for (size_t i = 0; i < x.size(); i++)
            xs[i] = std::complex<int16_t>(static_cast<int16_t>(std::lrint(x[i].real() * 10.0f)), static_cast<int16_t>(std::lrint(x[i].imag() * 10.0f)));
        InMemoryConfiguration config;
        config.set_property("GNSS-SDR.internal_fs_sps", "4000000");
        config.set_property("Tracking_1C.item_type", "cshort");
        config.set_property("Tracking_1C.extend_correlation_ms", "5");
        config.set_property("Tracking_1C.pll_bw_hz", "35.0");
        config.set_property("Tracking_1C.pll_bw_narrow_hz", "15.0");
        GpsL1CaDllPllCAidTrackingHip trk(&config, "Tracking_1C", 1, 1);
        EXPECT(trk.item_type() == "cshort" && trk.item_size() == 4, "C-aid adapter (cshort)");
        c_aid_case<std::complex<int16_t>>("GPS L1 C/A C-aid tracking (cshort)", trk.block_cshort(), trk, xs, fd, cn0, fs);
    }
}

// The same block on the device loop: a work() call brings a chunk of items and gets back one Gnss_Synchro per complete code
// period -- against the host-loop block fed the same stream: identical block boundaries and states, Doppler within a fraction of a Hz.
static void test_device_loop_block()
{
    const double fs = 4e6, fd = 2210.0, cn0 = 47.0, delay_samples = 3456.0;
    Component b, c;
    b.code.resize(8184);
    c.code.resize(8184);
    char sb[3] = "1B", sc[3] = "1C";
    gc_galileo_e1_code_gen_sinboc11_float(b.code.data(), sb, 19);
    gc_galileo_e1_code_gen_sinboc11_float(c.code.data(), sc, 19);
    std::mt19937 gen(5);
    for (int i = 0; i < 500; i++) b.symbols.push_back((gen() & 1u) ? 1.0f : -1.0f);
    for (char ch : std::string("0011100000001010110110010")) c.symbols.push_back(ch == '0' ? 1.0f : -1.0f);
    auto x = synth_symbols({b, c}, 2.046e6, 1575.42e6, fs, 16000 * 160, fd, delay_samples, cn0, 21);
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "4000000");
    config.set_property("Tracking_1B.track_pilot", "true");
    config.set_property("Tracking_1B.extend_correlation_symbols", "4");
    config.set_property("Tracking_1B.pll_bw_hz", "25.0");
    config.set_property("Tracking_1B.dll_bw_hz", "2.0");
    config.set_property("Tracking_1B.pll_bw_narrow_hz", "10.0");
    config.set_property("Tracking_1B.dll_bw_narrow_hz", "0.5");
    config.set_property("Tracking_1B.early_late_space_narrow_chips", "0.1");
    config.set_property("Tracking_1B.very_early_late_space_narrow_chips", "0.5");
    config.set_property("Tracking_1B.pull_in_time_s", "0");
    Gnss_Synchro syn;
    syn.System = 'E';
    syn.Signal[0] = '1';
    syn.Signal[1] = 'B';
    syn.PRN = 19;
    syn.Acq_delay_samples = delay_samples;
    syn.Acq_doppler_hz = fd - 1.0;
    syn.Acq_samplestamp_samples = 0;
    // host-loop block: one period per call
    std::vector<Gnss_Synchro> host;
    double host_s = 0.0, dev_s = 0.0;
    {
        GalileoE1DllPllVemlTrackingHip trk(&config, "Tracking_1B", 1, 1);
        trk.set_gnss_synchro(&syn);
        trk.start_tracking();
        auto blk = trk.block();
        size_t pos = 0;
        Gnss_Synchro out;
        const auto t0 = std::chrono::steady_clock::now();
        while (pos + blk->required_input_items() <= x.size() && blk->state() != 0)
            {
                int produced = 0;
                pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
                if (produced) host.push_back(out);
            }
        host_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // device-loop block: chunks of uneven size, as a scheduler would deliver them
    GalileoE1DllPllVemlTrackingHipDev trk(&config, "Tracking_1B", 1, 1);
    EXPECT(trk.implementation() == "Galileo_E1_DLL_PLL_VEML_Tracking_HIP_DEV", "device adapter name %s", trk.implementation().c_str());
    trk.set_gnss_synchro(&syn);
    auto blk = trk.block();
    EXPECT(blk->last_status() == GC_OK, "device block: status %d (%s)", blk->last_status(), gc_last_error());
    trk.start_tracking();
    std::vector<Gnss_Synchro> dev, outs(80);
    size_t pos = 0;
    int calls = 0;
    std::mt19937 chunker(7);
    const auto t1 = std::chrono::steady_clock::now();
    while (pos + blk->required_input_items() <= x.size() && blk->state() != 0)
        {
            // the scheduler shows between 2 and 40 periods' worth of items, including those the block did not consume last time
            const size_t avail = std::min<size_t>(x.size() - pos, 16000 * (2 + chunker() % 39) + chunker() % 5000);
            int produced = 0;
            const int used = blk->work(x.data() + pos, static_cast<int>(avail), outs.data(), static_cast<int>(outs.size()), &produced);
            EXPECT(blk->last_status() == GC_OK, "device block: status %d (%s)", blk->last_status(), gc_last_error());
            if (blk->last_status() != GC_OK) break;
            for (int k = 0; k < produced; k++) dev.push_back(outs[k]);
            pos += used;
            calls++;
            if (used == 0 && produced == 0 && avail == x.size() - pos) break;  // the tail is shorter than a period
        }
    dev_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    EXPECT(dev.size() + 1 >= host.size() && host.size() > 150, "device block: %zu outputs, host block %zu", dev.size(), host.size());
    size_t same_counter = 0;
    double worst_doppler = 0.0, worst_prompt = 0.0;
    const size_t n = std::min(dev.size(), host.size());
    for (size_t k = 0; k < n; k++)
        {
            if (dev[k].Tracking_sample_counter == host[k].Tracking_sample_counter) same_counter++;
            worst_doppler = std::max(worst_doppler, std::fabs(dev[k].Carrier_Doppler_hz - host[k].Carrier_Doppler_hz));
            worst_prompt = std::max(worst_prompt, std::fabs(dev[k].Prompt_I - host[k].Prompt_I) / (std::fabs(host[k].Prompt_I) + 50.0));
        }
    EXPECT(same_counter == n, "device block: %zu of %zu block boundaries agree with the host block", same_counter, n);
    EXPECT(worst_doppler < 0.2 && worst_prompt < 0.02, "device block: Doppler differs by %.3f Hz, prompt by %.4f", worst_doppler, worst_prompt);
    EXPECT(blk->state() == 3 || blk->state() == 4, "device block: state %d", blk->state());
    std::printf("device-loop block: %zu Gnss_Synchro in %d work() calls, %.1f ms (host-loop block: %zu in as many calls, %.1f ms) for %.0f ms of signal; "
                "boundaries identical, Doppler within %.3f Hz\n",
        dev.size(), calls, dev_s * 1e3, host.size(), host_s * 1e3, x.size() / fs * 1e3, worst_doppler);
}

// All channels of a signal as one object on one RF stream ring: four satellites in the stream, six slots, hand-overs at different
// times, one slot pointed at a satellite that is not there (it loses lock and frees itself), one launch per pushed block.
static void test_tracking_group(int iq_format)
{
    const double fs = 4e6;
    const int prns[4] = {3, 11, 19, 27};
    const double dopplers[4] = {1200.0, -2500.0, 300.0, 3800.0};
    const double delays[4] = {100.0, 1500.0, 2900.0, 3999.0};
    const size_t n = 4000 * 1500;  // the lock detector only counts once the pull-in transitory (at least 1 s) is over
    std::vector<gr_complex> x(n, gr_complex(0, 0));
    for (int k = 0; k < 4; k++)
        {
            std::vector<float> code(1023);
            gc_gps_l1_ca_code_gen_float(code.data(), prns[k], 0);
            // noise once (first satellite), signal only for the others
            auto xs = synth(code, 1.023e6, 1575.42e6, fs, n, dopplers[k], 1023.0 - delays[k] * 1.023e6 / fs, 46.0, 70 + k);
            if (k == 0)
                x = xs;
            else
                {
                    auto noise_free = synth(code, 1.023e6, 1575.42e6, fs, n, dopplers[k], 1023.0 - delays[k] * 1.023e6 / fs, 46.0, 70 + k);
                    // subtract this call's own noise by regenerating it at -infinity dB
                    auto only_noise = synth(code, 1.023e6, 1575.42e6, fs, n, dopplers[k], 1023.0 - delays[k] * 1.023e6 / fs, -300.0, 70 + k);
                    for (size_t i = 0; i < n; i++) x[i] += noise_free[i] - only_noise[i];
                }
        }
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "4000000");
    config.set_property("Tracking_1C.pll_bw_hz", "35.0");
    config.set_property("Tracking_1C.pull_in_time_s", "0");
    config.set_property("Tracking_1C.max_lock_fail", "5");
    config.set_property("Tracking_1C.cn0_samples", "10");
    GpsL1CaDllPllTrackingHip conf_source(&config, "Tracking_1C", 1, 1);  // the adapter turns the configuration into a Dll_Pll_Conf
    const Dll_Pll_Conf conf = conf_source.conf();
    gc_ctx* ctx = nullptr;
    gc_stream* ring = nullptr;
    EXPECT(gc_ctx_create(0, &ctx) == GC_OK, "group: context");
    EXPECT(gc_stream_create(ctx, iq_format, 4000 * 128, 8000, &ring) == GC_OK, "group: ring (%s)", gc_last_error());
    // a cshort front end: 12 significant bits around unit-variance noise
    std::vector<std::complex<int16_t>> xs;
    if (iq_format == GC_IQ_I16)
        {
            xs.resize(n);
            for (size_t i = 0; i < n; i++)
                xs[i] = std::complex<int16_t>(static_cast<int16_t>(std::lrint(x[i].real() * 256.0f)), static_cast<int16_t>(std::lrint(x[i].imag() * 256.0f)));
        }
    {
        hip_tracking_group group(ctx, ring, conf, 6, iq_format);
        EXPECT(group.last_status() == GC_OK, "group: status %d (%s)", group.last_status(), gc_last_error());
        auto acq_of = [&](int k, uint64_t stamp) {
            Gnss_Synchro a;
            a.System = 'G';
            a.Signal[0] = '1';
            a.Signal[1] = 'C';
            a.PRN = prns[k];
            a.Acq_delay_samples = delays[k];
            a.Acq_doppler_hz = dopplers[k] + 30.0;
            a.Acq_samplestamp_samples = stamp;
            return a;
        };
        std::vector<std::vector<Gnss_Synchro>> out;
        const size_t block = 4000 * 40;  // 40 ms per push
        int launches = 0;
        for (size_t pos = 0, b = 0; pos + block <= n; pos += block, b++)
            {
                const void* src = iq_format == GC_IQ_I16 ? static_cast<const void*>(xs.data() + pos) : static_cast<const void*>(x.data() + pos);
                EXPECT(gc_stream_push(ring, src, block, nullptr) == GC_OK, "group: push (%s)", gc_last_error());
                // hand-overs as acquisitions would deliver them; the code delay is relative to the acquisition stamp (a multiple of a code period here)
                if (b == 0)
                    {
                        EXPECT(group.start_tracking(0, acq_of(0, 0), 0) == GC_OK, "group: start 0 (%s)", gc_last_error());
                        EXPECT(group.start_tracking(4, acq_of(1, 0), 0) == GC_OK, "group: start 4 (%s)", gc_last_error());
                    }
                if (b == 3)
                    {
                        EXPECT(group.start_tracking(2, acq_of(2, pos), pos) == GC_OK, "group: start 2 (%s)", gc_last_error());
                        Gnss_Synchro ghost = acq_of(3, pos);
                        ghost.PRN = 30;  // not in the stream
                        EXPECT(group.start_tracking(5, ghost, pos) == GC_OK, "group: start 5 (%s)", gc_last_error());
                    }
                if (b == 6) EXPECT(group.start_tracking(1, acq_of(3, pos), pos) == GC_OK, "group: start 1 (%s)", gc_last_error());
                const int produced = group.run(out);
                EXPECT(produced >= 0, "group: run failed, status %d (%s)", group.last_status(), gc_last_error());
                if (produced < 0) break;
                launches++;
            }
        const int slot_of[4] = {0, 4, 2, 1};
        const size_t expect_min[4] = {1470, 1470, 1350, 1230};
        for (int k = 0; k < 4; k++)
            {
                const auto& o = out[slot_of[k]];
                EXPECT(o.size() >= expect_min[k] && group.active(slot_of[k]), "group: PRN %d produced %zu items", prns[k], o.size());
                if (o.empty()) continue;
                double mean = 0.0;
                int cnt = 0;
                for (size_t i = o.size() > 200 ? o.size() - 200 : 0; i < o.size(); i++, cnt++) mean += o[i].Carrier_Doppler_hz;
                mean /= std::max(1, cnt);
                EXPECT(std::fabs(mean - dopplers[k]) < 3.0 && o.back().PRN == static_cast<uint32_t>(prns[k]), "group: PRN %d mean Doppler %.2f Hz, truth %.2f", prns[k], mean,
                    dopplers[k]);
                EXPECT(o.back().CN0_dB_hz > 40.0, "group: PRN %d C/N0 %.1f", prns[k], o.back().CN0_dB_hz);
                // consecutive periods, one Gnss_Synchro each
                bool monotone = true;
                for (size_t i = 1; i < o.size(); i++) monotone = monotone && o[i].Tracking_sample_counter > o[i - 1].Tracking_sample_counter && o[i].Tracking_sample_counter - o[i - 1].Tracking_sample_counter < 4003;
                EXPECT(monotone, "group: PRN %d sample counters are not consecutive code periods", prns[k]);
            }
        EXPECT(!group.active(5) && group.events(5).size() == 1 && group.events(5)[0] == 3, "group: the ghost satellite's slot did not report loss of lock (%zu events)",
            group.events(5).size());
        EXPECT(!group.active(3) && out[3].empty(), "group: the unused slot produced %zu items", out[3].size());
        std::printf("tracking group (%s ring): 4 satellites + 1 ghost on 6 slots, %d launches for %.0f ms of signal: %zu / %zu / %zu / %zu Gnss_Synchro, ghost lost lock after %zu\n",
            iq_format == GC_IQ_I16 ? "cshort" : "gr_complex", launches, n / fs * 1e3, out[0].size(), out[4].size(), out[2].size(), out[1].size(), out[5].size());
    }
    gc_stream_destroy(ring);
    gc_ctx_destroy(ctx);
}

static void test_glonass_c_aid_tracking()
{
    // the GLONASS twins of the C-aid block (glonass_l1_ca_dll_pll_c_aid_tracking_cc): FDMA channel in the loop filter's accumulator
    const double fs = 6.625e6, fd = -1800.0, cn0 = 47.0, delay_samples = 3000.0;
    const double f_channel = 4 * 562500.0;  // slot 21 = channel +4
    std::vector<float> code(511);
    gc_glonass_l1_ca_code_gen_float(code.data(), 0);
    auto x = synth(code, 0.511e6, 1.602e9 + f_channel, fs, 6625 * 900, fd, 511.0 - delay_samples * 0.511e6 / fs, cn0, 81);
    for (size_t i = 0; i < x.size(); i++)
        {
            const double ph = 2.0 * M_PI * std::fmod(f_channel * static_cast<double>(i) / fs, 1.0);
            x[i] *= gr_complex(static_cast<float>(std::cos(ph)), static_cast<float>(std::sin(ph)));
        }
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "6625000");
    config.set_property("Tracking_1G.pll_bw_hz", "35.0");
    config.set_property("Tracking_1G.dll_bw_hz", "3.0");
    Gnss_Synchro syn;
    syn.System = 'R';
    syn.Signal[0] = '1';
    syn.Signal[1] = 'G';
    syn.PRN = 21;
    syn.Acq_delay_samples = delay_samples;
    syn.Acq_doppler_hz = fd - 20.0;
    syn.Acq_samplestamp_samples = 0;
    GlonassL1CaDllPllCAidTrackingHip trk(&config, "Tracking_1G", 1, 1);
    EXPECT(trk.implementation() == "GLONASS_L1_CA_DLL_PLL_C_Aid_Tracking_HIP" && trk.vector_length() == 6625, "GLONASS C-aid adapter");
    trk.set_gnss_synchro(&syn);
    trk.start_tracking();
    auto blk = trk.block_gr_complex();
    size_t pos = 0;
    int epochs = 0, averaged = 0;
    double mean_nco = 0.0;
    Gnss_Synchro out;
    while (pos + blk->required_input_items() <= x.size() && blk->tracking_enabled())
        {
            int produced = 0;
            pos += blk->work(x.data() + pos, static_cast<int>(x.size() - pos), &out, &produced);
            if (epochs >= 400)
                {
                    mean_nco += blk->carrier_doppler_hz();  // the accumulator of this block holds Doppler + channel offset
                    averaged++;
                }
            epochs++;
        }
    mean_nco /= std::max(1, averaged);
    EXPECT(blk->last_status() == GC_OK, "GLONASS C-aid: engine status %d (%s)", blk->last_status(), gc_last_error());
    EXPECT(blk->tracking_enabled() && blk->events().empty() && epochs > 850, "GLONASS C-aid: lost lock after %d periods", epochs);
    EXPECT(std::fabs(mean_nco - (fd + f_channel)) < 3.0, "GLONASS C-aid: mean NCO frequency %.2f Hz, truth %.2f", mean_nco, fd + f_channel);
    EXPECT(blk->carrier_lock_test() > 0.85 && std::fabs(blk->cn0_db_hz() - cn0) < 3.5, "GLONASS C-aid: C/N0 %.1f dB-Hz, lock test %.3f", blk->cn0_db_hz(),
        blk->carrier_lock_test());
    std::printf("GLONASS L1 C/A C-aid (slot 21, channel +4): %d periods, mean NCO frequency %.2f Hz (truth %.2f), C/N0 %.1f dB-Hz, lock test %.3f\n", epochs, mean_nco,
        fd + f_channel, blk->cn0_db_hz(), blk->carrier_lock_test());
}

// The two stages as two objects on one ring: hip_acquisition_bank searches all 32 GPS PRNs at once on the first block, its detections
// go straight into hip_tracking_group slots, the rest of the stream is tracked one launch per pushed block.
static void test_acquisition_bank_into_tracking_group()
{
    const double fs = 4e6;
    const int prns[3] = {5, 14, 23};
    const double dopplers[3] = {-3300.0, 450.0, 2750.0};
    const double delays[3] = {250.0, 2020.0, 3700.0};
    const size_t n = 4000 * 600;
    std::vector<gr_complex> x;
    for (int k = 0; k < 3; k++)
        {
            std::vector<float> code(1023);
            gc_gps_l1_ca_code_gen_float(code.data(), prns[k], 0);
            auto with_noise = synth(code, 1.023e6, 1575.42e6, fs, n, dopplers[k], 1023.0 - delays[k] * 1.023e6 / fs, 48.0, 90 + k);
            if (k == 0)
                x = with_noise;
            else
                {
                    auto only_noise = synth(code, 1.023e6, 1575.42e6, fs, n, dopplers[k], 1023.0 - delays[k] * 1.023e6 / fs, -300.0, 90 + k);
                    for (size_t i = 0; i < n; i++) x[i] += with_noise[i] - only_noise[i];
                }
        }
    InMemoryConfiguration config;
    config.set_property("GNSS-SDR.internal_fs_sps", "4000000");
    config.set_property("Tracking_1C.pll_bw_hz", "50.0");
    GpsL1CaDllPllTrackingHip conf_source(&config, "Tracking_1C", 1, 1);
    gc_ctx* ctx = nullptr;
    gc_stream* ring = nullptr;
    EXPECT(gc_ctx_create(0, &ctx) == GC_OK && gc_stream_create(ctx, GC_IQ_F32, 4000 * 128, 8000, &ring) == GC_OK, "bank: ring (%s)", gc_last_error());
    {
        std::vector<uint32_t> all;
        for (uint32_t p = 1; p <= 32; p++) all.push_back(p);
        // statistic = peak / N^4 / input power: noise cells average 1 / N, the largest of 32 x 100 x 4000 about 16 / N; 48 dB-Hz gives 63 / N
        hip_acquisition_bank bank(ctx, ring, 'G', "1C", all, 4000000, 5000, 100, 30.0f / 4000.0f);
        hip_tracking_group group(ctx, ring, conf_source.conf(), 8);
        EXPECT(bank.last_status() == GC_OK && bank.consumed_samples() == 4000 && group.last_status() == GC_OK, "bank: status %d / %d (%s)", bank.last_status(),
            group.last_status(), gc_last_error());
        std::vector<std::vector<Gnss_Synchro>> out;
        std::vector<Gnss_Synchro> detections;
        const size_t block = 4000 * 40;
        for (size_t pos = 0, b = 0; pos + block <= n; pos += block, b++)
            {
                EXPECT(gc_stream_push(ring, x.data() + pos, block, nullptr) == GC_OK, "bank: push (%s)", gc_last_error());
                if (b == 0)
                    {
                        detections = bank.search(0);
                        EXPECT(bank.last_status() == GC_OK, "bank: search status %d (%s)", bank.last_status(), gc_last_error());
                        for (size_t d = 0; d < detections.size() && d < 8; d++)
                            EXPECT(group.start_tracking(static_cast<int>(d), detections[d], detections[d].Acq_samplestamp_samples) == GC_OK, "group: hand-over (%s)",
                                gc_last_error());
                    }
                EXPECT(group.run(out) >= 0, "group: run status %d (%s)", group.last_status(), gc_last_error());
            }
        EXPECT(detections.size() == 3, "bank: %zu detections", detections.size());
        for (size_t d = 0; d < detections.size(); d++)
            {
                int k = -1;
                for (int j = 0; j < 3; j++)
                    if (static_cast<uint32_t>(prns[j]) == detections[d].PRN) k = j;
                EXPECT(k >= 0, "bank: false alarm on PRN %u", detections[d].PRN);
                if (k < 0) continue;
                EXPECT(std::fabs(detections[d].Acq_delay_samples - delays[k]) <= 1.0 && std::fabs(detections[d].Acq_doppler_hz - dopplers[k]) <= 60.0,
                    "bank: PRN %d at %.0f samples / %.0f Hz", prns[k], detections[d].Acq_delay_samples, detections[d].Acq_doppler_hz);
                const auto& o = out[d];
                double mean = 0.0;
                int cnt = 0;
                for (size_t i = o.size() > 150 ? o.size() - 150 : 0; i < o.size(); i++, cnt++) mean += o[i].Carrier_Doppler_hz;
                mean /= std::max(1, cnt);
                EXPECT(o.size() > 570 && std::fabs(mean - dopplers[k]) < 3.0 && o.back().CN0_dB_hz > 42.0, "group: PRN %d: %zu items, mean Doppler %.2f (truth %.2f), C/N0 %.1f",
                    prns[k], o.size(), mean, dopplers[k], o.empty() ? 0.0 : o.back().CN0_dB_hz);
            }
        std::printf("acquisition bank -> tracking group: 32 PRNs searched at once, %zu detections (PRN %u, %u, %u), all tracked to the end of the stream\n", detections.size(),
            detections.size() > 0 ? detections[0].PRN : 0, detections.size() > 1 ? detections[1].PRN : 0, detections.size() > 2 ? detections[2].PRN : 0);
    }
    gc_stream_destroy(ring);
    gc_ctx_destroy(ctx);
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
    test_galileo_pilot_extended();
    test_beidou_secondary_lock();
    test_gps_bit_synchronisation();
    test_gps_l5_pilot();
    test_galileo_e5a_pilot();
    test_beidou_b3i_and_gps_l2c();
    test_glonass_fdma_tracking();
    test_gps_c_aid_tracking();
    test_glonass_c_aid_tracking();
    test_device_loop_block();
    test_tracking_group(GC_IQ_F32);
    test_tracking_group(GC_IQ_I16);
    test_acquisition_bank_into_tracking_group();
    test_loss_of_lock();
    std::printf(g_fail ? "%d FAILURES\n" : "tracking self-test passed\n", g_fail);
    return g_fail ? 1 : 0;
}
