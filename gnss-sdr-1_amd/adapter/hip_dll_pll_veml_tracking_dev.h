/*!
 * \file hip_dll_pll_veml_tracking_dev.h
 * \brief The dll_pll_veml_tracking block on the DEVICE loop (gc_trk_loop): one work() call pushes the new input items into an
 * HBM ring and runs every code period that is complete in it -- correlations, synchronisation, extended integration and loop
 * maths in one launch -- then hands out one Gnss_Synchro per valid period.  GNU Radio's general_work may produce several
 * output items per call (noutput_items), so the block keeps the reference interface while the per-millisecond host round trip
 * of the CPU block (and of hip_dll_pll_veml_tracking, which calls the level-1 correlator once per period) disappears.
 *
 * Same configuration (Dll_Pll_Conf), same signals and the same per-period results as hip_dll_pll_veml_tracking /
 * dll_pll_veml_tracking (src/algorithms/tracking/gnuradio_blocks/dll_pll_veml_tracking.cc); the state machine itself lives in
 * trk_closed_loop.hip.
 */
#ifndef GNSSCORR_HIP_DLL_PLL_VEML_TRACKING_DEV_H_
#define GNSSCORR_HIP_DLL_PLL_VEML_TRACKING_DEV_H_

#include "gnss_sdr_types.h"
#include "tracking_loop_maths.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace gnsscorr
{
//! per-signal constants of the block's constructor (dll_pll_veml_tracking.cc:113-336)
struct TrkSignalConstants
{
    double carrier_freq_hz, code_period_s, code_chip_rate_hz;
    uint32_t code_length_chips, code_samples_per_chip;
    bool veml, has_pilot, interchange_iq_with_pilot;
    int32_t correlation_length_ms;
};
inline bool trk_signal_constants(char system, const std::string& signal, TrkSignalConstants* c)
{
    if (system == 'G' && signal == "1C") *c = {1575.42e6, 0.001, 1.023e6, 1023, 1, false, false, false, 1};
    else if (system == 'G' && signal == "2S") *c = {1.22760e9, 0.02, 0.5115e6, 10230, 1, false, false, false, 20};
    else if (system == 'G' && signal == "L5") *c = {1.17645e9, 0.001, 10.23e6, 10230, 1, false, true, true, 1};
    else if (system == 'E' && signal == "1B") *c = {1575.42e6, 0.004, 1.023e6, 4092, 2, true, true, false, 4};
    else if (system == 'E' && signal == "5X") *c = {1.17645e9, 0.001, 1.023e7, 10230, 1, false, true, true, 1};
    else if (system == 'C' && signal == "B1") *c = {1.561098e9, 0.001, 2.046e6, 2046, 1, false, false, false, 1};
    else if (system == 'C' && signal == "B3") *c = {1.268520e9, 0.001, 10.23e6, 10230, 1, false, false, false, 1};
    else return false;
    return true;
}

/*! start_tracking (:549-747) of one channel of a device loop: replicas, synchronisation data, loop start.  `sample_counter` is the
 *  stream position the channel starts consuming at (the block's d_sample_counter). */
inline gc_status loop_start_channel(gc_trk_loop* loop, int ch, const Dll_Pll_Conf& trk_parameters, const TrkSignalConstants& sig, const Gnss_Synchro& acq,
    uint64_t sample_counter, float bit_sync_min_time_s = 10.0f)
{
    const uint32_t prn = acq.PRN;
    const std::string signal(trk_parameters.signal);
    const int code_len = static_cast<int>(sig.code_length_chips * sig.code_samples_per_chip);
    std::vector<float> code(code_len), data_code(code_len);
    const bool pilot = trk_parameters.track_pilot && sig.has_pilot;
    gc_status st = GC_OK;
    // the replica the loop runs on, and the data component's when the loop runs on the pilot (:566-705)
    if (trk_parameters.system == 'G' && signal == "1C") st = gc_gps_l1_ca_code_gen_float(code.data(), static_cast<int32_t>(prn), 0);
    else if (trk_parameters.system == 'G' && signal == "2S") st = gc_gps_l2c_m_code_gen_float(code.data(), prn);
    else if (trk_parameters.system == 'G' && signal == "L5")
        {
            st = pilot ? gc_gps_l5q_code_gen_float(code.data(), prn) : gc_gps_l5i_code_gen_float(code.data(), prn);
            if (pilot && st == GC_OK) st = gc_gps_l5i_code_gen_float(data_code.data(), prn);
        }
    else if (trk_parameters.system == 'E' && signal == "1B")
        {
            st = gc_galileo_e1_code_gen_sinboc11_float(code.data(), pilot ? "1C" : "1B", prn);
            if (pilot && st == GC_OK) st = gc_galileo_e1_code_gen_sinboc11_float(data_code.data(), "1B", prn);
        }
    else if (trk_parameters.system == 'E' && signal == "5X")
        {
            std::vector<float> aux(2 * code_len);
            st = gc_galileo_e5_a_code_gen_complex_primary(aux.data(), static_cast<int32_t>(prn), "5X");
            for (int i = 0; i < code_len; i++)
                {
                    code[i] = pilot ? aux[2 * i + 1] : aux[2 * i];
                    data_code[i] = aux[2 * i];
                }
        }
    else if (trk_parameters.system == 'C' && signal == "B1") st = gc_beidou_b1i_code_gen_float(code.data(), static_cast<int32_t>(prn), 0);
    else st = gc_beidou_b3i_code_gen_float(code.data(), static_cast<int32_t>(prn), 0);
    if (st != GC_OK) return st;
    gc_loop_sync_conf y;
    st = gc_loop_sync_for_signal(trk_parameters.system, trk_parameters.signal, prn, pilot ? 1 : 0, std::max(1, trk_parameters.extend_correlation_symbols), &y);
    if (st != GC_OK) return st;
    y.bit_sync_min_time_s = bit_sync_min_time_s;
    y.pll_bw_narrow_hz = trk_parameters.pll_bw_narrow_hz;
    y.dll_bw_narrow_hz = trk_parameters.dll_bw_narrow_hz;
    y.early_late_space_narrow_chips = trk_parameters.early_late_space_narrow_chips;
    y.very_early_late_space_narrow_chips = trk_parameters.very_early_late_space_narrow_chips;
    st = gc_trk_loop_set_sync(loop, ch, &y, pilot ? data_code.data() : nullptr, code_len);
    if (st != GC_OK) return st;
    gc_loop_conf c;
    std::memset(&c, 0, sizeof c);
    c.fs_in = trk_parameters.fs_in;
    c.signal_carrier_freq_hz = sig.carrier_freq_hz;
    c.code_chip_rate_hz = sig.code_chip_rate_hz;
    c.code_period_s = sig.code_period_s;
    c.carrier_lock_th = trk_parameters.carrier_lock_th;
    c.acq_delay_samples = acq.Acq_delay_samples;
    c.acq_doppler_hz = acq.Acq_doppler_hz;
    c.acq_samplestamp_samples = acq.Acq_samplestamp_samples;
    c.sample_counter = sample_counter;
    c.code_length_chips = sig.code_length_chips;
    c.code_samples_per_chip = sig.code_samples_per_chip;
    c.vector_length = trk_parameters.vector_length;
    c.pull_in_time_s = trk_parameters.pull_in_time_s;
    c.veml = sig.veml ? 1 : 0;
    c.pll_filter_order = trk_parameters.pll_filter_order;
    c.dll_filter_order = trk_parameters.dll_filter_order;
    c.enable_fll_pull_in = trk_parameters.enable_fll_pull_in ? 1 : 0;
    c.enable_fll_steady_state = trk_parameters.enable_fll_steady_state ? 1 : 0;
    c.cn0_samples = trk_parameters.cn0_samples;
    c.cn0_min = trk_parameters.cn0_min;
    c.max_lock_fail = trk_parameters.max_lock_fail;
    c.pll_bw_hz = trk_parameters.pll_bw_hz;
    c.dll_bw_hz = trk_parameters.dll_bw_hz;
    c.fll_bw_hz = trk_parameters.fll_bw_hz;
    c.early_late_space_chips = trk_parameters.early_late_space_chips;
    c.very_early_late_space_chips = trk_parameters.very_early_late_space_chips;
    c.high_dyn_smoother_length = trk_parameters.high_dyn ? std::min(16u, std::max(1u, trk_parameters.smoother_length)) : 0u;
    return gc_trk_loop_start(loop, ch, &c, code.data(), code_len);
}

//! what a valid period hands to the telemetry decoder (:1693-1725, :1898-1906), from the device loop's record
inline Gnss_Synchro synchro_from_record(const gc_loop_record& r, const Gnss_Synchro& acq, const TrkSignalConstants& sig, bool interchange_iq, double fs_in)
{
    Gnss_Synchro s = acq;
    s.Prompt_I = static_cast<double>(interchange_iq ? r.prompt_data[1] : r.prompt_data[0]);
    s.Prompt_Q = static_cast<double>(interchange_iq ? r.prompt_data[0] : r.prompt_data[1]);
    s.Code_phase_samples = r.rem_code_phase_samples;
    s.Carrier_phase_rads = r.acc_carrier_phase_rad;
    s.Carrier_Doppler_hz = r.carrier_doppler_hz;
    s.CN0_dB_hz = r.cn0_db_hz;
    s.correlation_length_ms = sig.correlation_length_ms;
    s.Flag_valid_symbol_output = true;
    s.fs = static_cast<int64_t>(fs_in);
    s.Tracking_sample_counter = r.sample_counter;
    return s;
}
}  // namespace gnsscorr

class hip_dll_pll_veml_tracking_dev
{
public:
    /*! ring_periods: code periods of input the HBM ring holds (a work() call may bring at most that many) */
    explicit hip_dll_pll_veml_tracking_dev(const Dll_Pll_Conf& conf_, int device = 0, int ring_periods = 64) : trk_parameters(conf_), d_ring_periods(ring_periods)
    {
        d_ok = gnsscorr::trk_signal_constants(trk_parameters.system, std::string(trk_parameters.signal), &d_sig);
        if (!d_ok)
            {
                d_status = GC_ERR_INVALID;
                return;
            }
        if (!d_sig.has_pilot) trk_parameters.track_pilot = false;
        if (trk_parameters.extend_correlation_symbols < 1) trk_parameters.extend_correlation_symbols = 1;
        const int code_len = static_cast<int>(d_sig.code_length_chips * d_sig.code_samples_per_chip);
        d_status = gc_ctx_create(device, &d_ctx);
        if (d_status == GC_OK)
            d_status = gc_stream_create(d_ctx, GC_IQ_F32, static_cast<uint64_t>(trk_parameters.vector_length) * ring_periods, 2 * trk_parameters.vector_length, &d_ring);
        if (d_status == GC_OK) d_status = gc_trk_loop_create(d_ctx, 1, code_len, &d_loop);
        if (d_status == GC_OK) d_status = gc_trk_loop_set_input_stream(d_loop, 0, d_ring);
        d_records.resize(ring_periods + 2);
    }

    ~hip_dll_pll_veml_tracking_dev()
    {
        if (d_loop) gc_trk_loop_destroy(d_loop);
        if (d_ring) gc_stream_destroy(d_ring);
        if (d_ctx) gc_ctx_destroy(d_ctx);
    }
    hip_dll_pll_veml_tracking_dev(const hip_dll_pll_veml_tracking_dev&) = delete;
    hip_dll_pll_veml_tracking_dev& operator=(const hip_dll_pll_veml_tracking_dev&) = delete;

    void set_channel(uint32_t channel) { d_channel = channel; }
    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) { d_acquisition_gnss_synchro = p_gnss_synchro; }
    //! the reference waits 10 s before looking for the telemetry preamble (:1648); tests shorten it
    void set_bit_sync_min_time_s(float t) { d_bit_sync_min_time_s = t; }

    //! start_tracking (:549-747): replicas, synchronisation data and the loop start on the device
    void start_tracking()
    {
        std::lock_guard<std::mutex> l(d_setlock);
        if (d_status != GC_OK || !d_ok) return;
        const bool pilot = trk_parameters.track_pilot;
        d_status = gnsscorr::loop_start_channel(d_loop, 0, trk_parameters, d_sig, *d_acquisition_gnss_synchro, d_sample_counter, d_bit_sync_min_time_s);
        d_interchange_iq = pilot && d_sig.interchange_iq_with_pilot;
        d_state = d_status == GC_OK ? 1 : 0;
    }

    void stop_tracking()
    {
        std::lock_guard<std::mutex> l(d_setlock);
        d_state = 0;
    }

    //! forecast: like the reference block, at least two code periods
    int required_input_items() const { return static_cast<int>(trk_parameters.vector_length) * 2; }

    /*! general_work: `in` points at the first unconsumed item, ninput_items are available, at most max_out Gnss_Synchro fit in
     *  `out`.  Returns the items consumed; *produced = Gnss_Synchro written. */
    int work(const gr_complex* in, int ninput_items, Gnss_Synchro* out, int max_out, int* produced)
    {
        std::lock_guard<std::mutex> l(d_setlock);
        *produced = 0;
        if (d_state == 0 || d_status != GC_OK)
            {
                // standby: the items pass by (they are not needed in HBM), the counters move on
                d_sample_counter += static_cast<uint64_t>(ninput_items);
                d_pushed = std::max(d_pushed, d_sample_counter);
                return ninput_items;
            }
        // 1. the items that are not in the ring yet (the scheduler shows again what was not consumed last time)
        const uint64_t have = d_pushed - d_sample_counter;  // already pushed, not yet consumed
        uint64_t fresh = static_cast<uint64_t>(ninput_items) > have ? static_cast<uint64_t>(ninput_items) - have : 0;
        const uint64_t room = static_cast<uint64_t>(trk_parameters.vector_length) * d_ring_periods - have;
        fresh = std::min(fresh, room);
        if (fresh > 0)
            {
                if (d_ring_head != d_pushed)
                    {
                        // items went by in standby: the ring's numbering continues at the stream position (zeros fill the gap)
                        d_status = skip_to(d_pushed);
                        if (d_status != GC_OK) return 0;
                    }
                d_status = gc_stream_push(d_ring, in + have, fresh, nullptr);
                if (d_status != GC_OK) return 0;
                d_pushed += fresh;
                d_ring_head = d_pushed;
            }
        // 2. every complete code period in one launch
        const int n_periods = std::min<int>(std::min<int>(max_out, static_cast<int>(d_records.size())),
            static_cast<int>((d_pushed - d_sample_counter) / trk_parameters.vector_length) + 1);
        if (n_periods <= 0) return 0;
        d_status = gc_trk_loop_run(d_loop, n_periods, d_records.data());
        if (d_status != GC_OK) return 0;
        // 3. records -> Gnss_Synchro (:1693-1725, :1898-1906); the launch stops at the first period that is not complete
        uint64_t position = d_sample_counter;
        for (int k = 0; k < n_periods; k++)
            {
                const gc_loop_record& r = d_records[k];
                if (r.sample_counter == position && r.valid == 0 && r.state >= 2) break;  // nothing was correlated: out of input
                position = r.sample_counter;
                d_state = r.state;
                d_last = r;
                if (r.valid)
                    {
                        const Gnss_Synchro s = gnsscorr::synchro_from_record(r, *d_acquisition_gnss_synchro, d_sig, d_interchange_iq, trk_parameters.fs_in);
                        out[(*produced)++] = s;
                    }
                if (r.state == 0)
                    {
                        d_events.push_back(3);  // loss of lock (:868)
                        break;
                    }
            }
        const int consumed = static_cast<int>(position - d_sample_counter);
        d_sample_counter = position;
        return consumed;
    }

    int32_t state() const { return d_state; }
    double carrier_doppler_hz() const { return d_last.carrier_doppler_hz; }
    double code_freq_chips() const { return d_last.code_freq_chips; }
    double cn0_db_hz() const { return d_last.cn0_db_hz; }
    double carrier_lock_test() const { return d_last.carrier_lock_test; }
    uint64_t sample_counter() const { return d_sample_counter; }
    const gc_loop_record& last_record() const { return d_last; }
    const std::vector<int>& events() const { return d_events; }
    gc_status last_status() const { return d_status; }

private:
    //! appends zeros until the ring's head is at absolute sample `index`
    gc_status skip_to(uint64_t index)
    {
        std::vector<gr_complex> zeros(std::min<uint64_t>(index - d_ring_head, 1u << 16));
        while (d_ring_head < index)
            {
                const uint64_t n = std::min<uint64_t>(index - d_ring_head, zeros.size());
                gc_status s = gc_stream_push(d_ring, zeros.data(), n, nullptr);
                if (s != GC_OK) return s;
                d_ring_head += n;
            }
        return GC_OK;
    }

    Dll_Pll_Conf trk_parameters;
    gnsscorr::TrkSignalConstants d_sig{};
    bool d_ok = false, d_interchange_iq = false;
    int d_ring_periods;
    std::mutex d_setlock;
    gc_ctx* d_ctx = nullptr;
    gc_stream* d_ring = nullptr;
    gc_trk_loop* d_loop = nullptr;
    gc_status d_status = GC_OK;
    Gnss_Synchro* d_acquisition_gnss_synchro = nullptr;
    uint32_t d_channel = 0;
    int32_t d_state = 0;
    float d_bit_sync_min_time_s = 10.0f;
    uint64_t d_sample_counter = 0;  // absolute index of the first unconsumed item (the block's d_sample_counter)
    uint64_t d_pushed = 0;          // absolute index one past the last item seen
    uint64_t d_ring_head = 0;       // absolute index one past the last item in the ring
    std::vector<gc_loop_record> d_records;
    gc_loop_record d_last{};
    std::vector<int> d_events;
};

#endif  // GNSSCORR_HIP_DLL_PLL_VEML_TRACKING_DEV_H_
