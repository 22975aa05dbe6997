/*!
 * \file hip_gps_l1_ca_dll_pll_c_aid_tracking.h
 * \brief Image of gps_l1_ca_dll_pll_c_aid_tracking_cc / _sc (src/algorithms/tracking/gnuradio_blocks/
 * gps_l1_ca_dll_pll_c_aid_tracking_cc.cc, ..._sc.cc: carrier-aided DLL, optional extended coherent integration keyed to
 * the telemetry preamble time stamp) with the correlations done by Hip_Multicorrelator (gr_complex items) or
 * Hip_Multicorrelator_16sc (cshort items), plus the adapter GpsL1CaDllPllCAidTrackingHip
 * (src/algorithms/tracking/adapters/gps_l1_ca_dll_pll_c_aid_tracking.cc:47-190, `item_type` gr_complex / cshort).
 *
 * Mirrored: start_tracking (:202-278), the pull-in alignment (:597-615), the correlator history and its coherent sum on the
 * milliseconds that are a multiple of extend_correlation_ms after the preamble stamp (:626-716), the PLL with the Doppler
 * accumulator inside the loop filter, the PLL-to-DLL assistance and the second-order DLL filter (:718-767), the block-length
 * bookkeeping with integer and fractional code-phase remainders, C/N0 and lock detector, Gnss_Synchro output.  The
 * "preamble_timestamp_s" message of the telemetry decoder becomes set_preamble_timestamp_s().  Not mirrored: the dump.
 */
#ifndef GNSSCORR_HIP_GPS_L1_CA_DLL_PLL_C_AID_TRACKING_H_
#define GNSSCORR_HIP_GPS_L1_CA_DLL_PLL_C_AID_TRACKING_H_

#include "gnss_sdr_types.h"
#include "hip_glonass_ca_dll_pll_tracking.h"  // Tracking_2nd_DLL_filter
#include "hip_multicorrelator.h"
#include "hip_multicorrelator_16sc.h"
#include "tracking_loop_maths.h"
#include <cmath>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace gnsscorr
{
//! what differs between the _cc and the _sc block: the item type, the correlator class, the replica's type
template <class Item>
struct CAidItem;
template <>
struct CAidItem<gr_complex>
{
    typedef Hip_Multicorrelator correlator;
    static gr_complex from_float(gr_complex v) { return v; }
    static gr_complex to_float(gr_complex v) { return v; }
};
template <>
struct CAidItem<std::complex<int16_t>>
{
    typedef Hip_Multicorrelator_16sc correlator;
    //! volk_gnsssdr_32fc_convert_16ic: saturating conversion (the chips are +-1)
    static std::complex<int16_t> from_float(gr_complex v) { return std::complex<int16_t>(static_cast<int16_t>(v.real()), static_cast<int16_t>(v.imag())); }
    static gr_complex to_float(std::complex<int16_t> v) { return gr_complex(v.real(), v.imag()); }
};
}  // namespace gnsscorr

//! which C-aid block: GPS L1 C/A, or GLONASS L1 / L2 C/A (glonass_l1_ca_dll_pll_c_aid_tracking_cc.cc, glonass_l2_..., and their _sc
//! twins).  The GLONASS blocks differ in the constants, in the FDMA channel offset that the carrier loop filter's accumulator
//! carries (so their "d_carrier_doppler_hz" is the NCO frequency, offset included), in a code-rate aiding that follows the CHANGE
//! of that frequency (:738-741), and in a fixed 10-sample C/N0 window.
struct CAidSignal
{
    bool is_glonass;
    char system;
    double code_rate_hz, code_length_chips, carrier_freq_hz, channel_spacing_hz;
    static CAidSignal gps_l1() { return {false, 'G', 1.023e6, 1023.0, 1575.42e6, 0.0}; }
    static CAidSignal glonass(int band) { return band == 2 ? CAidSignal{true, 'R', 0.511e6, 511.0, 1.246e9, 0.4375e6} : CAidSignal{true, 'R', 0.511e6, 511.0, 1.602e9, 0.5625e6}; }
};

template <class Item>
class hip_gps_l1_ca_dll_pll_c_aid_tracking
{
public:
    hip_gps_l1_ca_dll_pll_c_aid_tracking(int64_t fs_in, uint32_t vector_length, float pll_bw_hz, float dll_bw_hz, float pll_bw_narrow_hz, float dll_bw_narrow_hz,
        int32_t extend_correlation_ms, float early_late_space_chips, int cn0_samples = 20, int cn0_min = 25, int max_lock_fail = 50, double carrier_lock_th = 0.85,
        CAidSignal signal = CAidSignal::gps_l1())
        : d_sig(signal), d_glonass_prn(gnsscorr::glonass_default_channels()), d_fs_in(fs_in), d_vector_length(vector_length), d_pll_bw_hz(pll_bw_hz), d_dll_bw_hz(dll_bw_hz), d_pll_bw_narrow_hz(pll_bw_narrow_hz),
          d_dll_bw_narrow_hz(dll_bw_narrow_hz), d_extend_correlation_ms(extend_correlation_ms), d_cn0_samples(cn0_samples), d_cn0_min(cn0_min),
          d_max_lock_fail(max_lock_fail), d_carrier_lock_threshold(carrier_lock_th)
    {
        kCodeRateHz = d_sig.code_rate_hz;
        kCodeLengthChips = d_sig.code_length_chips;
        kL1FreqHz = d_sig.carrier_freq_hz;
        if (d_sig.is_glonass) d_cn0_samples = 10;  // CN0_ESTIMATION_SAMPLES
        cn0_samples = d_cn0_samples;
        d_correlation_length_samples = static_cast<int32_t>(d_vector_length);
        d_code_loop_filter.set_DLL_BW(d_dll_bw_hz);
        d_carrier_loop_filter.set_params(10.0, d_pll_bw_hz, 2);
        d_ca_code.assign(static_cast<size_t>(kCodeLengthChips), Item());
        d_correlator_outs_item.assign(3, Item());
        d_correlator_outs.assign(3, gr_complex(0, 0));
        d_local_code_shift_chips = {-early_late_space_chips, 0.0f, early_late_space_chips};
        multicorrelator_cpu.init(2 * d_correlation_length_samples, 3);
        d_code_freq_chips = kCodeRateHz;
        d_Prompt_buffer.assign(cn0_samples, gr_complex(0, 0));
    }

    void set_channel(uint32_t channel) { d_channel = channel; }
    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) { d_acquisition_gnss_synchro = p_gnss_synchro; }
    //! GLONASS: slot -> frequency channel (almanac data; defaults to the reference's GLONASS_PRN table)
    void set_glonass_channel_map(const std::map<uint32_t, int32_t>& prn_to_channel) { d_glonass_prn = prn_to_channel; }

    //! msg_handler_preamble_index (:78-88): the telemetry decoder's preamble time stamp enables the extended integration once
    void set_preamble_timestamp_s(double t)
    {
        if (!d_enable_extended_integration)
            {
                d_preamble_timestamp_s = t;
                d_enable_extended_integration = true;
                d_preamble_synchronized = false;
            }
    }

    //! (:202-278)
    void start_tracking()
    {
        d_acq_code_phase_samples = d_acquisition_gnss_synchro->Acq_delay_samples;
        d_acq_carrier_doppler_hz = d_acquisition_gnss_synchro->Acq_doppler_hz;
        d_acq_sample_stamp = d_acquisition_gnss_synchro->Acq_samplestamp_samples;
        const int64_t acq_trk_diff_samples = static_cast<int64_t>(d_sample_counter) - static_cast<int64_t>(d_acq_sample_stamp);
        const double acq_trk_diff_seconds = static_cast<double>(acq_trk_diff_samples) / static_cast<double>(d_fs_in);
        // GLONASS: the slot's own carrier (band centre + channel x spacing) takes the place of the L1 frequency from here on
        const double channel_offset_hz = d_sig.is_glonass ? d_sig.channel_spacing_hz * static_cast<double>(d_glonass_prn.at(d_acquisition_gnss_synchro->PRN)) : 0.0;
        kL1FreqHz = d_sig.carrier_freq_hz + channel_offset_hz;
        const double radial_velocity = (kL1FreqHz + d_acq_carrier_doppler_hz) / kL1FreqHz;
        d_code_freq_chips = radial_velocity * kCodeRateHz;
        d_code_phase_step_chips = static_cast<double>(d_code_freq_chips) / static_cast<double>(d_fs_in);
        const double T_prn_mod_seconds = (1.0 / d_code_freq_chips) * kCodeLengthChips;
        const double T_prn_mod_samples = T_prn_mod_seconds * static_cast<double>(d_fs_in);
        d_correlation_length_samples = std::round(T_prn_mod_samples);
        const double T_prn_true_seconds = kCodeLengthChips / kCodeRateHz;
        const double T_prn_true_samples = T_prn_true_seconds * static_cast<double>(d_fs_in);
        const double N_prn_diff = acq_trk_diff_seconds / T_prn_true_seconds;
        double corrected = std::fmod(d_acq_code_phase_samples + (T_prn_true_seconds - T_prn_mod_seconds) * N_prn_diff * static_cast<double>(d_fs_in), T_prn_true_samples);
        if (corrected < 0) corrected = T_prn_mod_samples + corrected;
        d_acq_code_phase_samples = corrected;
        d_carrier_doppler_hz = d_acq_carrier_doppler_hz;
        const double nco_frequency_hz = d_acq_carrier_doppler_hz + channel_offset_hz;  // d_carrier_frequency_hz of the GLONASS blocks
        d_carrier_phase_step_rad = kTwoPi * nco_frequency_hz / static_cast<double>(d_fs_in);
        d_carrier_loop_filter.initialize(nco_frequency_hz);  // the carrier loop filter holds the frequency accumulator
        d_code_loop_filter.initialize();
        const int n_chips = static_cast<int>(kCodeLengthChips);
        std::vector<float> chips(n_chips);
        if (d_sig.is_glonass)
            gc_glonass_l1_ca_code_gen_float(chips.data(), 0);
        else
            gc_gps_l1_ca_code_gen_float(chips.data(), static_cast<int32_t>(d_acquisition_gnss_synchro->PRN), 0);
        for (int i = 0; i < n_chips; i++) d_ca_code[i] = gnsscorr::CAidItem<Item>::from_float(gr_complex(chips[i], 0.0f));
        multicorrelator_cpu.set_local_code_and_taps(n_chips, d_ca_code.data(), d_local_code_shift_chips.data());
        std::fill(d_correlator_outs.begin(), d_correlator_outs.end(), gr_complex(0, 0));
        d_carrier_lock_fail_counter = 0;
        d_rem_code_phase_samples = 0.0;
        d_rem_carrier_phase_rad = 0.0;
        d_rem_code_phase_chips = 0.0;
        d_acc_carrier_phase_cycles = 0.0;
        d_pll_to_dll_assist_secs_Ti = 0.0;
        d_pull_in = true;
        d_enable_tracking = true;
        d_enable_extended_integration = false;
        d_preamble_synchronized = false;
        d_E_history.clear();
        d_P_history.clear();
        d_L_history.clear();
    }

    void stop_tracking() { d_enable_tracking = false; }
    int required_input_items() const { return static_cast<int>(d_vector_length) * 2; }

    /*! general_work (:577-920): one output item per call */
    int work(const Item* in, int /*ninput_items*/, Gnss_Synchro* out, int* produced)
    {
        Gnss_Synchro current_synchro_data = Gnss_Synchro();
        double code_error_filt_secs_Ti = 0.0;
        double CURRENT_INTEGRATION_TIME_S = 0.0;
        double CORRECTED_INTEGRATION_TIME_S = 0.0;
        *produced = 1;
        if (d_enable_tracking)
            {
                current_synchro_data = *d_acquisition_gnss_synchro;
                if (d_pull_in)
                    {
                        const int32_t acq_to_trk_delay_samples = d_sample_counter - d_acq_sample_stamp;
                        const double shift_correction = d_correlation_length_samples -
                                                        std::fmod(static_cast<double>(acq_to_trk_delay_samples), static_cast<double>(d_correlation_length_samples));
                        const int32_t samples_offset = std::round(d_acq_code_phase_samples + shift_correction);
                        d_sample_counter += static_cast<uint64_t>(samples_offset);
                        current_synchro_data.Tracking_sample_counter = d_sample_counter;
                        d_pull_in = false;
                        d_acc_carrier_phase_cycles -= d_carrier_phase_step_rad * samples_offset / kTwoPi;
                        current_synchro_data.Carrier_phase_rads = d_acc_carrier_phase_cycles * kTwoPi;
                        current_synchro_data.Carrier_Doppler_hz = d_carrier_doppler_hz;
                        current_synchro_data.fs = d_fs_in;
                        *out = current_synchro_data;
                        return samples_offset;
                    }
                // the hot path: one launch of the HIP multicorrelator (complex or 16-bit chips)
                multicorrelator_cpu.set_input_output_vectors(d_correlator_outs_item.data(), in);
                multicorrelator_cpu.Carrier_wipeoff_multicorrelator_resampler(d_rem_carrier_phase_rad, d_carrier_phase_step_rad, d_rem_code_phase_chips,
                    d_code_phase_step_chips, d_correlation_length_samples);
                // the last extend_correlation_ms outputs, in the items' own arithmetic
                d_E_history.push_back(d_correlator_outs_item[0]);
                d_P_history.push_back(d_correlator_outs_item[1]);
                d_L_history.push_back(d_correlator_outs_item[2]);
                if (static_cast<int32_t>(d_P_history.size()) > d_extend_correlation_ms)
                    {
                        d_E_history.pop_front();
                        d_P_history.pop_front();
                        d_L_history.pop_front();
                    }
                bool enable_dll_pll;
                if (d_enable_extended_integration)
                    {
                        const int64_t symbol_diff = std::round(
                            1000.0 * ((static_cast<double>(d_sample_counter) + d_rem_code_phase_samples) / static_cast<double>(d_fs_in) - d_preamble_timestamp_s));
                        if (symbol_diff > 0 and symbol_diff % d_extend_correlation_ms == 0)
                            {
                                // coherent sum of the history, then a loop update
                                Item e = Item(), p = Item(), l = Item();
                                for (int32_t n = 0; n < d_extend_correlation_ms && n < static_cast<int32_t>(d_P_history.size()); n++)
                                    {
                                        e += d_E_history.at(n);
                                        p += d_P_history.at(n);
                                        l += d_L_history.at(n);
                                    }
                                d_correlator_outs_item[0] = e;
                                d_correlator_outs_item[1] = p;
                                d_correlator_outs_item[2] = l;
                                if (!d_preamble_synchronized)
                                    {
                                        d_code_loop_filter.set_DLL_BW(d_dll_bw_narrow_hz);
                                        d_carrier_loop_filter.set_params(10.0, d_pll_bw_narrow_hz, 2);
                                        d_preamble_synchronized = true;
                                    }
                                CURRENT_INTEGRATION_TIME_S = static_cast<double>(d_extend_correlation_ms) * 0.001;
                                d_code_loop_filter.set_pdi(CURRENT_INTEGRATION_TIME_S);
                                enable_dll_pll = true;
                            }
                        else if (d_preamble_synchronized)
                            {
                                // inside an extended integration: advance the NCOs by one code, no loop update (:676-700)
                                const double T_prn_samples = (1.0 / d_code_freq_chips) * kCodeLengthChips * static_cast<double>(d_fs_in);
                                const int32_t K_prn_samples = std::round(T_prn_samples);
                                d_rem_code_phase_samples = d_rem_code_phase_samples - (K_prn_samples - T_prn_samples);
                                d_rem_code_phase_integer_samples = std::round(d_rem_code_phase_samples);
                                d_correlation_length_samples = K_prn_samples + d_rem_code_phase_integer_samples;
                                d_rem_code_phase_samples = d_rem_code_phase_samples - d_rem_code_phase_integer_samples;
                                d_code_phase_step_chips = d_code_freq_chips / static_cast<double>(d_fs_in);
                                d_rem_code_phase_chips = d_rem_code_phase_samples * (d_code_freq_chips / static_cast<double>(d_fs_in));
                                d_rem_carrier_phase_rad = std::fmod(d_rem_carrier_phase_rad + d_carrier_phase_step_rad * static_cast<double>(d_correlation_length_samples), kTwoPi);
                                d_acc_carrier_phase_cycles -= d_carrier_phase_step_rad * d_correlation_length_samples / kTwoPi;
                                enable_dll_pll = false;
                            }
                        else
                            {
                                CURRENT_INTEGRATION_TIME_S = static_cast<double>(d_correlation_length_samples) / static_cast<double>(d_fs_in);
                                d_code_loop_filter.set_pdi(CURRENT_INTEGRATION_TIME_S);
                                enable_dll_pll = true;
                            }
                    }
                else
                    {
                        CURRENT_INTEGRATION_TIME_S = static_cast<double>(d_correlation_length_samples) / static_cast<double>(d_fs_in);
                        enable_dll_pll = true;
                    }
                for (int t = 0; t < 3; t++) d_correlator_outs[t] = gnsscorr::CAidItem<Item>::to_float(d_correlator_outs_item[t]);
                if (enable_dll_pll)
                    {
                        // PLL: the filter output IS the Doppler (accumulator inside, Kaplan)
                        d_carr_phase_error_secs_Ti = pll_cloop_two_quadrant_atan(d_correlator_outs[1]) / kTwoPi;
                        const double carrier_doppler_old_hz = d_carrier_doppler_hz;
                        d_carrier_doppler_hz = d_carrier_loop_filter.get_carrier_error(0.0, d_carr_phase_error_secs_Ti, CURRENT_INTEGRATION_TIME_S);
                        d_pll_to_dll_assist_secs_Ti = (d_carrier_doppler_hz * CURRENT_INTEGRATION_TIME_S) / kL1FreqHz;
                        // GLONASS blocks: the accumulator holds Doppler + channel offset, and the code rate follows its CHANGE (:738-741)
                        const double aiding_hz = d_sig.is_glonass ? d_carrier_doppler_hz - carrier_doppler_old_hz : d_carrier_doppler_hz;
                        d_code_freq_chips = kCodeRateHz + ((aiding_hz * kCodeRateHz) / kL1FreqHz);
                        // DLL
                        d_code_error_chips_Ti = dll_nc_e_minus_l_normalized(d_correlator_outs[0], d_correlator_outs[2]);
                        d_code_error_filt_chips_s = d_code_loop_filter.get_code_nco(d_code_error_chips_Ti);
                        d_code_error_filt_chips_Ti = d_code_error_filt_chips_s * CURRENT_INTEGRATION_TIME_S;
                        code_error_filt_secs_Ti = d_code_error_filt_chips_Ti / d_code_freq_chips;
                        // next block: one code, plus the whole samples of the accumulated code-phase remainder
                        const double T_prn_samples = (1.0 / d_code_freq_chips) * kCodeLengthChips * static_cast<double>(d_fs_in);
                        const double K_prn_samples = std::round(T_prn_samples);
                        d_rem_code_phase_samples = d_rem_code_phase_samples - (K_prn_samples - T_prn_samples) + code_error_filt_secs_Ti * static_cast<double>(d_fs_in);
                        d_rem_code_phase_integer_samples = std::round(d_rem_code_phase_samples);
                        d_correlation_length_samples = K_prn_samples + d_rem_code_phase_integer_samples;
                        d_rem_code_phase_samples = d_rem_code_phase_samples - d_rem_code_phase_integer_samples;
                        d_carrier_phase_step_rad = kTwoPi * d_carrier_doppler_hz / static_cast<double>(d_fs_in);
                        d_acc_carrier_phase_cycles -= d_carrier_phase_step_rad * d_correlation_length_samples / kTwoPi;
                        CORRECTED_INTEGRATION_TIME_S = (static_cast<double>(d_correlation_length_samples) / static_cast<double>(d_fs_in));
                        d_rem_carrier_phase_rad = std::fmod(d_rem_carrier_phase_rad + kTwoPi * d_carrier_doppler_hz * CORRECTED_INTEGRATION_TIME_S, kTwoPi);
                        d_code_phase_step_chips = d_code_freq_chips / static_cast<double>(d_fs_in);
                        d_rem_code_phase_chips = d_rem_code_phase_samples * (d_code_freq_chips / static_cast<double>(d_fs_in));
                        // C/N0 and lock detector
                        if (d_cn0_estimation_counter < d_cn0_samples)
                            {
                                d_Prompt_buffer[d_cn0_estimation_counter] = d_correlator_outs[1];
                                d_cn0_estimation_counter++;
                            }
                        else
                            {
                                d_cn0_estimation_counter = 0;
                                d_CN0_SNV_dB_Hz = cn0_svn_estimator(d_Prompt_buffer.data(), d_cn0_samples, 0.001);  // both code periods are 1 ms
                                d_carrier_lock_test = carrier_lock_detector(d_Prompt_buffer.data(), d_cn0_samples);
                                if (d_carrier_lock_test < d_carrier_lock_threshold or d_CN0_SNV_dB_Hz < d_cn0_min)
                                    d_carrier_lock_fail_counter++;
                                else if (d_carrier_lock_fail_counter > 0)
                                    d_carrier_lock_fail_counter--;
                                if (d_carrier_lock_fail_counter > d_max_lock_fail)
                                    {
                                        d_events.push_back(3);  // 3 -> loss of lock
                                        d_carrier_lock_fail_counter = 0;
                                        d_enable_tracking = false;
                                    }
                            }
                        current_synchro_data.Flag_valid_symbol_output = true;
                        current_synchro_data.correlation_length_ms = d_preamble_synchronized ? d_extend_correlation_ms : 1;
                    }
                current_synchro_data.Prompt_I = static_cast<double>(d_correlator_outs[1].real());
                current_synchro_data.Prompt_Q = static_cast<double>(d_correlator_outs[1].imag());
                current_synchro_data.Tracking_sample_counter = d_sample_counter + static_cast<uint64_t>(d_correlation_length_samples);
                current_synchro_data.Code_phase_samples = d_rem_code_phase_samples;
                current_synchro_data.Carrier_phase_rads = kTwoPi * d_acc_carrier_phase_cycles;
                current_synchro_data.Carrier_Doppler_hz = d_carrier_doppler_hz;
                current_synchro_data.CN0_dB_hz = d_CN0_SNV_dB_Hz;
            }
        else
            {
                std::fill(d_correlator_outs.begin(), d_correlator_outs.end(), gr_complex(0, 0));
                current_synchro_data.System = d_sig.system;
                current_synchro_data.Tracking_sample_counter = d_sample_counter + static_cast<uint64_t>(d_correlation_length_samples);
            }
        current_synchro_data.fs = d_fs_in;
        *out = current_synchro_data;
        d_sample_counter += d_correlation_length_samples;
        return d_correlation_length_samples;
    }

    bool tracking_enabled() const { return d_enable_tracking; }
    bool preamble_synchronized() const { return d_preamble_synchronized; }
    double carrier_doppler_hz() const { return d_carrier_doppler_hz; }
    double code_freq_chips() const { return d_code_freq_chips; }
    double cn0_db_hz() const { return d_CN0_SNV_dB_Hz; }
    double carrier_lock_test() const { return d_carrier_lock_test; }
    double rem_code_phase_samples() const { return d_rem_code_phase_samples; }
    uint64_t sample_counter() const { return d_sample_counter; }
    const std::vector<gr_complex>& correlator_outs() const { return d_correlator_outs; }
    const std::vector<int>& events() const { return d_events; }
    gc_status last_status() const { return multicorrelator_cpu.last_status(); }

private:
    static constexpr double kTwoPi = 6.283185307179586;  // GPS_TWO_PI / GLONASS_TWO_PI
    double kCodeRateHz = 1.023e6, kCodeLengthChips = 1023.0;
    double kL1FreqHz = 1575.42e6;  // the carrier the code rate is aided against (GLONASS: the slot's own, set in start_tracking)

    CAidSignal d_sig;
    std::map<uint32_t, int32_t> d_glonass_prn;
    int64_t d_fs_in;
    uint32_t d_vector_length;
    float d_pll_bw_hz, d_dll_bw_hz, d_pll_bw_narrow_hz, d_dll_bw_narrow_hz;
    int32_t d_extend_correlation_ms;
    int d_cn0_samples, d_cn0_min, d_max_lock_fail;
    double d_carrier_lock_threshold;
    uint32_t d_channel = 0;
    Gnss_Synchro* d_acquisition_gnss_synchro = nullptr;
    typename gnsscorr::CAidItem<Item>::correlator multicorrelator_cpu;
    std::vector<Item> d_ca_code, d_correlator_outs_item;
    std::vector<gr_complex> d_correlator_outs, d_Prompt_buffer;
    std::vector<float> d_local_code_shift_chips;
    std::deque<Item> d_E_history, d_P_history, d_L_history;
    Tracking_2nd_DLL_filter d_code_loop_filter;
    Tracking_FLL_PLL_filter d_carrier_loop_filter;
    double d_acq_code_phase_samples = 0.0, d_acq_carrier_doppler_hz = 0.0;
    uint64_t d_acq_sample_stamp = 0, d_sample_counter = 0;
    double d_code_freq_chips = 0.0, d_code_phase_step_chips = 0.0, d_carrier_doppler_hz = 0.0, d_carrier_phase_step_rad = 0.0;
    double d_rem_code_phase_samples = 0.0, d_rem_code_phase_chips = 0.0, d_rem_carrier_phase_rad = 0.0, d_acc_carrier_phase_cycles = 0.0;
    int32_t d_rem_code_phase_integer_samples = 0;
    double d_pll_to_dll_assist_secs_Ti = 0.0, d_carr_phase_error_secs_Ti = 0.0;
    double d_code_error_chips_Ti = 0.0, d_code_error_filt_chips_s = 0.0, d_code_error_filt_chips_Ti = 0.0;
    int32_t d_correlation_length_samples = 0;
    int32_t d_cn0_estimation_counter = 0, d_carrier_lock_fail_counter = 0;
    double d_carrier_lock_test = 1.0, d_CN0_SNV_dB_Hz = 0.0;
    double d_preamble_timestamp_s = 0.0;
    bool d_enable_tracking = false, d_pull_in = false, d_enable_extended_integration = false, d_preamble_synchronized = false;
    std::vector<int> d_events;
};

//! TrackingInterface adapter; item_type "gr_complex" or "cshort" picks the block (gps_l1_ca_dll_pll_c_aid_tracking.cc:62-125,
//! glonass_l1_ca_dll_pll_c_aid_tracking.cc, glonass_l2_ca_dll_pll_c_aid_tracking.cc).  BAND: 0 = GPS L1 C/A, 1 / 2 = GLONASS L1 / L2 C/A
template <int BAND>
class DllPllCAidTrackingHip : public TrackingInterface
{
public:
    typedef hip_gps_l1_ca_dll_pll_c_aid_tracking<gr_complex> block_cc;
    typedef hip_gps_l1_ca_dll_pll_c_aid_tracking<std::complex<int16_t>> block_sc;

    DllPllCAidTrackingHip(ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
        : role_(role), in_streams_(in_streams), out_streams_(out_streams)
    {
        item_type_ = configuration->property(role + ".item_type", std::string("gr_complex"));
        int fs_in_deprecated = configuration->property("GNSS-SDR.internal_fs_hz", 2048000);
        const int fs_in = configuration->property("GNSS-SDR.internal_fs_sps", fs_in_deprecated);
        const float pll_bw_hz = configuration->property(role + ".pll_bw_hz", 50.0f);
        const float dll_bw_hz = configuration->property(role + ".dll_bw_hz", 2.0f);
        const float pll_bw_narrow_hz = configuration->property(role + ".pll_bw_narrow_hz", 20.0f);
        const float dll_bw_narrow_hz = configuration->property(role + ".dll_bw_narrow_hz", 2.0f);
        const int extend_correlation_ms = configuration->property(role + ".extend_correlation_ms", 1);
        const float early_late_space_chips = configuration->property(role + ".early_late_space_chips", 0.5f);
        const CAidSignal sig = BAND == 0 ? CAidSignal::gps_l1() : CAidSignal::glonass(BAND);
        vector_length_ = std::round(fs_in / (sig.code_rate_hz / sig.code_length_chips));
        const int cn0_samples = configuration->property(role + ".cn0_samples", 20), cn0_min = configuration->property(role + ".cn0_min", 25);
        const int max_lock_fail = configuration->property(role + ".max_lock_fail", 50);
        const double lock_th = configuration->property(role + ".carrier_lock_th", 0.85);
        if (item_type_ == "cshort")
            tracking_sc_ = std::make_shared<block_sc>(fs_in, vector_length_, pll_bw_hz, dll_bw_hz, pll_bw_narrow_hz, dll_bw_narrow_hz, extend_correlation_ms,
                early_late_space_chips, cn0_samples, cn0_min, max_lock_fail, lock_th, sig);
        else
            {
                item_type_ = "gr_complex";
                tracking_cc_ = std::make_shared<block_cc>(fs_in, vector_length_, pll_bw_hz, dll_bw_hz, pll_bw_narrow_hz, dll_bw_narrow_hz, extend_correlation_ms,
                    early_late_space_chips, cn0_samples, cn0_min, max_lock_fail, lock_th, sig);
            }
    }

    std::string role() override { return role_; }
    std::string implementation() override
    {
        return BAND == 0 ? "GPS_L1_CA_DLL_PLL_C_Aid_Tracking_HIP" : BAND == 1 ? "GLONASS_L1_CA_DLL_PLL_C_Aid_Tracking_HIP" : "GLONASS_L2_CA_DLL_PLL_C_Aid_Tracking_HIP";
    }
    size_t item_size() override { return item_type_ == "cshort" ? sizeof(std::complex<int16_t>) : sizeof(gr_complex); }
    void start_tracking() override { tracking_cc_ ? tracking_cc_->start_tracking() : tracking_sc_->start_tracking(); }
    void stop_tracking() override { tracking_cc_ ? tracking_cc_->stop_tracking() : tracking_sc_->stop_tracking(); }
    void set_channel(unsigned int channel) override { tracking_cc_ ? tracking_cc_->set_channel(channel) : tracking_sc_->set_channel(channel); }
    void set_gnss_synchro(Gnss_Synchro* p) override { tracking_cc_ ? tracking_cc_->set_gnss_synchro(p) : tracking_sc_->set_gnss_synchro(p); }
    std::shared_ptr<block_cc> block_gr_complex() { return tracking_cc_; }
    std::shared_ptr<block_sc> block_cshort() { return tracking_sc_; }
    unsigned int vector_length() const { return vector_length_; }
    const std::string& item_type() const { return item_type_; }

private:
    std::shared_ptr<block_cc> tracking_cc_;
    std::shared_ptr<block_sc> tracking_sc_;
    std::string role_, item_type_;
    unsigned int in_streams_, out_streams_;
    unsigned int vector_length_ = 0;
};

using GpsL1CaDllPllCAidTrackingHip = DllPllCAidTrackingHip<0>;
using GlonassL1CaDllPllCAidTrackingHip = DllPllCAidTrackingHip<1>;
using GlonassL2CaDllPllCAidTrackingHip = DllPllCAidTrackingHip<2>;

#endif  // GNSSCORR_HIP_GPS_L1_CA_DLL_PLL_C_AID_TRACKING_H_
