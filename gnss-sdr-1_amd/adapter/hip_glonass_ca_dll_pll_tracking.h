/*!
 * \file hip_glonass_ca_dll_pll_tracking.h
 * \brief Image of Glonass_L1_Ca_Dll_Pll_Tracking_cc / Glonass_L2_Ca_Dll_Pll_Tracking_cc
 * (src/algorithms/tracking/gnuradio_blocks/glonass_l1_ca_dll_pll_tracking_cc.{h,cc}; the L2 block differs in the carrier
 * constants only) with the correlations done by Hip_Multicorrelator (complex chips, libgnsscorr.so, MI355X), plus the
 * TrackingInterface adapters GlonassL1CaDllPllTrackingHip / GlonassL2CaDllPllTrackingHip
 * (src/algorithms/tracking/adapters/glonass_l1_ca_dll_pll_tracking.cc:46-130).
 *
 * Mirrored: start_tracking (:160-243) with the FDMA frequency channel of the slot (GLONASS_PRN, GLONASS_L1_L2_CA.h:129) in the
 * carrier NCO and in the code-rate aiding, the pull-in alignment (:570-590), the per-period loop of general_work (:592-693):
 * 5-argument Carrier_wipeoff_multicorrelator_resampler, two-quadrant PLL and normalised E-L DLL discriminators, the
 * second-order loop filters (tracking_2nd_PLL_filter.cc:40-88, tracking_2nd_DLL_filter.cc:40-76), the round()-ed block
 * length with the code NCO command folded in (:617-629), C/N0 and lock detector, Gnss_Synchro output.
 * Not mirrored: the dump file and its .mat conversion.
 *
 * general_work becomes work(in, ninput_items, out, produced) like hip_dll_pll_veml_tracking::work.
 */
#ifndef GNSSCORR_HIP_GLONASS_CA_DLL_PLL_TRACKING_H_
#define GNSSCORR_HIP_GLONASS_CA_DLL_PLL_TRACKING_H_

#include "gnss_sdr_types.h"
#include "hip_multicorrelator.h"
#include "tracking_loop_maths.h"
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>

//! Tracking_2nd_DLL_filter (src/algorithms/tracking/libs/tracking_2nd_DLL_filter.cc:40-96)
class Tracking_2nd_DLL_filter
{
public:
    explicit Tracking_2nd_DLL_filter(float pdi_code = 0.001f) : d_pdi_code(pdi_code) {}
    void set_DLL_BW(float dll_bw_hz)
    {
        d_dllnoisebandwidth = dll_bw_hz;
        loop_coefficients(&d_tau1_code, &d_tau2_code, d_dllnoisebandwidth, d_dlldampingratio, 1.0f);
    }
    void set_pdi(float pdi_code) { d_pdi_code = pdi_code; }
    void initialize()
    {
        d_old_code_nco = 0.0f;
        d_old_code_error = 0.0f;
    }
    float get_code_nco(float DLL_discriminator)
    {
        float code_nco = d_old_code_nco + (d_tau2_code / d_tau1_code) * (DLL_discriminator - d_old_code_error) +
                         (DLL_discriminator + d_old_code_error) * (d_pdi_code / (2.0 * d_tau1_code));
        d_old_code_nco = code_nco;
        d_old_code_error = DLL_discriminator;
        return code_nco;
    }
    //! natural frequency from the noise bandwidth, then the two time constants (Kaplan)
    static void loop_coefficients(float* tau1, float* tau2, float lbw, float zeta, float k)
    {
        float Wn = lbw * 8.0 * zeta / (4.0 * zeta * zeta + 1.0);
        *tau1 = k / (Wn * Wn);
        *tau2 = 2.0 * zeta / Wn;
    }

private:
    float d_tau1_code = 0.0f, d_tau2_code = 0.0f, d_pdi_code, d_dllnoisebandwidth = 0.0f, d_dlldampingratio = 0.7f;
    float d_old_code_error = 0.0f, d_old_code_nco = 0.0f;
};

//! Tracking_2nd_PLL_filter (src/algorithms/tracking/libs/tracking_2nd_PLL_filter.cc:40-104)
class Tracking_2nd_PLL_filter
{
public:
    explicit Tracking_2nd_PLL_filter(float pdi_carr = 0.001f) : d_pdi_carr(pdi_carr) {}
    void set_PLL_BW(float pll_bw_hz)
    {
        d_pllnoisebandwidth = pll_bw_hz;
        Tracking_2nd_DLL_filter::loop_coefficients(&d_tau1_carr, &d_tau2_carr, d_pllnoisebandwidth, d_plldampingratio, 0.25f);
    }
    void set_pdi(float pdi_carr) { d_pdi_carr = pdi_carr; }
    void initialize()
    {
        d_old_carr_nco = 0.0f;
        d_old_carr_error = 0.0f;
    }
    float get_carrier_nco(float PLL_discriminator)
    {
        float carr_nco = d_old_carr_nco + (d_tau2_carr / d_tau1_carr) * (PLL_discriminator - d_old_carr_error) +
                         (PLL_discriminator + d_old_carr_error) * (d_pdi_carr / (2.0 * d_tau1_carr));
        d_old_carr_nco = carr_nco;
        d_old_carr_error = PLL_discriminator;
        return carr_nco;
    }

private:
    float d_tau1_carr = 0.0f, d_tau2_carr = 0.0f, d_pdi_carr, d_pllnoisebandwidth = 0.0f, d_plldampingratio = 0.7f;
    float d_old_carr_error = 0.0f, d_old_carr_nco = 0.0f;
};

namespace gnsscorr
{
//! GLONASS_L1_L2_CA.h:84-97
struct GlonassBand
{
    double freq_hz, dfreq_hz;
    const char* implementation;
};
inline GlonassBand glonass_band(int band)
{
    return band == 2 ? GlonassBand{1.246e9, 0.4375e6, "GLONASS_L2_CA_DLL_PLL_Tracking_HIP"} : GlonassBand{1.602e9, 0.5625e6, "GLONASS_L1_CA_DLL_PLL_Tracking_HIP"};
}
//! GLONASS_PRN (GLONASS_L1_L2_CA.h:129-154): slot -> frequency channel of the constellation the reference was written against
inline const std::map<uint32_t, int32_t>& glonass_default_channels()
{
    static const std::map<uint32_t, int32_t> m = {{0, 8}, {1, 1}, {2, -4}, {3, 5}, {4, 6}, {5, 1}, {6, -4}, {7, 5}, {8, 6}, {9, -2}, {10, -7}, {11, 0}, {12, -1},
        {13, -2}, {14, -7}, {15, 0}, {16, -1}, {17, 4}, {18, -3}, {19, 3}, {20, -5}, {21, 4}, {22, -3}, {23, 3}, {24, 2}};
    return m;
}
}  // namespace gnsscorr

class hip_glonass_ca_dll_pll_tracking
{
public:
    hip_glonass_ca_dll_pll_tracking(int band, int64_t fs_in, uint32_t vector_length, float pll_bw_hz, float dll_bw_hz, float early_late_space_chips, int cn0_samples = 20,
        int cn0_min = 25, int max_lock_fail = 50, double carrier_lock_th = 0.85)
        : d_band(gnsscorr::glonass_band(band)), d_fs_in(fs_in), d_vector_length(vector_length), d_cn0_samples(cn0_samples), d_cn0_min(cn0_min), d_max_lock_fail(max_lock_fail),
          d_carrier_lock_threshold(carrier_lock_th), d_glonass_prn(gnsscorr::glonass_default_channels())
    {
        d_current_prn_length_samples = static_cast<int32_t>(d_vector_length);
        d_code_loop_filter.set_DLL_BW(dll_bw_hz);
        d_carrier_loop_filter.set_PLL_BW(pll_bw_hz);
        d_ca_code.assign(511, gr_complex(0, 0));
        d_correlator_outs.assign(3, gr_complex(0, 0));
        d_local_code_shift_chips = {-early_late_space_chips, 0.0f, early_late_space_chips};
        multicorrelator_cpu.init(2 * d_current_prn_length_samples, 3);
        d_code_freq_chips = kCodeRateHz;
        d_Prompt_buffer.assign(cn0_samples, gr_complex(0, 0));
    }

    void set_channel(uint32_t channel) { d_channel = channel; }
    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) { d_acquisition_gnss_synchro = p_gnss_synchro; }
    //! the slot -> frequency channel map comes from the almanac in a live receiver
    void set_glonass_channel_map(const std::map<uint32_t, int32_t>& prn_to_channel) { d_glonass_prn = prn_to_channel; }

    //! (:160-243)
    void start_tracking()
    {
        d_acq_code_phase_samples = d_acquisition_gnss_synchro->Acq_delay_samples;
        d_acq_carrier_doppler_hz = d_acquisition_gnss_synchro->Acq_doppler_hz;
        d_acq_sample_stamp = d_acquisition_gnss_synchro->Acq_samplestamp_samples;
        const int64_t acq_trk_diff_samples = static_cast<int64_t>(d_sample_counter) - static_cast<int64_t>(d_acq_sample_stamp);
        const double acq_trk_diff_seconds = static_cast<float>(acq_trk_diff_samples) / static_cast<float>(d_fs_in);
        const double channel_offset_hz = d_band.dfreq_hz * d_glonass_prn.at(d_acquisition_gnss_synchro->PRN);
        d_glonass_freq_ch = d_band.freq_hz + channel_offset_hz;
        // code-rate aiding from the acquisition Doppler, against the slot's own carrier
        const double radial_velocity = (d_glonass_freq_ch + d_acq_carrier_doppler_hz) / d_glonass_freq_ch;
        d_code_freq_chips = radial_velocity * kCodeRateHz;
        d_code_phase_step_chips = static_cast<double>(d_code_freq_chips) / static_cast<double>(d_fs_in);
        const double T_prn_mod_seconds = (1 / d_code_freq_chips) * kCodeLengthChips;
        const double T_prn_mod_samples = T_prn_mod_seconds * static_cast<double>(d_fs_in);
        d_current_prn_length_samples = std::round(T_prn_mod_samples);
        // code phase drift between the acquisition stamp and now
        const double T_prn_true_seconds = kCodeLengthChips / kCodeRateHz;
        const double T_prn_true_samples = T_prn_true_seconds * static_cast<double>(d_fs_in);
        const double N_prn_diff = acq_trk_diff_seconds / T_prn_true_seconds;
        double corrected = std::fmod(d_acq_code_phase_samples + (T_prn_true_seconds - T_prn_mod_seconds) * N_prn_diff * static_cast<double>(d_fs_in), T_prn_true_samples);
        if (corrected < 0) corrected = T_prn_mod_samples + corrected;
        d_acq_code_phase_samples = corrected;
        // the carrier NCO carries the FDMA offset, the Doppler bookkeeping does not
        d_carrier_frequency_hz = d_acq_carrier_doppler_hz + channel_offset_hz;
        d_carrier_doppler_hz = d_acq_carrier_doppler_hz;
        d_carrier_phase_step_rad = kTwoPi * d_carrier_frequency_hz / static_cast<double>(d_fs_in);
        d_carrier_doppler_phase_step_rad = kTwoPi * d_carrier_doppler_hz / static_cast<double>(d_fs_in);
        d_carrier_loop_filter.initialize();
        d_code_loop_filter.initialize();
        std::vector<float> chips(511);
        gc_glonass_l1_ca_code_gen_float(chips.data(), 0);  // glonass_l1_ca_code_gen_complex: (+-1, 0)
        for (int i = 0; i < 511; i++) d_ca_code[i] = gr_complex(chips[i], 0.0f);
        multicorrelator_cpu.set_local_code_and_taps(511, d_ca_code.data(), d_local_code_shift_chips.data());
        std::fill(d_correlator_outs.begin(), d_correlator_outs.end(), gr_complex(0, 0));
        d_carrier_lock_fail_counter = 0;
        d_rem_code_phase_samples = 0;
        d_rem_carr_phase_rad = 0.0;
        d_rem_code_phase_chips = 0.0;
        d_acc_carrier_phase_rad = 0.0;
        d_pull_in = true;
        d_enable_tracking = true;
    }

    void stop_tracking() { d_enable_tracking = false; }
    int required_input_items() const { return static_cast<int>(d_vector_length) * 2; }

    /*! general_work (:549-777): one output item per call, consumes the period just correlated */
    int work(const gr_complex* in, int /*ninput_items*/, Gnss_Synchro* out, int* produced)
    {
        Gnss_Synchro current_synchro_data = Gnss_Synchro();
        *produced = 1;
        if (d_enable_tracking)
            {
                current_synchro_data = *d_acquisition_gnss_synchro;
                if (d_pull_in)
                    {
                        const int32_t acq_to_trk_delay_samples = d_sample_counter - d_acq_sample_stamp;
                        const double shift_correction = d_current_prn_length_samples -
                                                        std::fmod(static_cast<float>(acq_to_trk_delay_samples), static_cast<float>(d_current_prn_length_samples));
                        const int32_t samples_offset = std::round(d_acq_code_phase_samples + shift_correction);
                        d_sample_counter = d_sample_counter + static_cast<uint64_t>(samples_offset);
                        current_synchro_data.Tracking_sample_counter = d_sample_counter;
                        d_pull_in = false;
                        d_acc_carrier_phase_rad -= d_carrier_doppler_phase_step_rad * samples_offset;
                        current_synchro_data.Carrier_phase_rads = d_acc_carrier_phase_rad;
                        current_synchro_data.Carrier_Doppler_hz = d_carrier_doppler_hz;
                        current_synchro_data.fs = d_fs_in;
                        current_synchro_data.correlation_length_ms = 1;
                        *out = current_synchro_data;
                        return samples_offset;
                    }
                // the hot path: one launch of the HIP complex-chip multicorrelator
                multicorrelator_cpu.set_input_output_vectors(d_correlator_outs.data(), in);
                multicorrelator_cpu.Carrier_wipeoff_multicorrelator_resampler(d_rem_carr_phase_rad, d_carrier_phase_step_rad, d_rem_code_phase_chips,
                    d_code_phase_step_chips, d_current_prn_length_samples);
                // PLL
                const double carr_error_hz = pll_cloop_two_quadrant_atan(d_correlator_outs[1]) / kTwoPi;
                const double carr_error_filt_hz = d_carrier_loop_filter.get_carrier_nco(carr_error_hz);
                d_carrier_frequency_hz += carr_error_filt_hz;
                d_carrier_doppler_hz += carr_error_filt_hz;
                d_code_freq_chips = kCodeRateHz + ((d_carrier_doppler_hz * kCodeRateHz) / d_glonass_freq_ch);
                // DLL: the filtered code error becomes a time shift of the next block
                const double code_error_chips = dll_nc_e_minus_l_normalized(d_correlator_outs[0], d_correlator_outs[2]);
                const double code_error_filt_chips = d_code_loop_filter.get_code_nco(code_error_chips);
                const double T_chip_seconds = 1.0 / static_cast<double>(d_code_freq_chips);
                const double T_prn_seconds = T_chip_seconds * kCodeLengthChips;
                const double code_error_filt_secs = (T_prn_seconds * code_error_filt_chips * T_chip_seconds);
                const double T_prn_samples = T_prn_seconds * static_cast<double>(d_fs_in);
                const double K_blk_samples = T_prn_samples + d_rem_code_phase_samples + code_error_filt_secs * static_cast<double>(d_fs_in);
                d_current_prn_length_samples = std::round(K_blk_samples);
                // NCO commands for the next block
                d_carrier_doppler_phase_step_rad = kTwoPi * d_carrier_doppler_hz / static_cast<double>(d_fs_in);
                d_carrier_phase_step_rad = kTwoPi * d_carrier_frequency_hz / static_cast<double>(d_fs_in);
                d_rem_carr_phase_rad = d_rem_carr_phase_rad + d_carrier_phase_step_rad * d_current_prn_length_samples;
                d_rem_carr_phase_rad = std::fmod(d_rem_carr_phase_rad, kTwoPi);
                d_acc_carrier_phase_rad -= d_carrier_doppler_phase_step_rad * d_current_prn_length_samples;
                d_code_phase_step_chips = d_code_freq_chips / static_cast<double>(d_fs_in);
                d_rem_code_phase_samples = K_blk_samples - d_current_prn_length_samples;
                d_rem_code_phase_chips = d_code_freq_chips * (d_rem_code_phase_samples / static_cast<double>(d_fs_in));
                // C/N0 and lock detector
                if (d_cn0_estimation_counter < d_cn0_samples)
                    {
                        d_Prompt_buffer[d_cn0_estimation_counter] = d_correlator_outs[1];
                        d_cn0_estimation_counter++;
                    }
                else
                    {
                        d_cn0_estimation_counter = 0;
                        d_CN0_SNV_dB_Hz = cn0_svn_estimator(d_Prompt_buffer.data(), d_cn0_samples, 0.001);
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
                current_synchro_data.Prompt_I = static_cast<double>(d_correlator_outs[1].real());
                current_synchro_data.Prompt_Q = static_cast<double>(d_correlator_outs[1].imag());
                current_synchro_data.Tracking_sample_counter = d_sample_counter + static_cast<uint64_t>(d_current_prn_length_samples);
                current_synchro_data.Code_phase_samples = d_rem_code_phase_samples;
                current_synchro_data.Carrier_phase_rads = d_acc_carrier_phase_rad;
                current_synchro_data.Carrier_Doppler_hz = d_carrier_doppler_hz;
                current_synchro_data.CN0_dB_hz = d_CN0_SNV_dB_Hz;
                current_synchro_data.Flag_valid_symbol_output = true;
                current_synchro_data.correlation_length_ms = 1;
            }
        else
            {
                std::fill(d_correlator_outs.begin(), d_correlator_outs.end(), gr_complex(0, 0));
                current_synchro_data.Tracking_sample_counter = d_sample_counter + static_cast<uint64_t>(d_current_prn_length_samples);
                current_synchro_data.System = 'R';
                current_synchro_data.correlation_length_ms = 1;
            }
        current_synchro_data.fs = d_fs_in;
        *out = current_synchro_data;
        d_sample_counter += d_current_prn_length_samples;
        return d_current_prn_length_samples;
    }

    bool tracking_enabled() const { return d_enable_tracking; }
    double carrier_doppler_hz() const { return d_carrier_doppler_hz; }
    double carrier_frequency_hz() const { return d_carrier_frequency_hz; }
    double code_freq_chips() const { return d_code_freq_chips; }
    double cn0_db_hz() const { return d_CN0_SNV_dB_Hz; }
    double carrier_lock_test() const { return d_carrier_lock_test; }
    uint64_t sample_counter() const { return d_sample_counter; }
    const std::vector<gr_complex>& correlator_outs() const { return d_correlator_outs; }
    const std::vector<int>& events() const { return d_events; }
    gc_status last_status() const { return multicorrelator_cpu.last_status(); }

private:
    static constexpr double kTwoPi = 6.283185307179586;  // GLONASS_TWO_PI
    static constexpr double kCodeRateHz = 0.511e6;       // GLONASS_L1_CA_CODE_RATE_HZ (L2 C/A: the same)
    static constexpr double kCodeLengthChips = 511.0;

    gnsscorr::GlonassBand d_band;
    int64_t d_fs_in;
    uint32_t d_vector_length;
    int d_cn0_samples, d_cn0_min, d_max_lock_fail;
    double d_carrier_lock_threshold;
    std::map<uint32_t, int32_t> d_glonass_prn;
    uint32_t d_channel = 0;
    Gnss_Synchro* d_acquisition_gnss_synchro = nullptr;
    Hip_Multicorrelator multicorrelator_cpu;
    std::vector<gr_complex> d_ca_code, d_correlator_outs, d_Prompt_buffer;
    std::vector<float> d_local_code_shift_chips;
    Tracking_2nd_DLL_filter d_code_loop_filter;
    Tracking_2nd_PLL_filter d_carrier_loop_filter;
    double d_glonass_freq_ch = 0.0;
    double d_acq_code_phase_samples = 0.0, d_acq_carrier_doppler_hz = 0.0;
    uint64_t d_acq_sample_stamp = 0, d_sample_counter = 0;
    double d_code_freq_chips = 0.0, d_code_phase_step_chips = 0.0;
    double d_carrier_frequency_hz = 0.0, d_carrier_doppler_hz = 0.0;
    double d_carrier_phase_step_rad = 0.0, d_carrier_doppler_phase_step_rad = 0.0;
    double d_rem_code_phase_samples = 0.0, d_rem_code_phase_chips = 0.0, d_rem_carr_phase_rad = 0.0, d_acc_carrier_phase_rad = 0.0;
    int32_t d_current_prn_length_samples = 0;
    int32_t d_cn0_estimation_counter = 0, d_carrier_lock_fail_counter = 0;
    double d_carrier_lock_test = 1.0, d_CN0_SNV_dB_Hz = 0.0;
    bool d_enable_tracking = false, d_pull_in = false;
    std::vector<int> d_events;
};

//! TrackingInterface adapter (glonass_l1_ca_dll_pll_tracking.cc:46-130; glonass_l2_ca_dll_pll_tracking.cc is the same with "2G")
template <int BAND>
class GlonassCaDllPllTrackingHip : public TrackingInterface
{
public:
    GlonassCaDllPllTrackingHip(ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
        : role_(role), in_streams_(in_streams), out_streams_(out_streams)
    {
        int fs_in_deprecated = configuration->property("GNSS-SDR.internal_fs_hz", 2048000);
        const int fs_in = configuration->property("GNSS-SDR.internal_fs_sps", fs_in_deprecated);
        const float pll_bw_hz = configuration->property(role + ".pll_bw_hz", 50.0f);
        const float dll_bw_hz = configuration->property(role + ".dll_bw_hz", 2.0f);
        const float early_late_space_chips = configuration->property(role + ".early_late_space_chips", 0.5f);
        vector_length_ = std::round(fs_in / (0.511e6 / 511.0));
        tracking_ = std::make_shared<hip_glonass_ca_dll_pll_tracking>(BAND, fs_in, vector_length_, pll_bw_hz, dll_bw_hz, early_late_space_chips,
            configuration->property(role + ".cn0_samples", 20), configuration->property(role + ".cn0_min", 25), configuration->property(role + ".max_lock_fail", 50),
            configuration->property(role + ".carrier_lock_th", 0.85));
    }

    std::string role() override { return role_; }
    std::string implementation() override { return gnsscorr::glonass_band(BAND).implementation; }
    size_t item_size() override { return sizeof(gr_complex); }
    void start_tracking() override { tracking_->start_tracking(); }
    void stop_tracking() override { tracking_->stop_tracking(); }
    void set_channel(unsigned int channel) override { tracking_->set_channel(channel); }
    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) override { tracking_->set_gnss_synchro(p_gnss_synchro); }
    std::shared_ptr<hip_glonass_ca_dll_pll_tracking> block() { return tracking_; }
    unsigned int vector_length() const { return vector_length_; }

private:
    std::shared_ptr<hip_glonass_ca_dll_pll_tracking> tracking_;
    std::string role_;
    unsigned int in_streams_, out_streams_;
    unsigned int vector_length_ = 0;
};

using GlonassL1CaDllPllTrackingHip = GlonassCaDllPllTrackingHip<1>;
using GlonassL2CaDllPllTrackingHip = GlonassCaDllPllTrackingHip<2>;

#endif  // GNSSCORR_HIP_GLONASS_CA_DLL_PLL_TRACKING_H_
