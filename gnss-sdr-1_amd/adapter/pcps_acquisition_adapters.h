/*!
 * \file pcps_acquisition_adapters.h
 * \brief AcquisitionInterface adapters (GPS L1 C/A, L2C(M), L5I; Galileo E1, E5a; BeiDou B1I, B3I; GLONASS L1 C/A) backed by
 * hip_pcps_acquisition.
 *
 * Configuration keys, derived sizes, local-code generation and the Pfa -> threshold rule follow
 *   GpsL1CaPcpsAcquisition           src/algorithms/acquisition/adapters/gps_l1_ca_pcps_acquisition.cc:46-358
 *   GalileoE1PcpsAmbiguousAcquisition src/algorithms/acquisition/adapters/galileo_e1_pcps_ambiguous_acquisition.cc:46-330
 *   BeidouB1iPcpsAcquisition          src/algorithms/acquisition/adapters/beidou_b1i_pcps_acquisition.cc:45-330
 *   GlonassL1CaPcpsAcquisition        src/algorithms/acquisition/adapters/glonass_l1_ca_pcps_acquisition.cc:44-330
 *   GpsL2MPcpsAcquisition, GpsL5iPcpsAcquisition, GalileoE5aPcpsAcquisition, BeidouB3iPcpsAcquisition (same directory)
 * (item_type gr_complex only; `dump` / `dump_filename` / `dump_channel` write the reference's .mat
 * variables, see hip_pcps_acquisition::dump_results; the acquisition resampler and the GNU Radio
 * connect()/get_left_block() plumbing are outside this path).  Registration in a GNSS-SDR tree is one
 * more `else if (implementation == "GPS_L1_CA_PCPS_Acquisition_HIP")` in
 * GNSSBlockFactory::GetAcqBlock (src/core/receiver/gnss_block_factory.cc:1946), see INTEGRATION.md.
 */
#ifndef GNSSCORR_PCPS_ACQUISITION_ADAPTERS_H_
#define GNSSCORR_PCPS_ACQUISITION_ADAPTERS_H_

#include "hip_pcps_acquisition.h"
#include <cmath>
#include <memory>
#include <string>
#include <vector>

namespace gnsscorr
{
enum class AcqSignal
{
    GPS_L1_CA,
    GALILEO_E1,
    BEIDOU_B1I,
    GLONASS_L1_CA,
    GPS_L2_M,
    GPS_L5I,
    GALILEO_E5A,
    BEIDOU_B3I
};

// constants of GPS_L1_CA.h:54-61, Galileo_E1.h:55-60, Beidou_B1I.h:52-56
struct AcqSignalTraits
{
    double code_rate_hz;
    double code_length_chips;
    double chip_period_s;
    double code_period_s;
    uint32_t ms_per_code;
    const char* implementation;
};

inline AcqSignalTraits acq_traits(AcqSignal s)
{
    switch (s)
        {
        case AcqSignal::GALILEO_E1:
            return {1.023e6, 4092.0, 1.0 / 1.023e6, 0.004, 4, "Galileo_E1_PCPS_Ambiguous_Acquisition_HIP"};
        case AcqSignal::BEIDOU_B1I:
            return {2.046e6, 2046.0, 4.8875e-07, 0.001, 1, "BEIDOU_B1I_PCPS_Acquisition_HIP"};
        case AcqSignal::GLONASS_L1_CA:
            return {0.511e6, 511.0, 1.9569e-06, 0.001, 1, "GLONASS_L1_CA_PCPS_Acquisition_HIP"};  // GLONASS_L1_L2_CA.h:93-97
        case AcqSignal::GPS_L2_M:
            return {0.5115e6, 10230.0, 1.0 / 0.5115e6, 0.02, 20, "GPS_L2_M_PCPS_Acquisition_HIP"};  // GPS_L2C.h:56-58
        case AcqSignal::GPS_L5I:
            return {10.23e6, 10230.0, 1.0 / 10.23e6, 0.001, 1, "GPS_L5i_PCPS_Acquisition_HIP"};  // GPS_L5.h:54-57
        case AcqSignal::GALILEO_E5A:
            return {1.023e7, 10230.0, 1.0 / 1.023e7, 0.001, 1, "Galileo_E5a_Pcps_Acquisition_HIP"};  // Galileo_E5a.h:44-51
        case AcqSignal::BEIDOU_B3I:
            return {10.23e6, 10230.0, 1.0 / 10.23e6, 0.001, 1, "BEIDOU_B3I_PCPS_Acquisition_HIP"};  // Beidou_B3I.h:41-44
        default:
            return {1.023e6, 1023.0, 9.7752e-07, 0.001, 1, "GPS_L1_CA_PCPS_Acquisition_HIP"};
        }
}
}  // namespace gnsscorr


template <gnsscorr::AcqSignal SIG>
class PcpsAcquisitionHip : public AcquisitionInterface
{
public:
    PcpsAcquisitionHip(ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
        : configuration_(configuration), role_(role), in_streams_(in_streams), out_streams_(out_streams)
    {
        using gnsscorr::AcqSignal;
        const gnsscorr::AcqSignalTraits t = gnsscorr::acq_traits(SIG);
        item_type_ = configuration_->property(role + ".item_type", std::string("gr_complex"));
        int64_t fs_in_deprecated = configuration_->property("GNSS-SDR.internal_fs_hz", static_cast<int64_t>(2048000));
        fs_in_ = configuration_->property("GNSS-SDR.internal_fs_sps", fs_in_deprecated);
        acq_parameters_.fs_in = fs_in_;
        acq_parameters_.resampled_fs = fs_in_;
        acq_parameters_.blocking = configuration_->property(role + ".blocking", true);
        doppler_max_ = configuration_->property(role + ".doppler_max", 5000);
        acq_parameters_.doppler_max = doppler_max_;
        acq_parameters_.bit_transition_flag = configuration_->property(role + ".bit_transition_flag", false);
        acq_parameters_.use_CFAR_algorithm_flag = configuration_->property(role + ".use_CFAR_algorithm", true);
        acq_parameters_.max_dwells = configuration_->property(role + ".max_dwells", 1);
        acq_parameters_.num_doppler_bins_step2 = configuration_->property(role + ".second_nbins", 4);
        acq_parameters_.doppler_step2 = configuration_->property(role + ".second_doppler_step", 125.0f);
        acq_parameters_.make_2_steps = configuration_->property(role + ".make_two_steps", false);
        // gps_l1_ca_pcps_acquisition.cc:56,64-66,84-85
        acq_parameters_.dump = configuration_->property(role + ".dump", false);
        acq_parameters_.dump_channel = configuration_->property(role + ".dump_channel", 0);
        acq_parameters_.dump_filename = configuration_->property(role + ".dump_filename", std::string("./acquisition.mat"));
        acq_parameters_.it_size = sizeof(gr_complex);
        if (SIG == AcqSignal::GPS_L1_CA)
            {
                // gps_l1_ca_pcps_acquisition.cc:75-120
                sampled_ms_ = configuration_->property(role + ".coherent_integration_time_ms", 1);
                acq_parameters_.sampled_ms = sampled_ms_;
                acq_parameters_.ms_per_code = 1;
                code_length_ = static_cast<unsigned int>(std::floor(static_cast<double>(fs_in_) / (t.code_rate_hz / t.code_length_chips)));
                acq_parameters_.samples_per_ms = static_cast<float>(fs_in_) * 0.001;
                acq_parameters_.samples_per_chip = static_cast<unsigned int>(std::ceil(t.chip_period_s * static_cast<float>(acq_parameters_.fs_in)));
                acq_parameters_.samples_per_code = acq_parameters_.samples_per_ms * static_cast<float>(t.code_period_s * 1000.0);
                vector_length_ = std::floor(acq_parameters_.sampled_ms * acq_parameters_.samples_per_ms) * (acq_parameters_.bit_transition_flag ? 2 : 1);
            }
        else if (SIG == AcqSignal::GALILEO_E1)
            {
                // galileo_e1_pcps_ambiguous_acquisition.cc:67-123
                acq_parameters_.ms_per_code = 4;
                sampled_ms_ = configuration_->property(role + ".coherent_integration_time_ms", acq_parameters_.ms_per_code);
                acq_parameters_.sampled_ms = sampled_ms_;
                if ((acq_parameters_.sampled_ms % acq_parameters_.ms_per_code) != 0)
                    {
                        acq_parameters_.sampled_ms = acq_parameters_.ms_per_code;
                        sampled_ms_ = acq_parameters_.ms_per_code;
                    }
                acquire_pilot_ = configuration_->property(role + ".acquire_pilot", false);
                code_length_ = static_cast<unsigned int>(std::floor(static_cast<double>(fs_in_) / (t.code_rate_hz / t.code_length_chips)));
                acq_parameters_.samples_per_ms = static_cast<float>(fs_in_) * 0.001;
                acq_parameters_.samples_per_chip = static_cast<unsigned int>(std::ceil((1.0 / t.code_rate_hz) * static_cast<float>(acq_parameters_.fs_in)));
                acq_parameters_.samples_per_code = acq_parameters_.samples_per_ms * static_cast<float>(4);
                vector_length_ = sampled_ms_ * acq_parameters_.samples_per_ms;
                if (acq_parameters_.bit_transition_flag) vector_length_ *= 2;
            }
        else if (SIG == AcqSignal::GPS_L2_M)
            {
                // gps_l2_m_pcps_acquisition.cc:81-128,155: whole 20 ms codes
                acq_parameters_.ms_per_code = 20;
                sampled_ms_ = configuration_->property(role + ".coherent_integration_time_ms", acq_parameters_.ms_per_code);
                if ((sampled_ms_ % acq_parameters_.ms_per_code) != 0) sampled_ms_ = acq_parameters_.ms_per_code;
                acq_parameters_.sampled_ms = sampled_ms_;
                code_length_ = static_cast<unsigned int>(std::floor(static_cast<double>(fs_in_) / (t.code_rate_hz / t.code_length_chips)));
                acq_parameters_.samples_per_ms = static_cast<float>(fs_in_) * 0.001;
                acq_parameters_.samples_per_chip = static_cast<unsigned int>(std::ceil((1.0 / t.code_rate_hz) * static_cast<float>(acq_parameters_.fs_in)));
                acq_parameters_.samples_per_code = acq_parameters_.samples_per_ms * static_cast<float>(t.code_period_s * 1000.0);
                vector_length_ = acq_parameters_.sampled_ms * acq_parameters_.samples_per_ms * (acq_parameters_.bit_transition_flag ? 2 : 1);
            }
        else if (SIG == AcqSignal::GPS_L5I)
            {
                // gps_l5i_pcps_acquisition.cc:82-135
                sampled_ms_ = configuration_->property(role + ".coherent_integration_time_ms", 1);
                acq_parameters_.sampled_ms = sampled_ms_;
                acq_parameters_.ms_per_code = 1;
                code_length_ = static_cast<unsigned int>(std::floor(static_cast<double>(fs_in_) / (t.code_rate_hz / t.code_length_chips)));
                acq_parameters_.samples_per_ms = static_cast<float>(fs_in_) * 0.001;
                acq_parameters_.samples_per_chip = static_cast<unsigned int>(std::ceil((1.0 / t.code_rate_hz) * static_cast<float>(acq_parameters_.fs_in)));
                acq_parameters_.samples_per_code = acq_parameters_.samples_per_ms * static_cast<float>(t.code_period_s * 1000.0);
                vector_length_ = std::floor(acq_parameters_.sampled_ms * acq_parameters_.samples_per_ms) * (acq_parameters_.bit_transition_flag ? 2 : 1);
            }
        else if (SIG == AcqSignal::GALILEO_E5A)
            {
                // galileo_e5a_pcps_acquisition.cc:60-144: 1 ms, data (5I), pilot (5Q) or both components (5X) in the replica
                acquire_pilot_ = configuration_->property(role + ".acquire_pilot", false);
                acquire_iq_ = configuration_->property(role + ".acquire_iq", false);
                if (acquire_iq_) acquire_pilot_ = false;
                sampled_ms_ = 1;
                acq_parameters_.sampled_ms = sampled_ms_;
                acq_parameters_.ms_per_code = 1;
                acq_parameters_.samples_per_ms = static_cast<float>(fs_in_) * 0.001;
                acq_parameters_.samples_per_chip = static_cast<unsigned int>(std::ceil((1.0 / t.code_rate_hz) * static_cast<float>(acq_parameters_.fs_in)));
                code_length_ = static_cast<unsigned int>(std::round(static_cast<double>(fs_in_) / t.code_rate_hz * t.code_length_chips));
                vector_length_ = code_length_ * sampled_ms_;
                if (acq_parameters_.bit_transition_flag) vector_length_ *= 2;
                acq_parameters_.samples_per_code = acq_parameters_.samples_per_ms * 1.0f;
            }
        else if (SIG == AcqSignal::GLONASS_L1_CA)
            {
                // glonass_l1_ca_pcps_acquisition.cc:62-113
                sampled_ms_ = configuration_->property(role + ".coherent_integration_time_ms", 1);
                acq_parameters_.sampled_ms = sampled_ms_;
                acq_parameters_.ms_per_code = 1;
                acq_parameters_.samples_per_chip = static_cast<unsigned int>(std::ceil(t.chip_period_s * static_cast<float>(acq_parameters_.fs_in)));
                code_length_ = static_cast<unsigned int>(std::round(static_cast<double>(fs_in_) / (t.code_rate_hz / t.code_length_chips)));
                vector_length_ = code_length_ * sampled_ms_;
                if (acq_parameters_.bit_transition_flag) vector_length_ *= 2;
                acq_parameters_.samples_per_ms = static_cast<float>(fs_in_) * 0.001;
                acq_parameters_.samples_per_code = acq_parameters_.samples_per_ms * static_cast<float>(t.code_period_s * 1000.0);
            }
        else
            {
                // beidou_b1i_pcps_acquisition.cc:73-104, beidou_b3i_pcps_acquisition.cc:71-104: ms_per_code and samples_per_chip keep
                // Acq_Conf's zeros
                sampled_ms_ = configuration_->property(role + ".coherent_integration_time_ms", 1);
                acq_parameters_.sampled_ms = sampled_ms_;
                code_length_ = static_cast<uint32_t>(std::round(static_cast<double>(fs_in_) / (t.code_rate_hz / t.code_length_chips)));
                vector_length_ = code_length_ * sampled_ms_;
                if (acq_parameters_.bit_transition_flag) vector_length_ *= 2;
                acq_parameters_.samples_per_ms = code_length_;
                acq_parameters_.samples_per_code = code_length_;
            }
        code_.resize(vector_length_);
        acquisition_ = std::make_shared<hip_pcps_acquisition>(acq_parameters_);
    }

    std::string role() override { return role_; }
    std::string implementation() override { return gnsscorr::acq_traits(SIG).implementation; }
    size_t item_size() override { return sizeof(gr_complex); }

    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) override
    {
        gnss_synchro_ = p_gnss_synchro;
        acquisition_->set_gnss_synchro(gnss_synchro_);
    }

    void set_channel(unsigned int channel) override
    {
        channel_ = channel;
        acquisition_->set_channel(channel_);
    }

    void set_channel_fsm(std::shared_ptr<ChannelFsm> channel_fsm) override
    {
        channel_fsm_ = channel_fsm;
        acquisition_->set_channel_fsm(channel_fsm);
    }

    void set_threshold(float threshold) override
    {
        float pfa = configuration_->property(role_ + ".pfa", 0.0f);
        threshold_ = (pfa == 0.0f) ? threshold : calculate_threshold(pfa);
        acquisition_->set_threshold(threshold_);
    }

    void set_doppler_max(unsigned int doppler_max) override
    {
        doppler_max_ = doppler_max;
        acquisition_->set_doppler_max(doppler_max_);
    }

    void set_doppler_step(unsigned int doppler_step) override
    {
        doppler_step_ = doppler_step;
        acquisition_->set_doppler_step(doppler_step_);
    }

    void init() override { acquisition_->init(); }

    void set_local_code() override
    {
        using gnsscorr::AcqSignal;
        std::vector<gr_complex> code(code_length_ + 8);
        float* dst = reinterpret_cast<float*>(code.data());
        unsigned int reps = sampled_ms_;
        if (SIG == AcqSignal::GPS_L1_CA)
            gc_gps_l1_ca_code_gen_complex_sampled(dst, gnss_synchro_->PRN, static_cast<int32_t>(fs_in_), 0, nullptr);
        else if (SIG == AcqSignal::BEIDOU_B1I)
            gc_beidou_b1i_code_gen_complex_sampled(dst, gnss_synchro_->PRN, static_cast<int32_t>(fs_in_), 0, nullptr);
        else if (SIG == AcqSignal::GLONASS_L1_CA)
            gc_glonass_l1_ca_code_gen_complex_sampled(dst, static_cast<int32_t>(fs_in_), 0, nullptr);  // one code for every slot (FDMA)
        else if (SIG == AcqSignal::BEIDOU_B3I)
            gc_beidou_b3i_code_gen_complex_sampled(dst, gnss_synchro_->PRN, static_cast<int32_t>(fs_in_), 0, nullptr);
        else if (SIG == AcqSignal::GPS_L5I)
            gc_gps_l5i_code_gen_complex_sampled(dst, gnss_synchro_->PRN, static_cast<int32_t>(fs_in_), nullptr);
        else if (SIG == AcqSignal::GPS_L2_M)
            {
                gc_gps_l2c_m_code_gen_complex_sampled(dst, gnss_synchro_->PRN, static_cast<int32_t>(fs_in_), nullptr);
                reps = sampled_ms_ / 20;
            }
        else if (SIG == AcqSignal::GALILEO_E5A)
            {
                // galileo_e5a_pcps_acquisition.cc:238-266
                const char* sig = acquire_iq_ ? "5X" : acquire_pilot_ ? "5Q" : "5I";
                gc_galileo_e5_a_code_gen_complex_sampled(dst, sig, gnss_synchro_->PRN, static_cast<int32_t>(fs_in_), 0, nullptr);
            }
        else
            {
                // galileo_e1_pcps_ambiguous_acquisition.cc:240-284: the cboc flag is looked up per channel
                bool cboc = configuration_->property("Acquisition" + std::to_string(channel_) + ".cboc", false);
                char pilot_signal[3] = "1C";
                const char* sig = acquire_pilot_ ? pilot_signal : gnss_synchro_->Signal;
                char sig3[3] = {sig[0], sig[1], 0};
                gc_galileo_e1_code_gen_complex_sampled(dst, sig3, cboc ? 1 : 0, gnss_synchro_->PRN, static_cast<int32_t>(fs_in_), 0, nullptr);
                reps = sampled_ms_ / 4;
            }
        for (unsigned int i = 0; i < reps; i++) std::memcpy(&code_[i * code_length_], code.data(), sizeof(gr_complex) * code_length_);
        acquisition_->set_local_code(code_.data());
    }

    void set_state(int state) override { acquisition_->set_state(state); }
    signed int mag() override { return acquisition_->mag(); }
    void reset() override { acquisition_->set_active(true); }
    void stop_acquisition() override {}
    void set_resampler_latency(uint32_t latency_samples) override { acquisition_->set_resampler_latency(latency_samples); }

    //! the block (get_left_block()/get_right_block() in the reference)
    std::shared_ptr<hip_pcps_acquisition> block() { return acquisition_; }
    unsigned int vector_length() const { return vector_length_; }
    float threshold() const { return threshold_; }

private:
    //! gps_l1_ca_pcps_acquisition.cc:262-279 (same rule in the Galileo and BeiDou adapters); the quantile
    //! of an exponential distribution of rate lambda is -ln(1 - p) / lambda
    float calculate_threshold(float pfa)
    {
        unsigned int frequency_bins = 0;
        for (int doppler = static_cast<int>(-doppler_max_); doppler <= static_cast<int>(doppler_max_); doppler += doppler_step_) frequency_bins++;
        unsigned int ncells = vector_length_ * frequency_bins;
        double exponent = 1 / static_cast<double>(ncells);
        double val = std::pow(1.0 - pfa, exponent);
        auto lambda = double(vector_length_);
        return static_cast<float>(-std::log(1.0 - val) / lambda);
    }

    ConfigurationInterface* configuration_;
    std::shared_ptr<hip_pcps_acquisition> acquisition_;
    Acq_Conf acq_parameters_;
    std::string item_type_;
    std::string role_;
    unsigned int in_streams_;
    unsigned int out_streams_;
    unsigned int vector_length_ = 0;
    unsigned int code_length_ = 0;
    unsigned int channel_ = 0;
    unsigned int doppler_max_ = 0;
    unsigned int doppler_step_ = 0;
    unsigned int sampled_ms_ = 1;
    bool acquire_pilot_ = false;
    bool acquire_iq_ = false;
    float threshold_ = 0.0f;
    int64_t fs_in_ = 0;
    std::shared_ptr<ChannelFsm> channel_fsm_;
    std::vector<gr_complex> code_;
    Gnss_Synchro* gnss_synchro_ = nullptr;
};

using GpsL1CaPcpsAcquisitionHip = PcpsAcquisitionHip<gnsscorr::AcqSignal::GPS_L1_CA>;
using GalileoE1PcpsAmbiguousAcquisitionHip = PcpsAcquisitionHip<gnsscorr::AcqSignal::GALILEO_E1>;
using BeidouB1iPcpsAcquisitionHip = PcpsAcquisitionHip<gnsscorr::AcqSignal::BEIDOU_B1I>;
using GpsL2MPcpsAcquisitionHip = PcpsAcquisitionHip<gnsscorr::AcqSignal::GPS_L2_M>;
using GpsL5iPcpsAcquisitionHip = PcpsAcquisitionHip<gnsscorr::AcqSignal::GPS_L5I>;
using GalileoE5aPcpsAcquisitionHip = PcpsAcquisitionHip<gnsscorr::AcqSignal::GALILEO_E5A>;
using BeidouB3iPcpsAcquisitionHip = PcpsAcquisitionHip<gnsscorr::AcqSignal::BEIDOU_B3I>;
//! the block applies the FDMA offset in set_local_code() once hip_pcps_acquisition::set_glonass_channel_map() holds the almanac
using GlonassL1CaPcpsAcquisitionHip = PcpsAcquisitionHip<gnsscorr::AcqSignal::GLONASS_L1_CA>;

#endif  // GNSSCORR_PCPS_ACQUISITION_ADAPTERS_H_
