/*!
 * \file gnss_sdr_types.h
 * \brief The GNSS-SDR types the acquisition / tracking adapters exchange with their callers.
 *
 * Inside a GNSS-SDR build define GNSSCORR_WITH_GNSS_SDR: the real headers are used
 * (acq_conf.h, gnss_synchro.h, configuration_interface.h, channel_fsm.h, the
 * core/interfaces). Stand-alone (this repository's tests) the declarations below
 * mirror the members this path touches, with the reference's names:
 *   Acq_Conf                 src/algorithms/acquisition/libs/acq_conf.h:38-66
 *   Gnss_Synchro             src/core/system_parameters/gnss_synchro.h:45-79
 *   ConfigurationInterface   src/core/interfaces/configuration_interface.h:49-61
 *   GNSSBlockInterface       src/core/interfaces/gnss_block_interface.h:53-82 (without the GNU Radio
 *                            connect()/get_left_block() members: there is no flowgraph here)
 *   AcquisitionInterface     src/core/interfaces/acquisition_interface.h:56-72
 *   TrackingInterface        src/core/interfaces/tracking_interface.h:55-62
 *   ChannelFsm               src/algorithms/channel/libs/channel_fsm.h (the events acquisition raises)
 */
#ifndef GNSSCORR_GNSS_SDR_TYPES_H_
#define GNSSCORR_GNSS_SDR_TYPES_H_

#ifdef GNSSCORR_WITH_GNSS_SDR
#include "acq_conf.h"
#include "acquisition_interface.h"
#include "channel_fsm.h"
#include "configuration_interface.h"
#include "gnss_synchro.h"
#include "tracking_interface.h"
#else

#include <complex>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>

using gr_complex = std::complex<float>;

class Acq_Conf
{
public:
    // defaults of Acq_Conf::Acq_Conf() (src/algorithms/acquisition/libs/acq_conf.cc:34-60): the adapters
    // rely on them (e.g. the BeiDou adapter never sets ms_per_code, so its FFT is twice the code period)
    uint32_t sampled_ms = 0U;
    uint32_t ms_per_code = 0U;
    uint32_t samples_per_chip = 0U;
    uint32_t max_dwells = 0U;
    uint32_t doppler_max = 0U;
    uint32_t num_doppler_bins_step2 = 0U;
    float doppler_step2 = 0.0f;
    int64_t fs_in = 0LL;
    float samples_per_ms = 0.0f;
    float samples_per_code = 0.0f;
    bool bit_transition_flag = false;
    bool use_CFAR_algorithm_flag = false;
    bool dump = false;
    bool blocking = false;
    bool blocking_on_standby = false;
    bool make_2_steps = false;
    bool use_automatic_resampler = false;
    float resampler_ratio = 1.0f;
    int64_t resampled_fs = 0LL;
    uint32_t resampler_latency_samples = 0U;
    std::string dump_filename;
    uint32_t dump_channel = 0U;
    size_t it_size = sizeof(char);
};

class Gnss_Synchro
{
public:
    char System = 0;
    char Signal[3] = {0, 0, 0};
    uint32_t PRN = 0;
    int32_t Channel_ID = 0;
    double Acq_delay_samples = 0.0;
    double Acq_doppler_hz = 0.0;
    uint64_t Acq_samplestamp_samples = 0;
    uint32_t Acq_doppler_step = 0;
    bool Flag_valid_acquisition = false;
    int64_t fs = 0;
    double Prompt_I = 0.0;
    double Prompt_Q = 0.0;
    double CN0_dB_hz = 0.0;
    double Carrier_Doppler_hz = 0.0;
    double Carrier_phase_rads = 0.0;
    double Code_phase_samples = 0.0;
    uint64_t Tracking_sample_counter = 0;
    bool Flag_valid_symbol_output = false;
    int32_t correlation_length_ms = 0;
    bool Flag_valid_word = false;
    uint32_t TOW_at_current_symbol_ms = 0;
    double Pseudorange_m = 0.0;
    double RX_time = 0.0;
    bool Flag_valid_pseudorange = false;
    double interp_TOW_ms = 0.0;
};

class ConfigurationInterface
{
public:
    virtual ~ConfigurationInterface() = default;
    virtual std::string property(std::string property_name, std::string default_value) = 0;
    virtual bool property(std::string property_name, bool default_value) = 0;
    virtual int64_t property(std::string property_name, int64_t default_value) = 0;
    virtual int32_t property(std::string property_name, int32_t default_value) = 0;
    virtual uint32_t property(std::string property_name, uint32_t default_value) = 0;
    virtual float property(std::string property_name, float default_value) = 0;
    virtual double property(std::string property_name, double default_value) = 0;
    virtual void set_property(std::string property_name, std::string value) = 0;
};

//! src/core/receiver/in_memory_configuration.h equivalent
class InMemoryConfiguration : public ConfigurationInterface
{
public:
    std::string property(std::string n, std::string d) override { return has(n) ? d_map[n] : d; }
    bool property(std::string n, bool d) override { return has(n) ? (d_map[n] == "true" || d_map[n] == "1") : d; }
    int64_t property(std::string n, int64_t d) override { return has(n) ? std::strtoll(d_map[n].c_str(), nullptr, 10) : d; }
    int32_t property(std::string n, int32_t d) override { return has(n) ? static_cast<int32_t>(std::strtol(d_map[n].c_str(), nullptr, 10)) : d; }
    uint32_t property(std::string n, uint32_t d) override { return has(n) ? static_cast<uint32_t>(std::strtoul(d_map[n].c_str(), nullptr, 10)) : d; }
    float property(std::string n, float d) override { return has(n) ? std::strtof(d_map[n].c_str(), nullptr) : d; }
    double property(std::string n, double d) override { return has(n) ? std::strtod(d_map[n].c_str(), nullptr) : d; }
    void set_property(std::string n, std::string v) override { d_map[n] = v; }

private:
    bool has(const std::string& n) const { return d_map.count(n) != 0; }
    std::map<std::string, std::string> d_map;
};

//! The channel state machine as seen from the acquisition block (pcps_acquisition.cc:432-436)
class ChannelFsm
{
public:
    virtual ~ChannelFsm() = default;
    virtual bool Event_valid_acquisition() = 0;
};

class GNSSBlockInterface
{
public:
    virtual ~GNSSBlockInterface() = default;
    virtual std::string role() = 0;
    virtual std::string implementation() = 0;
    virtual size_t item_size() = 0;
};

class AcquisitionInterface : public GNSSBlockInterface
{
public:
    virtual void set_gnss_synchro(Gnss_Synchro* gnss_synchro) = 0;
    virtual void set_channel(unsigned int channel_id) = 0;
    virtual void set_channel_fsm(std::shared_ptr<ChannelFsm> channel_fsm) = 0;
    virtual void set_threshold(float threshold) = 0;
    virtual void set_doppler_max(unsigned int doppler_max) = 0;
    virtual void set_doppler_step(unsigned int doppler_step) = 0;
    virtual void init() = 0;
    virtual void set_local_code() = 0;
    virtual void set_state(int state) = 0;
    virtual signed int mag() = 0;
    virtual void reset() = 0;
    virtual void stop_acquisition() = 0;
    virtual void set_resampler_latency(uint32_t latency_samples) = 0;
};

class TrackingInterface : public GNSSBlockInterface
{
public:
    virtual void start_tracking() = 0;
    virtual void stop_tracking() = 0;
    virtual void set_gnss_synchro(Gnss_Synchro* gnss_synchro) = 0;
    virtual void set_channel(unsigned int channel) = 0;
};

#endif  // GNSSCORR_WITH_GNSS_SDR
#endif  // GNSSCORR_GNSS_SDR_TYPES_H_
