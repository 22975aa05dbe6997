/*!
 * \file dll_pll_tracking_adapters.h
 * \brief TrackingInterface adapters for GPS L1 C/A, L2C(M), L5, Galileo E1, E5a and BeiDou B1I, B3I backed by
 * hip_dll_pll_veml_tracking.  Configuration keys and defaults follow
 *   GpsL1CaDllPllTracking        src/algorithms/tracking/adapters/gps_l1_ca_dll_pll_tracking.cc:47-214
 *   GalileoE1DllPllVemlTracking  src/algorithms/tracking/adapters/galileo_e1_dll_pll_veml_tracking.cc:47-215
 *   BeidouB1iDllPllTracking      src/algorithms/tracking/adapters/beidou_b1i_dll_pll_tracking.cc:47-200
 *   GpsL2MDllPllTracking, GpsL5DllPllTracking, GalileoE5aDllPllTracking, BeidouB3iDllPllTracking (same directory)
 * (gflags overrides, dump files and the GNU Radio connect()/get_left_block() plumbing are outside this path).
 * Registration: `else if (implementation == "GPS_L1_CA_DLL_PLL_Tracking_HIP")` in
 * GNSSBlockFactory::GetTrkBlock (src/core/receiver/gnss_block_factory.cc:2131), see INTEGRATION.md.
 */
#ifndef GNSSCORR_DLL_PLL_TRACKING_ADAPTERS_H_
#define GNSSCORR_DLL_PLL_TRACKING_ADAPTERS_H_

#include "hip_dll_pll_veml_tracking.h"
#include "hip_dll_pll_veml_tracking_dev.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <type_traits>

namespace gnsscorr
{
enum class TrkSignal
{
    GPS_L1_CA,
    GALILEO_E1,
    BEIDOU_B1I,
    GPS_L2_M,
    GPS_L5,
    GALILEO_E5A,
    BEIDOU_B3I
};

//! what differs between the reference's adapters: signal constants, default loop settings, extension / pilot rules
struct TrkSignalTraits
{
    const char* implementation;
    char system;
    const char* signal;
    double code_rate_hz, code_length_chips;
    float pll_bw_hz, dll_bw_hz, pll_bw_narrow_hz, dll_bw_narrow_hz;
    float early_late_space_chips, early_late_space_narrow_chips;
    bool veml;
    int cn0_min;
    double carrier_lock_th;
    bool has_pilot;
    int max_extension_data;  //!< longest integration (symbols) on the data component; 0: only with the pilot
};

inline const TrkSignalTraits& trk_traits(TrkSignal s)
{
    static const TrkSignalTraits t[] = {
        {"GPS_L1_CA_DLL_PLL_Tracking_HIP", 'G', "1C", 1.023e6, 1023.0, 50.0f, 2.0f, 20.0f, 2.0f, 0.5f, 0.5f, false, 30, 0.80, false, 20},
        {"Galileo_E1_DLL_PLL_VEML_Tracking_HIP", 'E', "1B", 1.023e6, 4092.0, 5.0f, 0.5f, 2.0f, 0.25f, 0.15f, 0.15f, true, 25, 0.85, true, 0},
        {"BEIDOU_B1I_DLL_PLL_Tracking_HIP", 'C', "B1", 2.046e6, 2046.0, 50.0f, 2.0f, 20.0f, 2.0f, 0.5f, 0.5f, false, 25, 0.85, false, 20},
        // gps_l2_m_dll_pll_tracking.cc:47-170: one 20 ms code per symbol, extension forced to 1
        {"GPS_L2_M_DLL_PLL_Tracking_HIP", 'G', "2S", 0.5115e6, 10230.0, 2.0f, 0.75f, 2.0f, 0.75f, 0.5f, 0.5f, false, 25, 0.85, false, 1},
        // gps_l5_dll_pll_tracking.cc:47-200: data component L5I (NH10) or pilot L5Q (NH20)
        {"GPS_L5_DLL_PLL_Tracking_HIP", 'G', "L5", 10.23e6, 10230.0, 50.0f, 2.0f, 2.0f, 0.25f, 0.5f, 0.15f, false, 25, 0.75, true, 10},
        // galileo_e5a_dll_pll_tracking.cc:47-200: data component E5a-I (CS20) or pilot E5a-Q (CS100 per PRN)
        {"Galileo_E5a_DLL_PLL_Tracking_HIP", 'E', "5X", 10.23e6, 10230.0, 20.0f, 20.0f, 5.0f, 2.0f, 0.5f, 0.15f, false, 25, 0.85, true, 20},
        // beidou_b3i_dll_pll_tracking.cc:47-190
        {"BEIDOU_B3I_DLL_PLL_Tracking_HIP", 'C', "B3", 10.23e6, 10230.0, 50.0f, 2.0f, 20.0f, 2.0f, 0.5f, 0.5f, false, 25, 0.85, false, 20}};
    return t[static_cast<int>(s)];
}
}  // namespace gnsscorr

//! Block: hip_dll_pll_veml_tracking (one level-1 correlator call per code period, host loop maths) or
//! hip_dll_pll_veml_tracking_dev (every complete code period of a work() call in one launch of the device loop)
template <gnsscorr::TrkSignal SIG, class Block = hip_dll_pll_veml_tracking>
class DllPllTrackingHip : public TrackingInterface
{
public:
    DllPllTrackingHip(ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
        : role_(role), in_streams_(in_streams), out_streams_(out_streams)
    {
        const gnsscorr::TrkSignalTraits& t = gnsscorr::trk_traits(SIG);
        Dll_Pll_Conf trk_param = Dll_Pll_Conf();
        int fs_in_deprecated = configuration->property("GNSS-SDR.internal_fs_hz", 2048000);
        int fs_in = configuration->property("GNSS-SDR.internal_fs_sps", fs_in_deprecated);
        trk_param.fs_in = fs_in;
        trk_param.high_dyn = configuration->property(role + ".high_dyn", false);
        trk_param.dump = configuration->property(role + ".dump", false);
        trk_param.dump_filename = configuration->property(role + ".dump_filename", std::string("./track_ch"));
        trk_param.dump_mat = configuration->property(role + ".dump_mat", true);
        trk_param.smoother_length = std::max(1, configuration->property(role + ".smoother_length", 10));
        trk_param.pll_bw_hz = configuration->property(role + ".pll_bw_hz", t.pll_bw_hz);
        trk_param.pll_bw_narrow_hz = configuration->property(role + ".pll_bw_narrow_hz", t.pll_bw_narrow_hz);
        trk_param.dll_bw_narrow_hz = configuration->property(role + ".dll_bw_narrow_hz", t.dll_bw_narrow_hz);
        trk_param.dll_bw_hz = configuration->property(role + ".dll_bw_hz", t.dll_bw_hz);
        trk_param.dll_filter_order = std::min(3, std::max(1, configuration->property(role + ".dll_filter_order", 2)));
        trk_param.pll_filter_order = std::min(3, std::max(2, configuration->property(role + ".pll_filter_order", 3)));
        trk_param.fll_filter_order = (trk_param.pll_filter_order == 2) ? 1 : 2;
        trk_param.enable_fll_pull_in = configuration->property(role + ".enable_fll_pull_in", false);
        trk_param.fll_bw_hz = configuration->property(role + ".fll_bw_hz", 35.0f);
        trk_param.pull_in_time_s = static_cast<unsigned int>(configuration->property(role + ".pull_in_time_s", 2.0f));
        trk_param.early_late_space_chips = configuration->property(role + ".early_late_space_chips", t.early_late_space_chips);
        trk_param.early_late_space_narrow_chips = configuration->property(role + ".early_late_space_narrow_chips", t.early_late_space_narrow_chips);
        // extend_correlation_symbols / track_pilot rules of the reference adapters (gps_l1_ca_dll_pll_tracking.cc:140-165,
        // galileo_e1_dll_pll_veml_tracking.cc:134-159, beidou_b1i_dll_pll_tracking.cc:128-151, gps_l2_m_dll_pll_tracking.cc:125-130,
        // gps_l5_dll_pll_tracking.cc:138-157, galileo_e5a_dll_pll_tracking.cc:137-156, beidou_b3i_dll_pll_tracking.cc:123-134)
        int extend_correlation_symbols = configuration->property(role + ".extend_correlation_symbols", 1);
        bool track_pilot = configuration->property(role + ".track_pilot", false);
        if (!t.has_pilot) track_pilot = false;
        if (extend_correlation_symbols < 1) extend_correlation_symbols = 1;
        if (!track_pilot)
            {
                // on the data component the integration may not cross a symbol: one telemetry bit (or one code: Galileo E1-B, L2C)
                const int longest = std::max(1, t.max_extension_data);
                if (extend_correlation_symbols > longest) extend_correlation_symbols = longest;
            }
        trk_param.extend_correlation_symbols = extend_correlation_symbols;
        if (t.veml)
            {
                trk_param.very_early_late_space_chips = configuration->property(role + ".very_early_late_space_chips", 0.6f);
                trk_param.very_early_late_space_narrow_chips = configuration->property(role + ".very_early_late_space_narrow_chips", 0.6f);
            }
        else
            {
                trk_param.very_early_late_space_chips = 0.0;
                trk_param.very_early_late_space_narrow_chips = 0.0;
            }
        trk_param.vector_length = std::round(static_cast<double>(fs_in) / (t.code_rate_hz / t.code_length_chips));
        trk_param.system = t.system;
        std::memcpy(trk_param.signal, t.signal, 3);
        trk_param.track_pilot = track_pilot;
        trk_param.cn0_samples = configuration->property(role + ".cn0_samples", 20);
        trk_param.cn0_min = configuration->property(role + ".cn0_min", t.cn0_min);
        trk_param.max_lock_fail = configuration->property(role + ".max_lock_fail", 50);
        trk_param.carrier_lock_th = configuration->property(role + ".carrier_lock_th", t.carrier_lock_th);
        conf_ = trk_param;
        tracking_ = std::make_shared<Block>(trk_param);
    }

    std::string role() override { return role_; }
    std::string implementation() override
    {
        const std::string name = gnsscorr::trk_traits(SIG).implementation;
        return std::is_same<Block, hip_dll_pll_veml_tracking>::value ? name : name + "_DEV";
    }
    size_t item_size() override { return sizeof(gr_complex); }

    void start_tracking() override { tracking_->start_tracking(); }
    void stop_tracking() override { tracking_->stop_tracking(); }
    void set_channel(unsigned int channel) override
    {
        channel_ = channel;
        tracking_->set_channel(channel);
    }
    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) override { tracking_->set_gnss_synchro(p_gnss_synchro); }

    //! the block (get_left_block()/get_right_block() in the reference)
    std::shared_ptr<Block> block() { return tracking_; }
    const Dll_Pll_Conf& conf() const { return conf_; }

private:
    std::shared_ptr<Block> tracking_;
    Dll_Pll_Conf conf_;
    std::string role_;
    unsigned int channel_ = 0;
    unsigned int in_streams_;
    unsigned int out_streams_;
};

using GpsL1CaDllPllTrackingHip = DllPllTrackingHip<gnsscorr::TrkSignal::GPS_L1_CA>;
using GalileoE1DllPllVemlTrackingHip = DllPllTrackingHip<gnsscorr::TrkSignal::GALILEO_E1>;
using BeidouB1iDllPllTrackingHip = DllPllTrackingHip<gnsscorr::TrkSignal::BEIDOU_B1I>;
using GpsL2MDllPllTrackingHip = DllPllTrackingHip<gnsscorr::TrkSignal::GPS_L2_M>;
using GpsL5DllPllTrackingHip = DllPllTrackingHip<gnsscorr::TrkSignal::GPS_L5>;
using GalileoE5aDllPllTrackingHip = DllPllTrackingHip<gnsscorr::TrkSignal::GALILEO_E5A>;
using BeidouB3iDllPllTrackingHip = DllPllTrackingHip<gnsscorr::TrkSignal::BEIDOU_B3I>;
//! the same adapters on the device loop ("..._HIP_DEV")
using GpsL1CaDllPllTrackingHipDev = DllPllTrackingHip<gnsscorr::TrkSignal::GPS_L1_CA, hip_dll_pll_veml_tracking_dev>;
using GalileoE1DllPllVemlTrackingHipDev = DllPllTrackingHip<gnsscorr::TrkSignal::GALILEO_E1, hip_dll_pll_veml_tracking_dev>;
using BeidouB1iDllPllTrackingHipDev = DllPllTrackingHip<gnsscorr::TrkSignal::BEIDOU_B1I, hip_dll_pll_veml_tracking_dev>;
using GpsL2MDllPllTrackingHipDev = DllPllTrackingHip<gnsscorr::TrkSignal::GPS_L2_M, hip_dll_pll_veml_tracking_dev>;
using GpsL5DllPllTrackingHipDev = DllPllTrackingHip<gnsscorr::TrkSignal::GPS_L5, hip_dll_pll_veml_tracking_dev>;
using GalileoE5aDllPllTrackingHipDev = DllPllTrackingHip<gnsscorr::TrkSignal::GALILEO_E5A, hip_dll_pll_veml_tracking_dev>;
using BeidouB3iDllPllTrackingHipDev = DllPllTrackingHip<gnsscorr::TrkSignal::BEIDOU_B3I, hip_dll_pll_veml_tracking_dev>;

#endif  // GNSSCORR_DLL_PLL_TRACKING_ADAPTERS_H_
