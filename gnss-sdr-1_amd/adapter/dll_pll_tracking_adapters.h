/*!
 * \file dll_pll_tracking_adapters.h
 * \brief TrackingInterface adapters for GPS L1 C/A, Galileo E1 and BeiDou B1I backed by
 * hip_dll_pll_veml_tracking.  Configuration keys and defaults follow
 *   GpsL1CaDllPllTracking        src/algorithms/tracking/adapters/gps_l1_ca_dll_pll_tracking.cc:47-214
 *   GalileoE1DllPllVemlTracking  src/algorithms/tracking/adapters/galileo_e1_dll_pll_veml_tracking.cc:47-215
 *   BeidouB1iDllPllTracking      src/algorithms/tracking/adapters/beidou_b1i_dll_pll_tracking.cc:47-200
 * (gflags overrides, dump files and the GNU Radio connect()/get_left_block() plumbing are outside this path).
 * Registration: `else if (implementation == "GPS_L1_CA_DLL_PLL_Tracking_HIP")` in
 * GNSSBlockFactory::GetTrkBlock (src/core/receiver/gnss_block_factory.cc:2131), see INTEGRATION.md.
 */
#ifndef GNSSCORR_DLL_PLL_TRACKING_ADAPTERS_H_
#define GNSSCORR_DLL_PLL_TRACKING_ADAPTERS_H_

#include "hip_dll_pll_veml_tracking.h"
#include <algorithm>
#include <cmath>
#include <memory>
#include <string>

namespace gnsscorr
{
enum class TrkSignal
{
    GPS_L1_CA,
    GALILEO_E1,
    BEIDOU_B1I
};
}

template <gnsscorr::TrkSignal SIG>
class DllPllTrackingHip : public TrackingInterface
{
public:
    DllPllTrackingHip(ConfigurationInterface* configuration, const std::string& role, unsigned int in_streams, unsigned int out_streams)
        : role_(role), in_streams_(in_streams), out_streams_(out_streams)
    {
        using gnsscorr::TrkSignal;
        Dll_Pll_Conf trk_param = Dll_Pll_Conf();
        int fs_in_deprecated = configuration->property("GNSS-SDR.internal_fs_hz", 2048000);
        int fs_in = configuration->property("GNSS-SDR.internal_fs_sps", fs_in_deprecated);
        trk_param.fs_in = fs_in;
        trk_param.high_dyn = configuration->property(role + ".high_dyn", false);
        trk_param.dump = configuration->property(role + ".dump", false);
        trk_param.dump_filename = configuration->property(role + ".dump_filename", std::string("./track_ch"));
        trk_param.dump_mat = configuration->property(role + ".dump_mat", true);
        trk_param.smoother_length = std::max(1, configuration->property(role + ".smoother_length", 10));
        const bool gal = SIG == TrkSignal::GALILEO_E1;
        trk_param.pll_bw_hz = configuration->property(role + ".pll_bw_hz", gal ? 5.0f : 50.0f);
        trk_param.pll_bw_narrow_hz = configuration->property(role + ".pll_bw_narrow_hz", gal ? 2.0f : 20.0f);
        trk_param.dll_bw_narrow_hz = configuration->property(role + ".dll_bw_narrow_hz", gal ? 0.25f : 2.0f);
        trk_param.dll_bw_hz = configuration->property(role + ".dll_bw_hz", gal ? 0.5f : 2.0f);
        trk_param.dll_filter_order = std::min(3, std::max(1, configuration->property(role + ".dll_filter_order", 2)));
        trk_param.pll_filter_order = std::min(3, std::max(2, configuration->property(role + ".pll_filter_order", 3)));
        trk_param.fll_filter_order = (trk_param.pll_filter_order == 2) ? 1 : 2;
        trk_param.enable_fll_pull_in = configuration->property(role + ".enable_fll_pull_in", false);
        trk_param.fll_bw_hz = configuration->property(role + ".fll_bw_hz", 35.0f);
        trk_param.pull_in_time_s = static_cast<unsigned int>(configuration->property(role + ".pull_in_time_s", 2.0f));
        trk_param.early_late_space_chips = configuration->property(role + ".early_late_space_chips", gal ? 0.15f : 0.5f);
        trk_param.early_late_space_narrow_chips = configuration->property(role + ".early_late_space_narrow_chips", gal ? 0.15f : 0.5f);
        // extend_correlation_symbols / track_pilot rules of the reference adapters (gps_l1_ca_dll_pll_tracking.cc:140-165,
        // galileo_e1_dll_pll_veml_tracking.cc:134-159, beidou_b1i_dll_pll_tracking.cc:128-151)
        int extend_correlation_symbols = configuration->property(role + ".extend_correlation_symbols", 1);
        bool track_pilot = configuration->property(role + ".track_pilot", false);
        if (extend_correlation_symbols < 1) extend_correlation_symbols = 1;
        if (gal)
            {
                // extended integration needs the pilot: the data component has a symbol transition every code period
                if (!track_pilot && extend_correlation_symbols > 1) extend_correlation_symbols = 1;
            }
        else
            {
                if (extend_correlation_symbols > 20) extend_correlation_symbols = 20;  // one telemetry bit
                track_pilot = false;  // GPS L1 C/A and BeiDou B1I have no pilot component
            }
        trk_param.extend_correlation_symbols = extend_correlation_symbols;
        if (gal)
            {
                trk_param.very_early_late_space_chips = configuration->property(role + ".very_early_late_space_chips", 0.6f);
                trk_param.very_early_late_space_narrow_chips = configuration->property(role + ".very_early_late_space_narrow_chips", 0.6f);
                trk_param.vector_length = std::round(fs_in / (1.023e6 / 4092.0));
                trk_param.system = 'E';
                std::memcpy(trk_param.signal, "1B", 3);
            }
        else if (SIG == TrkSignal::BEIDOU_B1I)
            {
                trk_param.very_early_late_space_chips = 0.0;
                trk_param.very_early_late_space_narrow_chips = 0.0;
                trk_param.vector_length = std::round(fs_in / (2.046e6 / 2046.0));
                trk_param.system = 'C';
                std::memcpy(trk_param.signal, "B1", 3);
            }
        else
            {
                trk_param.very_early_late_space_chips = 0.0;
                trk_param.very_early_late_space_narrow_chips = 0.0;
                trk_param.vector_length = std::round(fs_in / (1.023e6 / 1023.0));
                trk_param.system = 'G';
                std::memcpy(trk_param.signal, "1C", 3);
            }
        trk_param.track_pilot = track_pilot;
        trk_param.cn0_samples = configuration->property(role + ".cn0_samples", 20);
        trk_param.cn0_min = configuration->property(role + ".cn0_min", SIG == TrkSignal::GPS_L1_CA ? 30 : 25);
        trk_param.max_lock_fail = configuration->property(role + ".max_lock_fail", 50);
        trk_param.carrier_lock_th = configuration->property(role + ".carrier_lock_th", SIG == TrkSignal::GPS_L1_CA ? 0.80 : 0.85);
        conf_ = trk_param;
        tracking_ = std::make_shared<hip_dll_pll_veml_tracking>(trk_param);
    }

    std::string role() override { return role_; }
    std::string implementation() override
    {
        return SIG == gnsscorr::TrkSignal::GPS_L1_CA ? "GPS_L1_CA_DLL_PLL_Tracking_HIP" : SIG == gnsscorr::TrkSignal::GALILEO_E1 ? "Galileo_E1_DLL_PLL_VEML_Tracking_HIP" : "BEIDOU_B1I_DLL_PLL_Tracking_HIP";
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
    std::shared_ptr<hip_dll_pll_veml_tracking> block() { return tracking_; }
    const Dll_Pll_Conf& conf() const { return conf_; }

private:
    std::shared_ptr<hip_dll_pll_veml_tracking> tracking_;
    Dll_Pll_Conf conf_;
    std::string role_;
    unsigned int channel_ = 0;
    unsigned int in_streams_;
    unsigned int out_streams_;
};

using GpsL1CaDllPllTrackingHip = DllPllTrackingHip<gnsscorr::TrkSignal::GPS_L1_CA>;
using GalileoE1DllPllVemlTrackingHip = DllPllTrackingHip<gnsscorr::TrkSignal::GALILEO_E1>;
using BeidouB1iDllPllTrackingHip = DllPllTrackingHip<gnsscorr::TrkSignal::BEIDOU_B1I>;

#endif  // GNSSCORR_DLL_PLL_TRACKING_ADAPTERS_H_
