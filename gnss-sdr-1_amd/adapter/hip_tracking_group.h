/*!
 * \file hip_tracking_group.h
 * \brief All tracking channels of one signal on one GPU as ONE object: the MI355X-native shape of the receiver's tracking stage.
 *
 * In the reference every channel is its own GNU Radio block reading the same RF stream (gnss_flowgraph.cc:496-499), and every block
 * runs its own loop one code period at a time.  Here the stream lives once in an HBM ring (gc_stream) and a group holds N channel
 * slots of a device loop (gc_trk_loop): start_tracking(ch, acquisition result) fills a slot the way
 * dll_pll_veml_tracking::start_tracking does, run() executes every code period that is complete in the ring for ALL active
 * channels in one launch and returns the Gnss_Synchro items per channel -- what N blocks of the reference would have written to
 * their N outputs.  Slots that are not tracking (never started, stopped, lost lock) cost nothing.  A hybrid receiver uses one group
 * per signal on the same ring (tap count and pilot mode are per group).
 *
 * In a GNSS-SDR tree this is one gr::block with one input and N Gnss_Synchro outputs: general_work pushes the new items
 * (gc_stream_push), calls run() and copies each channel's items to its output (produce(ch, n)); the channel state machine calls
 * start_tracking / stop_tracking through N thin TrackingInterface adapters that hold (group, slot).
 */
#ifndef GNSSCORR_HIP_TRACKING_GROUP_H_
#define GNSSCORR_HIP_TRACKING_GROUP_H_

#include "hip_dll_pll_veml_tracking_dev.h"
#include <memory>
#include <mutex>
#include <vector>

class hip_tracking_group
{
public:
    /*! ctx / ring: the GPU context and the RF stream the channels read; conf: the Dll_Pll_Conf every channel of the group shares
     *  (signal, loop settings); n_channels: slots; iq_format: the ring's sample format (GC_IQ_F32 gr_complex, GC_IQ_I16 cshort,
     *  GC_IQ_I8 cbyte -- integer samples are converted on load, the stream crosses PCIe and sits in HBM in its native size) */
    hip_tracking_group(gc_ctx* ctx, gc_stream* ring, const Dll_Pll_Conf& conf, int n_channels, int iq_format = GC_IQ_F32)
        : trk_parameters(conf), d_ring(ring), d_n(n_channels)
    {
        if (!gnsscorr::trk_signal_constants(trk_parameters.system, std::string(trk_parameters.signal), &d_sig))
            {
                d_status = GC_ERR_INVALID;
                return;
            }
        if (!d_sig.has_pilot) trk_parameters.track_pilot = false;
        d_status = gc_trk_loop_create(ctx, n_channels, static_cast<int>(d_sig.code_length_chips * d_sig.code_samples_per_chip), &d_loop);
        if (d_status == GC_OK) d_status = gc_trk_loop_set_input_format(d_loop, iq_format);
        for (int ch = 0; ch < n_channels && d_status == GC_OK; ch++) d_status = gc_trk_loop_set_input_stream(d_loop, ch, ring);
        d_acq.resize(n_channels);
        d_active.assign(n_channels, 0);
        d_position.assign(n_channels, 0);
        d_events.resize(n_channels);
        d_interchange_iq = trk_parameters.track_pilot && d_sig.interchange_iq_with_pilot;
    }
    ~hip_tracking_group()
    {
        if (d_loop) gc_trk_loop_destroy(d_loop);
    }
    hip_tracking_group(const hip_tracking_group&) = delete;
    hip_tracking_group& operator=(const hip_tracking_group&) = delete;

    //! the reference waits 10 s before looking for the telemetry preamble (dll_pll_veml_tracking.cc:1648); tests shorten it
    void set_bit_sync_min_time_s(float t) { d_bit_sync_min_time_s = t; }

    /*! Channel slot `ch` starts tracking the satellite of `acq` (PRN, Acq_delay_samples, Acq_doppler_hz, Acq_samplestamp_samples) at
     *  stream sample `start_index` -- any resident sample; the pull-in aligns the first code period after it. */
    gc_status start_tracking(int ch, const Gnss_Synchro& acq, uint64_t start_index)
    {
        std::lock_guard<std::mutex> l(d_mutex);
        if (d_loop == nullptr) return d_status;  // construction failed
        if (ch < 0 || ch >= d_n) return GC_ERR_INVALID;
        if (d_active[ch]) gc_trk_loop_stop(d_loop, ch);
        gc_status st = gnsscorr::loop_start_channel(d_loop, ch, trk_parameters, d_sig, acq, start_index, d_bit_sync_min_time_s);
        if (st != GC_OK) return st;
        d_acq[ch] = acq;
        d_active[ch] = 1;
        d_position[ch] = start_index;
        return GC_OK;
    }

    gc_status stop_tracking(int ch)
    {
        std::lock_guard<std::mutex> l(d_mutex);
        if (ch < 0 || ch >= d_n) return GC_ERR_INVALID;
        d_active[ch] = 0;
        return gc_trk_loop_stop(d_loop, ch);
    }

    /*! Every complete code period of every active channel, one launch; out[ch] receives the channel's Gnss_Synchro items (appended).
     *  Returns the number of items produced over all channels, or -1 (last_status()). */
    int run(std::vector<std::vector<Gnss_Synchro>>& out)
    {
        std::lock_guard<std::mutex> l(d_mutex);
        if (d_loop == nullptr) return -1;  // construction failed; a failed run() is NOT sticky: the next call tries again
        out.resize(d_n);
        uint64_t head = 0;
        gc_stream_info(d_ring, nullptr, &head, nullptr);
        int n_periods = 0;
        bool any = false;
        for (int ch = 0; ch < d_n; ch++)
            if (d_active[ch])
                {
                    any = true;
                    if (head > d_position[ch]) n_periods = std::max<int>(n_periods, static_cast<int>((head - d_position[ch]) / trk_parameters.vector_length) + 1);
                }
        if (!any || n_periods == 0) return 0;
        d_records.resize(static_cast<size_t>(d_n) * n_periods);
        d_status = gc_trk_loop_run(d_loop, n_periods, d_records.data());
        if (d_status != GC_OK)
            {
                // e.g. a channel that was not serviced in time fell behind the ring (GC_ERR_STATE): the caller parks that channel
                // (stop_tracking / a new start_tracking) and goes on; the other channels are unaffected
                return -1;
            }
        int produced = 0;
        for (int ch = 0; ch < d_n; ch++)
            {
                if (!d_active[ch]) continue;
                uint64_t position = d_position[ch];
                for (int k = 0; k < n_periods; k++)
                    {
                        const gc_loop_record& r = d_records[static_cast<size_t>(ch) * n_periods + k];
                        if (r.sample_counter == position && r.valid == 0 && r.state >= 2) break;  // out of input
                        position = r.sample_counter;
                        if (r.valid)
                            {
                                out[ch].push_back(gnsscorr::synchro_from_record(r, d_acq[ch], d_sig, d_interchange_iq, trk_parameters.fs_in));
                                produced++;
                            }
                        if (r.state == 0)
                            {
                                d_events[ch].push_back(3);  // loss of lock: the slot is free for the next acquisition
                                d_active[ch] = 0;
                                break;
                            }
                    }
                d_position[ch] = position;
            }
        return produced;
    }

    int n_channels() const { return d_n; }
    bool active(int ch) const { return d_active[ch] != 0; }
    //! stream sample up to which channel `ch` has consumed (a producer may evict everything before the minimum over the channels)
    uint64_t position(int ch) const { return d_position[ch]; }
    const std::vector<int>& events(int ch) const { return d_events[ch]; }
    gc_status last_status() const { return d_status; }

private:
    Dll_Pll_Conf trk_parameters;
    gnsscorr::TrkSignalConstants d_sig{};
    gc_stream* d_ring;
    gc_trk_loop* d_loop = nullptr;
    int d_n;
    bool d_interchange_iq = false;
    float d_bit_sync_min_time_s = 10.0f;
    gc_status d_status = GC_OK;
    std::mutex d_mutex;
    std::vector<Gnss_Synchro> d_acq;
    std::vector<char> d_active;
    std::vector<uint64_t> d_position;
    std::vector<std::vector<int>> d_events;
    std::vector<gc_loop_record> d_records;
};

#endif  // GNSSCORR_HIP_TRACKING_GROUP_H_
