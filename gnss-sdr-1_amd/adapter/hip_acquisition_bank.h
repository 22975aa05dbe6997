/*!
 * \file hip_acquisition_bank.h
 * \brief The acquisition stage of a whole constellation as ONE object on the RF stream ring: the counterpart of hip_tracking_group.
 *
 * In the reference every channel owns a pcps_acquisition block that searches ONE satellite at a time on its copy of the input
 * (channel.cc:59-115, pcps_acquisition.cc:668-927).  Here one engine (gc_acq) holds the replicas of every PRN of a signal, and a
 * search(first_index) call runs the same PCPS search -- wipe-off, FFT, x conj(FFT(code)), IFFT, |.|^2, statistic -- for ALL of them on
 * the samples already resident in the ring (the Doppler wipe-off and the forward FFT of the input are computed once per bin and
 * shared by the satellites), and returns a Gnss_Synchro with the Acq_* fields of every satellite above the threshold, ready for
 * hip_tracking_group::start_tracking.  Sizes, thresholds and the result mapping are those of the reference adapters
 * (pcps_acquisition_adapters.h); GPS L1 C/A, L5I, Galileo E1 B / C, E5a, BeiDou B1I, B3I.
 */
#ifndef GNSSCORR_HIP_ACQUISITION_BANK_H_
#define GNSSCORR_HIP_ACQUISITION_BANK_H_

#include "gnss_sdr_types.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

class hip_acquisition_bank
{
public:
    /*! system / signal: 'G' "1C" | "L5", 'E' "1B" | "5X", 'C' "B1" | "B3"; prns: the satellites searched; fs_in: the ring's sampling rate;
     *  doppler_max / doppler_step [Hz]; threshold on the block's statistic (pcps_acquisition.cc:565-665); max_dwells non-coherent dwells */
    hip_acquisition_bank(gc_ctx* ctx, gc_stream* ring, char system, const std::string& signal, const std::vector<uint32_t>& prns, int64_t fs_in, uint32_t doppler_max,
        uint32_t doppler_step, float threshold, uint32_t max_dwells = 1, bool use_cfar = true, int iq_format = GC_IQ_F32)
        : d_ring(ring), d_system(system), d_signal(signal), d_prns(prns), d_threshold(threshold), d_max_dwells(std::max(1u, max_dwells))
    {
        double code_rate = 1.023e6, code_len = 1023.0;
        uint32_t ms_per_code = 1;
        if (system == 'G' && signal == "1C") {}
        else if (system == 'G' && signal == "L5") { code_rate = 10.23e6; code_len = 10230.0; }
        else if (system == 'E' && signal == "1B") { code_len = 4092.0; ms_per_code = 4; }
        else if (system == 'E' && signal == "5X") { code_rate = 1.023e7; code_len = 10230.0; }
        else if (system == 'C' && signal == "B1") { code_rate = 2.046e6; code_len = 2046.0; }
        else if (system == 'C' && signal == "B3") { code_rate = 10.23e6; code_len = 10230.0; }
        else
            {
                d_status = GC_ERR_INVALID;
                return;
            }
        gc_acq_conf c;
        std::memset(&c, 0, sizeof c);
        c.fs_in = fs_in;
        c.sampled_ms = ms_per_code;
        c.ms_per_code = ms_per_code;
        c.samples_per_ms = static_cast<float>(fs_in) * 0.001f;
        c.samples_per_code = c.samples_per_ms * static_cast<float>(ms_per_code);
        c.samples_per_chip = static_cast<uint32_t>(std::ceil((1.0 / code_rate) * static_cast<float>(fs_in)));
        if (system == 'C')
            {
                // the BeiDou adapters leave ms_per_code and samples_per_chip at Acq_Conf's zeros and count in code lengths
                // (beidou_b1i_pcps_acquisition.cc:73-104): the block then doubles its FFT (pcps_acquisition.cc:78-85)
                c.ms_per_code = 0;
                c.samples_per_chip = 0;
                c.samples_per_ms = static_cast<float>(std::round(static_cast<double>(fs_in) / (code_rate / code_len)));
                c.samples_per_code = c.samples_per_ms;
            }
        c.doppler_max = doppler_max;
        c.doppler_step = doppler_step;
        c.max_dwells = d_max_dwells;
        c.use_CFAR_algorithm_flag = use_cfar ? 1 : 0;
        d_status = gc_acq_create(ctx, &c, static_cast<int>(prns.size()), &d_acq);
        if (d_status == GC_OK && iq_format != GC_IQ_F32) d_status = gc_acq_set_input_format(d_acq, iq_format);
        if (d_status != GC_OK) return;
        uint32_t fft = 0, consumed = 0, bins = 0;
        gc_acq_fft_size(d_acq, &fft, &consumed, &bins);
        d_consumed = consumed;
        d_samples_per_code = static_cast<uint32_t>(std::floor(static_cast<double>(fs_in) / (code_rate / code_len)));
        // replicas: one code period at fs, tiled over the coherent time (the adapters' set_local_code)
        std::vector<float> one(2 * (d_samples_per_code + 16)), tiled(2 * static_cast<size_t>(consumed));
        for (size_t s = 0; s < prns.size() && d_status == GC_OK; s++)
            {
                const int32_t fs = static_cast<int32_t>(fs_in);
                if (system == 'G' && signal == "1C") d_status = gc_gps_l1_ca_code_gen_complex_sampled(one.data(), prns[s], fs, 0, nullptr);
                else if (system == 'G') d_status = gc_gps_l5i_code_gen_complex_sampled(one.data(), prns[s], fs, nullptr);
                else if (system == 'E' && signal == "1B") d_status = gc_galileo_e1_code_gen_complex_sampled(one.data(), "1B", 0, prns[s], fs, 0, nullptr);
                else if (system == 'E') d_status = gc_galileo_e5_a_code_gen_complex_sampled(one.data(), "5X", prns[s], fs, 0, nullptr);
                else if (signal == "B1") d_status = gc_beidou_b1i_code_gen_complex_sampled(one.data(), prns[s], fs, 0, nullptr);
                else d_status = gc_beidou_b3i_code_gen_complex_sampled(one.data(), prns[s], fs, 0, nullptr);
                if (d_status != GC_OK) return;
                for (size_t i = 0; i < consumed; i++)
                    {
                        tiled[2 * i] = one[2 * (i % d_samples_per_code)];
                        tiled[2 * i + 1] = one[2 * (i % d_samples_per_code) + 1];
                    }
                d_status = gc_acq_set_local_code(d_acq, static_cast<int>(s), tiled.data());
            }
        d_results.resize(prns.size());
        d_ready = (d_status == GC_OK);
    }
    ~hip_acquisition_bank()
    {
        if (d_acq) gc_acq_destroy(d_acq);
    }
    hip_acquisition_bank(const hip_acquisition_bank&) = delete;
    hip_acquisition_bank& operator=(const hip_acquisition_bank&) = delete;

    //! samples one dwell consumes (pcps_acquisition: d_consumed_samples)
    uint32_t consumed_samples() const { return d_consumed; }

    /*! One search of every satellite on ring samples [first_index, first_index + max_dwells * consumed_samples()): max_dwells
     *  non-coherent dwells, then the detections (statistic > threshold), strongest first.  Acq_samplestamp_samples is the stream
     *  index of the LAST dwell's first sample, as the block stamps it (pcps_acquisition.cc:768). */
    std::vector<Gnss_Synchro> search(uint64_t first_index)
    {
        std::vector<Gnss_Synchro> found;
        if (d_acq == nullptr || !d_ready) return found;  // construction failed; a failed search is NOT sticky
        d_status = gc_acq_reset(d_acq);
        uint64_t stamp = first_index;
        for (uint32_t dwell = 0; dwell < d_max_dwells && d_status == GC_OK; dwell++)
            {
                stamp = first_index + static_cast<uint64_t>(dwell) * d_consumed;
                d_status = gc_acq_dwell_stream(d_acq, d_ring, stamp, d_results.data());
            }
        if (d_status != GC_OK) return found;
        std::vector<size_t> order;
        for (size_t s = 0; s < d_prns.size(); s++)
            if (d_results[s].test_statistics > d_threshold) order.push_back(s);
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return d_results[a].test_statistics > d_results[b].test_statistics; });
        for (size_t s : order)
            {
                Gnss_Synchro g;
                g.System = d_system;
                g.Signal[0] = d_signal[0];
                g.Signal[1] = d_signal[1];
                g.PRN = d_prns[s];
                g.Acq_delay_samples = d_results[s].acq_delay_samples;
                g.Acq_doppler_hz = d_results[s].acq_doppler_hz;
                g.Acq_samplestamp_samples = stamp;
                g.Flag_valid_acquisition = true;
                found.push_back(g);
            }
        return found;
    }

    //! the statistic of every searched satellite in the last search (same order as `prns`)
    float statistic(size_t sat) const { return d_results[sat].test_statistics; }
    gc_status last_status() const { return d_status; }

private:
    gc_stream* d_ring;
    gc_acq* d_acq = nullptr;
    bool d_ready = false;  // construction went through (every replica installed)
    char d_system;
    std::string d_signal;
    std::vector<uint32_t> d_prns;
    float d_threshold;
    uint32_t d_max_dwells;
    uint32_t d_consumed = 0, d_samples_per_code = 0;
    gc_status d_status = GC_OK;
    std::vector<gc_acq_result> d_results;
};

#endif  // GNSSCORR_HIP_ACQUISITION_BANK_H_
