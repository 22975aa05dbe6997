/*!
 * \file hip_pcps_acquisition.h
 * \brief Image of the pcps_acquisition block (src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.{h,cc})
 * with the Doppler/FFT search executed by libgnsscorr.so on an MI355X.
 *
 * Public members, their meaning and the state machine follow the reference block:
 *   set_gnss_synchro / mag / init / set_local_code / set_active / set_state / set_channel /
 *   set_channel_fsm / set_threshold / set_doppler_max / set_doppler_step      (pcps_acquisition.h:163-252)
 *   general_work()   -> work(in, n_items): states 0 (reset) / 1 (buffer d_consumed_samples) / 2 (search),
 *                       returns the number of items consumed                   (pcps_acquisition.cc:941-1054)
 *   acquisition_core -> one engine dwell + the decision logic                  (pcps_acquisition.cc:668-927)
 * The block is not a gr::block (there is no GNU Radio here): the scheduler's call of
 * general_work(noutput, ninput_items, input_items, output_items) becomes work(in, ninput_items[0]),
 * consume_each(n) becomes the return value, and the "events" message port becomes events().
 * Inside a GNSS-SDR tree a 30-line gr::block shell forwards to this class.
 */
#ifndef GNSSCORR_HIP_PCPS_ACQUISITION_H_
#define GNSSCORR_HIP_PCPS_ACQUISITION_H_

#include "gnss_sdr_types.h"
#include "gnsscorr.h"
#include "hip_multicorrelator_real_codes.h"  // gnsscorr::shared_context()
#include "mat5_writer.h"
#include <cmath>
#include <cstring>
#include <iostream>
#include <map>
#include <mutex>
#include <string>
#include <sys/stat.h>
#include <vector>

class hip_pcps_acquisition
{
public:
    explicit hip_pcps_acquisition(const Acq_Conf& conf_) : acq_parameters(conf_)
    {
        // sizes exactly as the reference constructor (pcps_acquisition.cc:77-117, :152-159)
        d_consumed_samples = acq_parameters.sampled_ms * acq_parameters.samples_per_ms * (acq_parameters.bit_transition_flag ? 2 : 1);
        d_fft_size = (acq_parameters.sampled_ms == acq_parameters.ms_per_code) ? d_consumed_samples : d_consumed_samples * 2;
        if (acq_parameters.bit_transition_flag)
            {
                d_fft_size = d_consumed_samples * 2;
                acq_parameters.max_dwells = 1;
            }
        d_use_CFAR_algorithm_flag = (acq_parameters.max_dwells == 1) ? acq_parameters.use_CFAR_algorithm_flag : false;
        d_num_doppler_bins_step2 = acq_parameters.num_doppler_bins_step2;
        d_data_buffer.resize(d_consumed_samples);
        // dump file name handling of the reference constructor (:160-193)
        d_dump_channel = acq_parameters.dump_channel;
        d_dump = acq_parameters.dump;
        d_dump_filename = acq_parameters.dump_filename;
        if (d_dump)
            {
                std::string dump_path;
                if (d_dump_filename.find_last_of('/') != std::string::npos)
                    {
                        const std::string dump_filename_ = d_dump_filename.substr(d_dump_filename.find_last_of('/') + 1);
                        dump_path = d_dump_filename.substr(0, d_dump_filename.find_last_of('/'));
                        d_dump_filename = dump_filename_;
                    }
                else
                    {
                        dump_path = std::string(".");
                    }
                if (d_dump_filename.empty()) d_dump_filename = "acquisition";
                // remove extension if any
                if (d_dump_filename.substr(1).find_last_of('.') != std::string::npos) d_dump_filename = d_dump_filename.substr(0, d_dump_filename.find_last_of('.'));
                d_dump_filename = dump_path + '/' + d_dump_filename;
                if (!create_directory(dump_path))
                    {
                        std::cerr << "GNSS-SDR cannot create dump file for the Acquisition block. Wrong permissions?" << std::endl;
                        d_dump = false;
                    }
            }
    }

    ~hip_pcps_acquisition()
    {
        if (d_acq != nullptr) gc_acq_destroy(d_acq);
    }

    hip_pcps_acquisition(const hip_pcps_acquisition&) = delete;
    hip_pcps_acquisition& operator=(const hip_pcps_acquisition&) = delete;

    inline void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro)
    {
        std::lock_guard<std::mutex> lock(d_setlock);
        d_gnss_synchro = p_gnss_synchro;
    }

    inline uint32_t mag() const { return d_mag; }

    /*! pcps_acquisition::init (:313-368): clears the synchro fields, fixes the Doppler grid and
     *  (re)creates the engine plan, whose wipe-off table is built like update_local_carrier does. */
    void init()
    {
        d_gnss_synchro->Flag_valid_acquisition = false;
        d_gnss_synchro->Flag_valid_symbol_output = false;
        d_gnss_synchro->Flag_valid_pseudorange = false;
        d_gnss_synchro->Flag_valid_word = false;
        d_gnss_synchro->Acq_doppler_step = 0U;
        d_gnss_synchro->Acq_delay_samples = 0.0;
        d_gnss_synchro->Acq_doppler_hz = 0.0;
        d_gnss_synchro->Acq_samplestamp_samples = 0ULL;
        d_mag = 0.0;
        d_input_power = 0.0;
        d_num_doppler_bins = static_cast<uint32_t>(std::ceil(static_cast<double>(static_cast<int32_t>(acq_parameters.doppler_max) - static_cast<int32_t>(-acq_parameters.doppler_max)) / static_cast<double>(d_doppler_step)));
        if (d_acq != nullptr)
            {
                gc_acq_destroy(d_acq);
                d_acq = nullptr;
            }
        gc_acq_conf c;
        std::memset(&c, 0, sizeof c);
        c.fs_in = acq_parameters.use_automatic_resampler ? acq_parameters.resampled_fs : acq_parameters.fs_in;
        c.sampled_ms = acq_parameters.sampled_ms;
        c.ms_per_code = acq_parameters.ms_per_code;
        c.samples_per_ms = acq_parameters.samples_per_ms;
        c.samples_per_code = acq_parameters.samples_per_code;
        c.samples_per_chip = acq_parameters.samples_per_chip;
        c.doppler_max = acq_parameters.doppler_max;
        c.doppler_step = d_doppler_step;
        c.max_dwells = acq_parameters.max_dwells;
        c.bit_transition_flag = acq_parameters.bit_transition_flag ? 1 : 0;
        c.use_CFAR_algorithm_flag = acq_parameters.use_CFAR_algorithm_flag ? 1 : 0;
        c.make_2_steps = acq_parameters.make_2_steps ? 1 : 0;
        c.num_doppler_bins_step2 = acq_parameters.num_doppler_bins_step2;
        c.doppler_step2 = acq_parameters.doppler_step2;
        gc_ctx* ctx = gnsscorr::shared_context();
        d_status = ctx ? gc_acq_create(ctx, &c, 1, &d_acq) : GC_ERR_NO_DEVICE;
        if (d_status == GC_OK && d_old_freq != 0) d_status = gc_acq_set_frequency_offset(d_acq, d_old_freq);
        if (d_status == GC_OK && !d_code.empty()) d_status = gc_acq_set_local_code(d_acq, 0, reinterpret_cast<const float*>(d_code.data()));
        d_worker_active = false;
        if (d_dump)
            {
                // (:364-369)
                const uint32_t effective_fft_size = (acq_parameters.bit_transition_flag ? (d_fft_size / 2) : d_fft_size);
                grid_.assign(static_cast<size_t>(effective_fft_size) * d_num_doppler_bins, 0.0f);
                narrow_grid_.assign(static_cast<size_t>(effective_fft_size) * d_num_doppler_bins_step2, 0.0f);
            }
    }

    /*! GLONASS slot -> frequency channel number, what GLONASS_PRN holds in the reference
     *  (src/core/system_parameters/GLONASS_L1_L2_CA.h:129); almanac data the receiver provides.  With it installed,
     *  set_local_code() applies the FDMA offset of is_fdma() (:276-293) for "1G" / "2G" signals. */
    void set_glonass_channel_map(const std::map<uint32_t, int32_t>& prn_to_channel) { d_glonass_prn = prn_to_channel; }

    /*! pcps_acquisition::set_local_code (:239-274): code = d_consumed_samples complex (fft_size/2 with bit transition) */
    void set_local_code(std::complex<float>* code)
    {
        std::lock_guard<std::mutex> lock(d_setlock);
        // reset the intermediate frequency, then the FDMA check (:242-247): DFRQ1_GLO = 562500 Hz, DFRQ2_GLO = 437500 Hz per channel
        d_old_freq = 0;
        if (d_gnss_synchro != nullptr && d_gnss_synchro->Signal[1] == 'G' && (d_gnss_synchro->Signal[0] == '1' || d_gnss_synchro->Signal[0] == '2'))
            {
                const auto it = d_glonass_prn.find(d_gnss_synchro->PRN);
                if (it != d_glonass_prn.end()) d_old_freq += static_cast<int64_t>(d_gnss_synchro->Signal[0] == '1' ? 562500 : 437500) * it->second;
            }
        if (d_acq != nullptr) d_status = gc_acq_set_frequency_offset(d_acq, d_old_freq);
        const size_t n = acq_parameters.bit_transition_flag ? d_fft_size / 2 : d_consumed_samples;
        d_code.assign(code, code + n);
        // (a failed frequency-offset call must stay visible in last_status(): the search would run on the old wipe-off tables)
        if (d_acq != nullptr && d_status == GC_OK) d_status = gc_acq_set_local_code(d_acq, 0, reinterpret_cast<const float*>(d_code.data()));
    }

    inline void set_active(bool active)
    {
        std::lock_guard<std::mutex> lock(d_setlock);
        d_active = active;
    }

    void set_state(int32_t state)
    {
        std::lock_guard<std::mutex> lock(d_setlock);
        d_state = state;
        if (d_state == 1)
            {
                d_gnss_synchro->Acq_delay_samples = 0.0;
                d_gnss_synchro->Acq_doppler_hz = 0.0;
                d_gnss_synchro->Acq_samplestamp_samples = 0ULL;
                d_gnss_synchro->Acq_doppler_step = 0U;
                d_mag = 0.0;
                d_input_power = 0.0;
                d_test_statistics = 0.0;
                d_active = true;
            }
    }

    inline void set_channel(uint32_t channel) { d_channel = channel; }
    inline void set_channel_fsm(std::shared_ptr<ChannelFsm> channel_fsm) { d_channel_fsm = channel_fsm; }

    inline void set_threshold(float threshold)
    {
        std::lock_guard<std::mutex> lock(d_setlock);
        d_threshold = threshold;
    }

    inline void set_doppler_max(uint32_t doppler_max)
    {
        std::lock_guard<std::mutex> lock(d_setlock);
        acq_parameters.doppler_max = doppler_max;
    }

    inline void set_doppler_step(uint32_t doppler_step)
    {
        std::lock_guard<std::mutex> lock(d_setlock);
        d_doppler_step = doppler_step;
    }

    void set_resampler_latency(uint32_t latency_samples)
    {
        std::lock_guard<std::mutex> lock(d_setlock);
        acq_parameters.resampler_latency_samples = latency_samples;
    }

    /*! general_work (:941-1054). Returns the number of input items consumed. */
    int work(const gr_complex* in, int ninput_items)
    {
        std::unique_lock<std::mutex> lk(d_setlock);
        int consumed = 0;
        if (!d_active or d_worker_active)
            {
                if (!acq_parameters.blocking_on_standby)
                    {
                        d_sample_counter += static_cast<uint64_t>(ninput_items);
                        consumed = ninput_items;
                    }
                if (d_step_two)
                    {
                        d_doppler_center_step_two = static_cast<float>(d_gnss_synchro->Acq_doppler_hz);
                        if (d_acq != nullptr) d_status = gc_acq_set_step_two(d_acq, 1, d_doppler_center_step_two);
                        d_state = 0;
                        d_active = true;
                    }
                return consumed;
            }
        switch (d_state)
            {
            case 0:
                d_gnss_synchro->Acq_delay_samples = 0.0;
                d_gnss_synchro->Acq_doppler_hz = 0.0;
                d_gnss_synchro->Acq_samplestamp_samples = 0ULL;
                d_gnss_synchro->Acq_doppler_step = 0U;
                d_mag = 0.0;
                d_input_power = 0.0;
                d_test_statistics = 0.0;
                d_state = 1;
                d_buffer_count = 0U;
                if (!acq_parameters.blocking_on_standby)
                    {
                        d_sample_counter += static_cast<uint64_t>(ninput_items);
                        consumed = ninput_items;
                    }
                break;
            case 1:
                {
                    uint32_t buff_increment;
                    if ((ninput_items + d_buffer_count) <= d_consumed_samples)
                        buff_increment = ninput_items;
                    else
                        buff_increment = d_consumed_samples - d_buffer_count;
                    std::memcpy(&d_data_buffer[d_buffer_count], in, sizeof(gr_complex) * buff_increment);
                    // "If buffer will be full in next iteration" (:1029-1032): tested before the increment
                    if (d_buffer_count >= d_consumed_samples) d_state = 2;
                    d_buffer_count += buff_increment;
                    d_sample_counter += static_cast<uint64_t>(buff_increment);
                    consumed = static_cast<int>(buff_increment);
                    break;
                }
            case 2:
                lk.unlock();
                acquisition_core(d_sample_counter);
                lk.lock();
                d_buffer_count = 0U;
                break;
            }
        return consumed;
    }

    /*! acquisition_core (:668-927) */
    void acquisition_core(uint64_t samp_count)
    {
        std::unique_lock<std::mutex> lk(d_setlock);
        d_input_power = 0.0;
        d_mag = 0.0;
        d_num_noncoherent_integrations_counter++;
        lk.unlock();
        gc_acq_result r;
        std::memset(&r, 0, sizeof r);
        d_status = d_acq ? gc_acq_dwell(d_acq, reinterpret_cast<const float*>(d_data_buffer.data()), &r) : GC_ERR_STATE;
        d_test_statistics = r.test_statistics;
        d_input_power = r.input_power;
        d_last = r;
        const uint32_t effective_fft_size = (acq_parameters.bit_transition_flag ? (d_fft_size / 2) : d_fft_size);
        if (d_dump and d_channel == d_dump_channel and d_status == GC_OK)
            {
                // "Record results to file if required" (:741-744, :799-802): the reference copies each
                // accumulated grid row after it is updated; here the rows come back from HBM in one piece
                const uint32_t bins = d_step_two ? d_num_doppler_bins_step2 : d_num_doppler_bins;
                std::vector<float>& dst = d_step_two ? narrow_grid_ : grid_;
                d_grid_tmp.resize(static_cast<size_t>(bins) * d_fft_size);
                if (gc_acq_get_grid(d_acq, 0, d_grid_tmp.data()) == GC_OK)
                    {
                        for (uint32_t b = 0; b < bins; b++)
                            std::memcpy(&dst[static_cast<size_t>(b) * effective_fft_size], &d_grid_tmp[static_cast<size_t>(b) * d_fft_size], sizeof(float) * effective_fft_size);
                    }
            }
        if (acq_parameters.use_automatic_resampler)
            {
                d_gnss_synchro->Acq_delay_samples = r.acq_delay_samples * acq_parameters.resampler_ratio;
                d_gnss_synchro->Acq_delay_samples -= static_cast<double>(acq_parameters.resampler_latency_samples);
                d_gnss_synchro->Acq_doppler_hz = r.acq_doppler_hz;
                d_gnss_synchro->Acq_samplestamp_samples = rint(static_cast<double>(samp_count) * acq_parameters.resampler_ratio);
            }
        else
            {
                d_gnss_synchro->Acq_delay_samples = r.acq_delay_samples;
                d_gnss_synchro->Acq_doppler_hz = r.acq_doppler_hz;
                d_gnss_synchro->Acq_samplestamp_samples = samp_count;
            }
        if (d_step_two) d_gnss_synchro->Acq_doppler_step = acq_parameters.doppler_step2;

        lk.lock();
        // decision (:833-927)
        if (!acq_parameters.bit_transition_flag)
            {
                if (d_test_statistics > d_threshold)
                    {
                        d_active = false;
                        if (acq_parameters.make_2_steps)
                            {
                                if (d_step_two)
                                    {
                                        send_positive_acquisition();
                                        d_step_two = false;
                                        d_state = 0;
                                    }
                                else
                                    {
                                        d_step_two = true;  // small grid around the coarse Doppler next
                                        d_num_noncoherent_integrations_counter = 0;
                                        d_positive_acq = 0;
                                        d_state = 0;
                                    }
                            }
                        else
                            {
                                send_positive_acquisition();
                                d_state = 0;
                            }
                    }
                else
                    {
                        d_buffer_count = 0;
                        d_state = 1;
                    }
                if (d_num_noncoherent_integrations_counter == acq_parameters.max_dwells)
                    {
                        if (d_state != 0) send_negative_acquisition();
                        d_state = 0;
                        d_active = false;
                        d_step_two = false;
                    }
            }
        else
            {
                d_active = false;
                if (d_test_statistics > d_threshold)
                    {
                        if (acq_parameters.make_2_steps)
                            {
                                if (d_step_two)
                                    {
                                        send_positive_acquisition();
                                        d_step_two = false;
                                        d_state = 0;
                                    }
                                else
                                    {
                                        d_step_two = true;
                                        d_num_noncoherent_integrations_counter = 0U;
                                        d_state = 0;
                                    }
                            }
                        else
                            {
                                send_positive_acquisition();
                                d_state = 0;
                            }
                    }
                else
                    {
                        d_state = 0;
                        d_step_two = false;
                        send_negative_acquisition();
                    }
            }
        d_worker_active = false;
        if ((d_num_noncoherent_integrations_counter == acq_parameters.max_dwells) or (d_positive_acq == 1))
            {
                // Record results to file if required (:913-918)
                if (d_dump and d_channel == d_dump_channel) dump_results(effective_fft_size);
                d_num_noncoherent_integrations_counter = 0U;
                d_positive_acq = 0;
                if (d_acq != nullptr)
                    {
                        gc_acq_reset(d_acq);  // "Reset grid" (:917-924)
                        if (!d_step_two) gc_acq_set_step_two(d_acq, 0, 0.0f);
                    }
            }
    }

    //! messages published on the "events" port: 1 = ACQ_SUCCESS, 2 = ACQ_FAIL (:418-460)
    const std::vector<int>& events() const { return d_events; }
    void clear_events() { d_events.clear(); }
    float test_statistics() const { return d_test_statistics; }
    float input_power() const { return d_input_power; }
    const gc_acq_result& last_result() const { return d_last; }
    uint32_t fft_size() const { return d_fft_size; }
    uint32_t consumed_samples() const { return d_consumed_samples; }
    uint32_t num_doppler_bins() const { return d_num_doppler_bins; }
    int32_t state() const { return d_state; }
    bool active() const { return d_active; }
    uint64_t sample_counter() const { return d_sample_counter; }
    gc_status last_status() const { return d_status; }

    //! per Doppler bin (maximum of the grid row, its index) of the last search as the engine's statistics kernel saw them: what a
    //! failing test prints so that a red log names the bins (gc_acq_peek, GC_ACQ_PEEK_ROW_MAX)
    std::vector<std::pair<float, uint32_t>> row_maxima() const
    {
        std::vector<std::pair<float, uint32_t>> out;
        if (d_acq == nullptr) return out;
        uint32_t fft = 0, cons = 0, bins = 0;
        if (gc_acq_fft_size(d_acq, &fft, &cons, &bins) != GC_OK) return out;
        std::vector<float> v(2 * static_cast<size_t>(bins));
        if (gc_acq_peek(d_acq, GC_ACQ_PEEK_ROW_MAX, 0, v.data()) != GC_OK) return out;
        for (uint32_t d = 0; d < bins; d++) out.emplace_back(v[2 * d], static_cast<uint32_t>(v[2 * d + 1]));
        return out;
    }

    //! name of the last file dump_results() wrote ("" if none)
    const std::string& last_dump_file() const { return d_last_dump_file; }

private:
    static bool create_directory(const std::string& path)
    {
        // gnss_sdr_create_directory (src/algorithms/libs/gnss_sdr_create_directory.cc): mkdir -p
        if (path.empty() || path == "." || path == "/") return true;
        struct stat st;
        if (::stat(path.c_str(), &st) == 0) return S_ISDIR(st.st_mode);
        const size_t slash = path.find_last_of('/');
        if (slash != std::string::npos && slash > 0 && !create_directory(path.substr(0, slash))) return false;
        return ::mkdir(path.c_str(), 0775) == 0 || (::stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode));
    }

    /*! pcps_acquisition::dump_results (:462-562): same file name, variable names, classes and
     *  dimensions; Level-5 container instead of matio's v7.3 (see mat5_writer.h). */
    void dump_results(int32_t effective_fft_size)
    {
        d_dump_number++;
        std::string filename = d_dump_filename;
        filename.append("_");
        filename.append(1, d_gnss_synchro->System);
        filename.append("_");
        filename.append(1, d_gnss_synchro->Signal[0]);
        filename.append(1, d_gnss_synchro->Signal[1]);
        filename.append("_ch_");
        filename.append(std::to_string(d_channel));
        filename.append("_");
        filename.append(std::to_string(d_dump_number));
        filename.append("_sat_");
        filename.append(std::to_string(d_gnss_synchro->PRN));
        filename.append(".mat");

        gnsscorr::Mat5Writer mat;
        if (!mat.open(filename))
            {
                std::cout << "Unable to create or open Acquisition dump file" << std::endl;
                return;
            }
        bool ok = mat.write_single_matrix("acq_grid", static_cast<size_t>(effective_fft_size), d_num_doppler_bins, grid_.data());
        ok = ok && mat.write_scalar("doppler_max", static_cast<uint32_t>(acq_parameters.doppler_max));
        ok = ok && mat.write_scalar("doppler_step", static_cast<uint32_t>(d_doppler_step));
        ok = ok && mat.write_scalar("d_positive_acq", static_cast<int32_t>(d_positive_acq));
        ok = ok && mat.write_scalar("acq_doppler_hz", static_cast<float>(d_gnss_synchro->Acq_doppler_hz));
        ok = ok && mat.write_scalar("acq_delay_samples", static_cast<float>(d_gnss_synchro->Acq_delay_samples));
        ok = ok && mat.write_scalar("test_statistic", d_test_statistics);
        ok = ok && mat.write_scalar("threshold", d_threshold);
        ok = ok && mat.write_scalar("input_power", d_input_power);
        ok = ok && mat.write_scalar("sample_counter", static_cast<uint64_t>(d_sample_counter));
        ok = ok && mat.write_scalar("PRN", static_cast<uint32_t>(d_gnss_synchro->PRN));
        ok = ok && mat.write_scalar("num_dwells", static_cast<uint32_t>(d_num_noncoherent_integrations_counter));
        if (acq_parameters.make_2_steps)
            {
                ok = ok && mat.write_single_matrix("acq_grid_narrow", static_cast<size_t>(effective_fft_size), d_num_doppler_bins_step2, narrow_grid_.data());
                ok = ok && mat.write_scalar("doppler_step_narrow", static_cast<float>(acq_parameters.doppler_step2));
                const float aux = d_doppler_center_step_two - static_cast<float>(std::floor(d_num_doppler_bins_step2 / 2.0)) * acq_parameters.doppler_step2;
                ok = ok && mat.write_scalar("doppler_grid_narrow_min", aux);
            }
        mat.close();
        if (ok) d_last_dump_file = filename;
    }

    void send_positive_acquisition()
    {
        d_positive_acq = 1;
        if (d_channel_fsm)
            d_channel_fsm->Event_valid_acquisition();  // direct notification (:432-436)
        else
            d_events.push_back(1);
    }

    void send_negative_acquisition()
    {
        d_positive_acq = 0;
        d_events.push_back(2);
    }

    Acq_Conf acq_parameters;
    int64_t d_old_freq = 0;
    std::map<uint32_t, int32_t> d_glonass_prn;
    bool d_dump = false;
    uint32_t d_dump_channel = 0U;
    int64_t d_dump_number = 0LL;
    std::string d_dump_filename;
    std::string d_last_dump_file;
    std::vector<float> grid_, narrow_grid_, d_grid_tmp;
    gc_acq* d_acq = nullptr;
    gc_status d_status = GC_OK;
    gc_acq_result d_last{};
    std::mutex d_setlock;
    Gnss_Synchro* d_gnss_synchro = nullptr;
    std::shared_ptr<ChannelFsm> d_channel_fsm;
    std::vector<gr_complex> d_data_buffer;
    std::vector<gr_complex> d_code;
    std::vector<int> d_events;
    bool d_active = false;
    bool d_worker_active = false;
    bool d_step_two = false;
    bool d_use_CFAR_algorithm_flag = false;
    int32_t d_positive_acq = 0;
    int32_t d_state = 0;
    uint32_t d_channel = 0U;
    uint32_t d_doppler_step = 0U;
    float d_doppler_center_step_two = 0.0f;
    float d_threshold = 0.0f;
    float d_mag = 0.0f;
    float d_input_power = 0.0f;
    float d_test_statistics = 0.0f;
    uint32_t d_num_noncoherent_integrations_counter = 0U;
    uint32_t d_fft_size = 0U;
    uint32_t d_consumed_samples = 0U;
    uint32_t d_num_doppler_bins = 0U;
    uint32_t d_num_doppler_bins_step2 = 0U;
    uint32_t d_buffer_count = 0U;
    uint64_t d_sample_counter = 0ULL;
};

#endif  // GNSSCORR_HIP_PCPS_ACQUISITION_H_
