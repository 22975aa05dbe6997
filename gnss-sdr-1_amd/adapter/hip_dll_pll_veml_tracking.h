/*!
 * \file hip_dll_pll_veml_tracking.h
 * \brief Image of the dll_pll_veml_tracking block
 * (src/algorithms/tracking/gnuradio_blocks/dll_pll_veml_tracking.{h,cc}) for GPS L1 C/A "1C", L2C(M) "2S", L5 "L5",
 * Galileo E1 "1B", E5a "5X", BeiDou B1I "B1" and B3I "B3", with the correlations done by
 * Hip_Multicorrelator_Real_Codes (libgnsscorr.so, MI355X).
 *
 * Mirrored: constructor set-up of taps/shifts and of the per-signal synchronisation data (dll_pll_veml_tracking.cc:113-440),
 * start_tracking (:549-747), the whole state machine of general_work (:1544-1907): standby (0), pull-in (1), wide
 * tracking with secondary-code lock (acquire_secondary, :800-836) or telemetry-preamble bit synchronisation (2),
 * extended coherent integration (3) and narrow tracking (4) with save_correlation_results (:1072-1125);
 * do_correlation_step (:886-911) incl. the data-component prompt correlator of pilot tracking (Galileo E1-C drives
 * the loop, E1-B is demodulated), run_dll_pll (:914-973), update_tracking_vars (:998-1070) incl. the
 * high-dynamics rate smoother, cn0_and_tracking_lock_status (:839-878), the Gnss_Synchro record written
 * per epoch (:1693-1725, :1898-1906).
 * Not mirrored (outside the correlator hot path): the telemetry fault message handler.  The binary dump (log_data,
 * :1128-1250) is written in the reference's record layout and converted to a .mat file with the reference's variable
 * names when the block is destroyed (save_matfile, :1253-1438; Level-5 instead of v7.3, see mat5_writer.h).
 *
 * general_work(noutput, ninput_items, input_items, output_items) becomes
 * work(in, ninput_items, out): returns the number of input items consumed (consume_each) and sets
 * *produced to 1 when *out was written; "events" messages (3 = loss of lock) are queued in events().
 */
#ifndef GNSSCORR_HIP_DLL_PLL_VEML_TRACKING_H_
#define GNSSCORR_HIP_DLL_PLL_VEML_TRACKING_H_

#include "gnss_sdr_types.h"
#include "hip_multicorrelator_real_codes.h"
#include "mat5_writer.h"
#include "tracking_loop_maths.h"
#include <cmath>
#include <cstring>
#include <deque>
#include <fstream>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

class hip_dll_pll_veml_tracking
{
public:
    explicit hip_dll_pll_veml_tracking(const Dll_Pll_Conf& conf_) : trk_parameters(conf_)
    {
        signal_type = std::string(trk_parameters.signal);
        d_symbols_per_bit = 1;
        if (trk_parameters.system == 'G' && signal_type == "1C")
            {
                d_signal_carrier_freq = 1575.42e6;  // GPS_L1_FREQ_HZ
                d_code_period = 0.001;
                d_code_chip_rate = 1.023e6;
                d_symbols_per_bit = 20;  // GPS_CA_TELEMETRY_SYMBOLS_PER_BIT
                d_correlation_length_ms = 1;
                d_code_samples_per_chip = 1;
                d_code_length_chips = 1023;
                trk_parameters.track_pilot = false;  // no pilot component, no secondary code
                set_preamble({1, 0, 0, 0, 1, 0, 1, 1}, 20);  // GPS_PREAMBLE, one entry per 1 ms symbol (:129-150)
            }
        else if (trk_parameters.system == 'E' && signal_type == "1B")
            {
                d_signal_carrier_freq = 1575.42e6;  // Galileo_E1_FREQ_HZ
                d_code_period = 0.004;
                d_code_chip_rate = 1.023e6;
                d_code_length_chips = 4092;
                d_correlation_length_ms = 4;
                d_code_samples_per_chip = 2;  // sinBOC(1,1) replica, 2 samples per chip
                d_veml = true;
                d_symbols_per_bit = 1;
                if (trk_parameters.track_pilot)
                    {
                        d_secondary = true;
                        d_secondary_code_string = "0011100000001010110110010";  // GALILEO_E1_C_SECONDARY_CODE (OS SIS ICD, CS25_1)
                    }
            }
        else if (trk_parameters.system == 'C' && signal_type == "B1")
            {
                d_signal_carrier_freq = 1.561098e9;  // BEIDOU_B1I_FREQ_HZ
                d_code_period = 0.001;
                d_code_chip_rate = 2.046e6;
                d_code_length_chips = 2046;
                d_correlation_length_ms = 1;
                d_code_samples_per_chip = 1;
                d_symbols_per_bit = 20;  // BEIDOU_B1I_TELEMETRY_SYMBOLS_PER_BIT
                d_secondary = true;
                d_secondary_code_string = "00000100110101001110";  // BEIDOU_B1I_SECONDARY_CODE_STR (NH20)
                trk_parameters.track_pilot = false;
            }
        else if (trk_parameters.system == 'G' && signal_type == "2S")
            {
                d_signal_carrier_freq = 1.22760e9;  // GPS_L2_FREQ_HZ
                d_code_period = 0.02;               // GPS_L2_M_PERIOD
                d_code_chip_rate = 0.5115e6;
                d_code_length_chips = 10230;
                d_symbols_per_bit = 1;  // GPS_L2_SAMPLES_PER_SYMBOL
                d_correlation_length_ms = 20;
                d_code_samples_per_chip = 1;
                trk_parameters.track_pilot = false;  // no pilot component, no secondary code
            }
        else if (trk_parameters.system == 'G' && signal_type == "L5")
            {
                d_signal_carrier_freq = 1.17645e9;  // GPS_L5_FREQ_HZ
                d_code_period = 0.001;
                d_code_chip_rate = 10.23e6;
                d_symbols_per_bit = 10;  // GPS_L5_SAMPLES_PER_SYMBOL
                d_correlation_length_ms = 1;
                d_code_samples_per_chip = 1;
                d_code_length_chips = 10230;
                d_secondary = true;
                // pilot: Q5 with NH20, in quadrature with the data component; data: I5 with NH10
                d_secondary_code_string = trk_parameters.track_pilot ? "00000100110101001110" : "0000110101";
                interchange_iq = trk_parameters.track_pilot;
            }
        else if (trk_parameters.system == 'E' && signal_type == "5X")
            {
                d_signal_carrier_freq = 1.17645e9;  // GALILEO_E5A_FREQ_HZ
                d_code_period = 0.001;
                d_code_chip_rate = 1.023e7;
                d_symbols_per_bit = 20;
                d_correlation_length_ms = 1;
                d_code_samples_per_chip = 1;
                d_code_length_chips = 10230;
                // pilot: E5a-Q with the 100-symbol secondary code of the PRN (set in start_tracking); on the data component the
                // secondary code is left to the telemetry decoder
                d_secondary = trk_parameters.track_pilot;
                interchange_iq = trk_parameters.track_pilot;
            }
        else if (trk_parameters.system == 'C' && signal_type == "B3")
            {
                d_signal_carrier_freq = 1.268520e9;  // BEIDOU_B3I_FREQ_HZ
                d_code_period = 0.001;
                d_code_chip_rate = 10.23e6;
                d_code_length_chips = 10230;
                d_symbols_per_bit = 20;
                d_correlation_length_ms = 1;
                d_code_samples_per_chip = 1;
                d_secondary = true;
                d_secondary_code_string = "00000100110101001110";  // BEIDOU_B3I_SECONDARY_CODE_STR (NH20)
                trk_parameters.track_pilot = false;
            }
        if (trk_parameters.extend_correlation_symbols > 1)
            d_enable_extended_integration = true;
        else
            {
                d_enable_extended_integration = false;
                trk_parameters.extend_correlation_symbols = 1;
            }
        if (trk_parameters.track_pilot)
            {
                // extra prompt correlator for the data component, slaved to the pilot prompt (:404-416)
                correlator_data_cpu.init(2 * trk_parameters.vector_length, 1);
                correlator_data_cpu.set_high_dynamics_resampler(trk_parameters.high_dyn);
                d_data_code.assign(2 * d_code_length_chips, 0.0f);
            }
        d_code_loop_filter = Tracking_loop_filter(d_code_period, trk_parameters.dll_bw_hz, trk_parameters.dll_filter_order, false);
        d_carrier_loop_filter.set_params(trk_parameters.fll_bw_hz, trk_parameters.pll_bw_hz, trk_parameters.pll_filter_order);
        d_tracking_code.assign(2 * d_code_length_chips, 0.0f);
        d_n_correlator_taps = d_veml ? 5 : 3;
        d_correlator_outs.assign(d_n_correlator_taps, gr_complex(0.0, 0.0));
        d_local_code_shift_chips.assign(d_n_correlator_taps, 0.0f);
        set_tap_shifts();
        multicorrelator_cpu.init(2 * trk_parameters.vector_length, d_n_correlator_taps);
        multicorrelator_cpu.set_high_dynamics_resampler(trk_parameters.high_dyn);
        d_code_freq_chips = d_code_chip_rate;
        d_current_prn_length_samples = static_cast<int32_t>(trk_parameters.vector_length);
        d_Prompt_buffer.assign(trk_parameters.cn0_samples, gr_complex(0.0, 0.0));
        d_carrier_lock_threshold = trk_parameters.carrier_lock_th;
        d_carr_ph_history_cap = 2 * trk_parameters.smoother_length;
    }

    //! (:750-797): closes the dump and, with dump_mat, converts it
    ~hip_dll_pll_veml_tracking()
    {
        if (d_dump_file.is_open())
            {
                d_dump_file.close();
                if (trk_parameters.dump_mat) save_matfile();
            }
    }

    /*! save_matfile (:1253-1438): <dump_filename><channel>.dat -> .mat, one 1 x num_epoch array per field of the record */
    int32_t save_matfile() const
    {
        std::string name = trk_parameters.dump_filename;
        name.append(std::to_string(d_channel));
        name.append(".dat");
        std::ifstream in(name.c_str(), std::ios::binary | std::ios::ate);
        if (!in.is_open()) return 1;
        const size_t record = 96;  // 19 floats + uint64 + double + uint32, packed
        const size_t num_epoch = static_cast<size_t>(in.tellg()) / record;
        if (num_epoch == 0) return 1;
        std::vector<char> raw(num_epoch * record);
        in.seekg(0, std::ios::beg);
        in.read(raw.data(), static_cast<std::streamsize>(raw.size()));
        if (!in) return 1;
        name.erase(name.length() - 4, 4);
        name.append(".mat");
        gnsscorr::Mat5Writer mat;
        if (!mat.open(name)) return 1;
        // field table of the record, in file order
        struct Field
        {
            const char* name;
            size_t offset, size;
            uint32_t mx, mi;
        };
        using W = gnsscorr::Mat5Writer;
        static const Field fields[] = {
            {"abs_VE", 0, 4, W::mxSINGLE, W::miSINGLE}, {"abs_E", 4, 4, W::mxSINGLE, W::miSINGLE}, {"abs_P", 8, 4, W::mxSINGLE, W::miSINGLE},
            {"abs_L", 12, 4, W::mxSINGLE, W::miSINGLE}, {"abs_VL", 16, 4, W::mxSINGLE, W::miSINGLE}, {"Prompt_I", 20, 4, W::mxSINGLE, W::miSINGLE},
            {"Prompt_Q", 24, 4, W::mxSINGLE, W::miSINGLE}, {"PRN_start_sample_count", 28, 8, W::mxUINT64, W::miUINT64},
            {"acc_carrier_phase_rad", 36, 4, W::mxSINGLE, W::miSINGLE}, {"carrier_doppler_hz", 40, 4, W::mxSINGLE, W::miSINGLE},
            {"carrier_doppler_rate_hz", 44, 4, W::mxSINGLE, W::miSINGLE}, {"code_freq_chips", 48, 4, W::mxSINGLE, W::miSINGLE},
            {"code_freq_rate_chips", 52, 4, W::mxSINGLE, W::miSINGLE}, {"carr_error_hz", 56, 4, W::mxSINGLE, W::miSINGLE},
            {"carr_error_filt_hz", 60, 4, W::mxSINGLE, W::miSINGLE}, {"code_error_chips", 64, 4, W::mxSINGLE, W::miSINGLE},
            {"code_error_filt_chips", 68, 4, W::mxSINGLE, W::miSINGLE}, {"CN0_SNV_dB_Hz", 72, 4, W::mxSINGLE, W::miSINGLE},
            {"carrier_lock_test", 76, 4, W::mxSINGLE, W::miSINGLE}, {"aux1", 80, 4, W::mxSINGLE, W::miSINGLE}, {"aux2", 84, 8, W::mxDOUBLE, W::miDOUBLE},
            {"PRN", 92, 4, W::mxUINT32, W::miUINT32}};
        std::vector<char> column(num_epoch * 8);
        bool ok = true;
        for (const Field& f : fields)
            {
                for (size_t e = 0; e < num_epoch; e++) std::memcpy(&column[e * f.size], &raw[e * record + f.offset], f.size);
                ok = ok && mat.write_array(f.name, f.mx, f.mi, f.size, 1, num_epoch, column.data());
            }
        mat.close();
        return ok ? 0 : 1;
    }

    //! set_channel (:1442-1480): with Tracking_XX.dump the binary dump file "<dump_filename><channel>.dat" is opened here
    void set_channel(uint32_t channel)
    {
        d_channel = channel;
        if (trk_parameters.dump && !d_dump_file.is_open())
            {
                std::string name = trk_parameters.dump_filename;
                name.append(std::to_string(d_channel));
                name.append(".dat");
                d_dump_file.open(name.c_str(), std::ios::out | std::ios::binary);
            }
    }
    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) { d_acquisition_gnss_synchro = p_gnss_synchro; }

    //! dll_pll_veml_tracking::start_tracking (:549-747)
    void start_tracking()
    {
        std::lock_guard<std::mutex> l(d_setlock);
        d_acq_code_phase_samples = d_acquisition_gnss_synchro->Acq_delay_samples;
        d_acq_carrier_doppler_hz = d_acquisition_gnss_synchro->Acq_doppler_hz;
        d_acq_sample_stamp = d_acquisition_gnss_synchro->Acq_samplestamp_samples;
        d_carrier_doppler_hz = d_acq_carrier_doppler_hz;
        d_carrier_phase_step_rad = PI_2 * d_carrier_doppler_hz / trk_parameters.fs_in;
        d_carrier_phase_rate_step_rad = 0.0;
        d_carr_ph_history.clear();
        d_code_ph_history.clear();
        d_carrier_loop_filter.initialize(static_cast<float>(d_acq_carrier_doppler_hz));
        d_code_loop_filter.initialize();
        const uint32_t prn = d_acquisition_gnss_synchro->PRN;
        if (trk_parameters.system == 'G' && signal_type == "2S")
            gc_gps_l2c_m_code_gen_float(d_tracking_code.data(), prn);
        else if (trk_parameters.system == 'G' && signal_type == "L5")
            {
                if (trk_parameters.track_pilot)
                    {
                        gc_gps_l5q_code_gen_float(d_tracking_code.data(), prn);
                        gc_gps_l5i_code_gen_float(d_data_code.data(), prn);
                        d_Prompt_Data = gr_complex(0.0, 0.0);
                        correlator_data_cpu.set_local_code_and_taps(d_code_length_chips, d_data_code.data(), &d_local_code_shift_chips[1]);
                    }
                else
                    gc_gps_l5i_code_gen_float(d_tracking_code.data(), prn);
            }
        else if (trk_parameters.system == 'E' && signal_type == "5X")
            {
                // the complex primary code holds E5a-I in the real and E5a-Q in the imaginary part (:606-626)
                std::vector<float> aux(2 * d_code_length_chips);
                gc_galileo_e5_a_code_gen_complex_primary(aux.data(), static_cast<int32_t>(prn), "5X");
                if (trk_parameters.track_pilot)
                    {
                        char sec[128];
                        int32_t len = 0;
                        if (gc_secondary_code("5Q", prn, sec, sizeof sec, &len) == GC_OK) d_secondary_code_string.assign(sec, len);
                        for (uint32_t i = 0; i < d_code_length_chips; i++)
                            {
                                d_tracking_code[i] = aux[2 * i + 1];
                                d_data_code[i] = aux[2 * i];
                            }
                        d_Prompt_Data = gr_complex(0.0, 0.0);
                        correlator_data_cpu.set_local_code_and_taps(d_code_length_chips, d_data_code.data(), &d_local_code_shift_chips[1]);
                    }
                else
                    for (uint32_t i = 0; i < d_code_length_chips; i++) d_tracking_code[i] = aux[2 * i];
            }
        else if (trk_parameters.system == 'C' && signal_type == "B3")
            {
                gc_beidou_b3i_code_gen_float(d_tracking_code.data(), static_cast<int32_t>(prn), 0);
                if (prn > 0 and prn < 6)
                    {
                        // GEO satellites broadcast D2 (:672-705)
                        d_symbols_per_bit = 2;
                        d_secondary = false;
                        d_secondary_code_string.clear();
                        set_preamble({1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 0}, 2);
                    }
            }
        else if (trk_parameters.system == 'G')
            gc_gps_l1_ca_code_gen_float(d_tracking_code.data(), static_cast<int32_t>(d_acquisition_gnss_synchro->PRN), 0);
        else if (trk_parameters.system == 'E')
            {
                char sig[3] = "1B";
                if (trk_parameters.track_pilot)
                    {
                        char pilot_signal[3] = "1C";
                        gc_galileo_e1_code_gen_sinboc11_float(d_tracking_code.data(), pilot_signal, d_acquisition_gnss_synchro->PRN);
                        gc_galileo_e1_code_gen_sinboc11_float(d_data_code.data(), sig, d_acquisition_gnss_synchro->PRN);
                        d_Prompt_Data = gr_complex(0.0, 0.0);
                        // the data correlator's only tap IS the prompt shift of the main correlator (pointer, not copy)
                        correlator_data_cpu.set_local_code_and_taps(d_code_samples_per_chip * d_code_length_chips, d_data_code.data(),
                            &d_local_code_shift_chips[2]);
                    }
                else
                    gc_galileo_e1_code_gen_sinboc11_float(d_tracking_code.data(), sig, d_acquisition_gnss_synchro->PRN);
            }
        else
            {
                gc_beidou_b1i_code_gen_float(d_tracking_code.data(), static_cast<int32_t>(d_acquisition_gnss_synchro->PRN), 0);
                if (d_acquisition_gnss_synchro->PRN > 0 and d_acquisition_gnss_synchro->PRN < 6)
                    {
                        // GEO satellites broadcast D2: 2 symbols per bit, no NH code, 11-bit preamble (:631-665)
                        d_symbols_per_bit = 2;
                        d_secondary = false;
                        d_secondary_code_string.clear();
                        set_preamble({1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 0}, 2);
                    }
            }
        multicorrelator_cpu.set_local_code_and_taps(d_code_samples_per_chip * d_code_length_chips, d_tracking_code.data(), d_local_code_shift_chips.data());
        std::fill(d_correlator_outs.begin(), d_correlator_outs.end(), gr_complex(0.0, 0.0));
        d_carrier_lock_fail_counter = 0;
        d_rem_code_phase_samples = 0.0;
        d_rem_carr_phase_rad = 0.0;
        d_rem_code_phase_chips = 0.0;
        d_acc_carrier_phase_rad = 0.0;
        d_cn0_estimation_counter = 0;
        d_carrier_lock_test = 1.0;
        d_CN0_SNV_dB_Hz = 0.0;
        set_tap_shifts();
        d_current_correlation_time_s = d_code_period;
        d_code_loop_filter.set_noise_bandwidth(trk_parameters.dll_bw_hz);
        d_code_loop_filter.set_update_interval(d_code_period);
        d_state = 1;  // pull-in
        d_cloop = true;
        d_pull_in_transitory = true;
        d_Prompt_circular_buffer.clear();
    }

    void stop_tracking()
    {
        std::lock_guard<std::mutex> l(d_setlock);
        d_state = 0;
    }

    //! forecast (:507-514): the scheduler calls work() with at least this many items
    int required_input_items() const { return static_cast<int>(trk_parameters.vector_length) * 2; }

    /*! general_work (:1544-1907) */
    int work(const gr_complex* in, int ninput_items, Gnss_Synchro* out, int* produced)
    {
        std::lock_guard<std::mutex> l(d_setlock);
        *produced = 0;
        Gnss_Synchro current_synchro_data = Gnss_Synchro();
        if (d_pull_in_transitory == true)
            {
                if (trk_parameters.pull_in_time_s < (d_sample_counter - d_acq_sample_stamp) / static_cast<int>(trk_parameters.fs_in)) d_pull_in_transitory = false;
            }
        switch (d_state)
            {
            case 0:  // standby
                d_sample_counter += static_cast<uint64_t>(ninput_items);
                return ninput_items;
            case 1:  // pull-in: skip samples until the incoming code is aligned with the local replica
                {
                    int64_t acq_trk_diff_samples = static_cast<int64_t>(d_sample_counter) - static_cast<int64_t>(d_acq_sample_stamp);
                    double delta_trk_to_acq_prn_start_samples = static_cast<double>(acq_trk_diff_samples) - d_acq_code_phase_samples;
                    d_code_freq_chips = d_code_chip_rate;
                    d_code_phase_step_chips = d_code_freq_chips / trk_parameters.fs_in;
                    d_code_phase_rate_step_chips = 0.0;
                    double T_chip_mod_seconds = 1.0 / d_code_freq_chips;
                    double T_prn_mod_seconds = T_chip_mod_seconds * static_cast<double>(d_code_length_chips);
                    double T_prn_mod_samples = T_prn_mod_seconds * trk_parameters.fs_in;
                    d_acq_code_phase_samples = T_prn_mod_samples - std::fmod(delta_trk_to_acq_prn_start_samples, T_prn_mod_samples);
                    d_current_prn_length_samples = std::round(T_prn_mod_samples);
                    int32_t samples_offset = std::round(d_acq_code_phase_samples);
                    d_acc_carrier_phase_rad -= d_carrier_phase_step_rad * static_cast<double>(samples_offset);
                    d_state = 2;
                    d_sample_counter += samples_offset;
                    return samples_offset;
                }
            case 2:  // wide tracking and symbol synchronisation (:1601-1773)
                {
                    do_correlation_step(in);
                    if (d_veml)
                        {
                            d_VE_accu = d_correlator_outs[0];
                            d_VL_accu = d_correlator_outs[4];
                        }
                    d_E_accu = d_correlator_outs[d_veml ? 1 : 0];
                    d_P_accu = d_correlator_outs[d_veml ? 2 : 1];
                    d_L_accu = d_correlator_outs[d_veml ? 3 : 2];
                    if (!cn0_and_tracking_lock_status(d_code_period))
                        {
                            clear_tracking_vars();
                            d_state = 0;  // loss of lock
                            break;
                        }
                    run_dll_pll();
                    update_tracking_vars();
                    log_data(false);
                    const gr_complex prompt = d_correlator_outs[d_veml ? 2 : 1];
                    bool next_state = false;
                    if (d_secondary)
                        {
                            d_Prompt_circular_buffer.push_back(prompt);
                            if (d_Prompt_circular_buffer.size() > d_secondary_code_string.size()) d_Prompt_circular_buffer.pop_front();
                            if (d_Prompt_circular_buffer.size() == d_secondary_code_string.size()) next_state = acquire_secondary();
                        }
                    else if (d_symbols_per_bit > 1)
                        {
                            // no secondary code: look for the telemetry preamble on the symbol signs, after 10 s of tracking
                            float current_tracking_time_s = static_cast<float>(d_sample_counter - d_acq_sample_stamp) / trk_parameters.fs_in;
                            if (current_tracking_time_s > d_bit_sync_min_time_s)
                                {
                                    d_symbol_history.push_back(prompt.real());
                                    if (d_symbol_history.size() > d_preambles_symbols.size()) d_symbol_history.pop_front();
                                    int32_t corr_value = 0;
                                    if (d_symbol_history.size() == d_preambles_symbols.size())
                                        {
                                            for (size_t i = 0; i < d_symbol_history.size(); i++)
                                                corr_value += d_symbol_history[i] < 0.0 ? -d_preambles_symbols[i] : d_preambles_symbols[i];
                                        }
                                    next_state = !d_preambles_symbols.empty() && corr_value == static_cast<int32_t>(d_preambles_symbols.size());
                                }
                        }
                    else
                        next_state = true;
                    fill_synchro(current_synchro_data);
                    if (next_state)
                        {
                            reset_accumulators();
                            d_Prompt_circular_buffer.clear();
                            d_current_symbol = 0;
                            if (d_enable_extended_integration)
                                {
                                    d_extend_correlation_symbols_count = 0;
                                    d_current_correlation_time_s = static_cast<float>(trk_parameters.extend_correlation_symbols) * static_cast<float>(d_code_period);
                                    d_state = 3;
                                    // narrow loop bandwidths and tap spacings (:1751-1766)
                                    d_code_loop_filter.set_update_interval(d_current_correlation_time_s);
                                    d_code_loop_filter.set_noise_bandwidth(trk_parameters.dll_bw_narrow_hz);
                                    d_carrier_loop_filter.set_params(trk_parameters.fll_bw_hz, trk_parameters.pll_bw_narrow_hz, trk_parameters.pll_filter_order);
                                    const float spc = static_cast<float>(d_code_samples_per_chip);
                                    if (d_veml)
                                        {
                                            d_local_code_shift_chips[0] = -trk_parameters.very_early_late_space_narrow_chips * spc;
                                            d_local_code_shift_chips[1] = -trk_parameters.early_late_space_narrow_chips * spc;
                                            d_local_code_shift_chips[3] = trk_parameters.early_late_space_narrow_chips * spc;
                                            d_local_code_shift_chips[4] = trk_parameters.very_early_late_space_narrow_chips * spc;
                                        }
                                    else
                                        {
                                            d_local_code_shift_chips[0] = -trk_parameters.early_late_space_narrow_chips * spc;
                                            d_local_code_shift_chips[2] = trk_parameters.early_late_space_narrow_chips * spc;
                                        }
                                }
                            else
                                d_state = 4;
                        }
                    break;
                }
            case 3:  // coherent integration (:1774-1826): accumulate, no loop update
                {
                    do_correlation_step(in);
                    update_tracking_vars();
                    save_correlation_results();
                    fill_synchro(current_synchro_data);
                    d_extend_correlation_symbols_count++;
                    if (d_extend_correlation_symbols_count == (trk_parameters.extend_correlation_symbols - 1))
                        {
                            d_extend_correlation_symbols_count = 0;
                            d_state = 4;
                        }
                    log_data(true);
                    break;
                }
            case 4:  // narrow tracking (:1827-1896): last period of the integration, then the loop update
                {
                    do_correlation_step(in);
                    save_correlation_results();
                    if (!cn0_and_tracking_lock_status(d_code_period * static_cast<double>(trk_parameters.extend_correlation_symbols)))
                        {
                            clear_tracking_vars();
                            d_state = 0;  // loss of lock
                            break;
                        }
                    run_dll_pll();
                    update_tracking_vars();
                    fill_synchro(current_synchro_data);
                    log_data(false);
                    reset_accumulators();
                    if (d_enable_extended_integration) d_state = 3;
                    break;
                }
            default:
                break;
            }
        const int consumed = d_current_prn_length_samples;
        d_sample_counter += static_cast<uint64_t>(d_current_prn_length_samples);
        if (current_synchro_data.Flag_valid_symbol_output)
            {
                current_synchro_data.System = d_acquisition_gnss_synchro->System;
                current_synchro_data.Signal[0] = d_acquisition_gnss_synchro->Signal[0];
                current_synchro_data.Signal[1] = d_acquisition_gnss_synchro->Signal[1];
                current_synchro_data.PRN = d_acquisition_gnss_synchro->PRN;
                current_synchro_data.Channel_ID = d_acquisition_gnss_synchro->Channel_ID;
                current_synchro_data.fs = static_cast<int64_t>(trk_parameters.fs_in);
                current_synchro_data.Tracking_sample_counter = d_sample_counter;
                *out = current_synchro_data;
                *produced = 1;
            }
        return consumed;
    }

    // observers for tests / the adapter
    int32_t state() const { return d_state; }
    double carrier_doppler_hz() const { return d_carrier_doppler_hz; }
    double code_freq_chips() const { return d_code_freq_chips; }
    double cn0_db_hz() const { return d_CN0_SNV_dB_Hz; }
    double carrier_lock_test() const { return d_carrier_lock_test; }
    double rem_code_phase_samples() const { return d_rem_code_phase_samples; }
    uint64_t sample_counter() const { return d_sample_counter; }
    const std::vector<gr_complex>& correlator_outs() const { return d_correlator_outs; }
    gr_complex prompt_data() const { return d_Prompt_Data; }
    gr_complex prompt_accu() const { return d_P_accu; }
    //! the reference waits 10 s before looking for the telemetry preamble (:1648); tests shorten it
    void set_bit_sync_min_time_s(float t) { d_bit_sync_min_time_s = t; }
    const std::vector<int>& events() const { return d_events; }
    gc_status last_status() const { return multicorrelator_cpu.last_status(); }

private:
    void set_preamble(std::initializer_list<int> bits, int symbols_per_bit)
    {
        d_preambles_symbols.clear();
        for (int bit : bits)
            for (int j = 0; j < symbols_per_bit; j++) d_preambles_symbols.push_back(bit == 1 ? 1 : -1);
        d_symbol_history.clear();
    }

    void reset_accumulators()
    {
        d_VE_accu = d_E_accu = d_P_accu = d_L_accu = d_VL_accu = gr_complex(0.0, 0.0);
    }

    //! what every valid period hands to the telemetry decoder (:1693-1725, :1789-1817, :1847-1878).  GPS L5 and Galileo E5a
    //! carry data and pilot in quadrature: with the loop on the pilot, I and Q of the data prompt are interchanged; E1-B and
    //! E1-C are in anti-phase and are not.
    void fill_synchro(Gnss_Synchro& d) const
    {
        const gr_complex p = trk_parameters.track_pilot ? d_Prompt_Data : d_correlator_outs[d_veml ? 2 : 1];
        d.Prompt_I = static_cast<double>(interchange_iq ? p.imag() : p.real());
        d.Prompt_Q = static_cast<double>(interchange_iq ? p.real() : p.imag());
        d.Code_phase_samples = d_rem_code_phase_samples;
        d.Carrier_phase_rads = d_acc_carrier_phase_rad;
        d.Carrier_Doppler_hz = d_carrier_doppler_hz;
        d.CN0_dB_hz = d_CN0_SNV_dB_Hz;
        d.correlation_length_ms = d_correlation_length_ms;
        d.Flag_valid_symbol_output = true;
    }

    //! (:800-836) all of the last secondary_code_length prompt signs follow the secondary code, or all oppose it
    bool acquire_secondary() const
    {
        int32_t corr_value = 0;
        for (size_t i = 0; i < d_secondary_code_string.size(); i++)
            {
                const bool negative = d_Prompt_circular_buffer[i].real() < 0.0;
                const bool zero = d_secondary_code_string[i] == '0';
                corr_value += (negative == zero) ? 1 : -1;
            }
        return std::abs(corr_value) == static_cast<int32_t>(d_secondary_code_string.size());
    }

    //! (:1072-1125)
    void save_correlation_results()
    {
        float sign = 1.0f;
        if (d_secondary)
            {
                if (d_secondary_code_string[d_current_symbol] != '0') sign = -1.0f;
                d_current_symbol = (d_current_symbol + 1) % static_cast<uint32_t>(d_secondary_code_string.size());
            }
        else
            {
                d_current_symbol++;
                d_current_symbol %= static_cast<uint32_t>(d_symbols_per_bit);
            }
        if (d_veml)
            {
                d_VE_accu += sign * d_correlator_outs[0];
                d_VL_accu += sign * d_correlator_outs[4];
            }
        d_E_accu += sign * d_correlator_outs[d_veml ? 1 : 0];
        d_P_accu += sign * d_correlator_outs[d_veml ? 2 : 1];
        d_L_accu += sign * d_correlator_outs[d_veml ? 3 : 2];
        d_cloop = !trk_parameters.track_pilot;  // pilot: no symbol transitions left, four-quadrant PLL
    }

    void set_tap_shifts()
    {
        // (:372-390, :720-732)
        const float spc = static_cast<float>(d_code_samples_per_chip);
        if (d_veml)
            {
                d_local_code_shift_chips[0] = -trk_parameters.very_early_late_space_chips * spc;
                d_local_code_shift_chips[1] = -trk_parameters.early_late_space_chips * spc;
                d_local_code_shift_chips[2] = 0.0;
                d_local_code_shift_chips[3] = trk_parameters.early_late_space_chips * spc;
                d_local_code_shift_chips[4] = trk_parameters.very_early_late_space_chips * spc;
            }
        else
            {
                d_local_code_shift_chips[0] = -trk_parameters.early_late_space_chips * spc;
                d_local_code_shift_chips[1] = 0.0;
                d_local_code_shift_chips[2] = trk_parameters.early_late_space_chips * spc;
            }
    }

    //! (:886-911) -- the hot path: one launch of the HIP multicorrelator
    void do_correlation_step(const gr_complex* input_samples)
    {
        multicorrelator_cpu.set_input_output_vectors(d_correlator_outs.data(), input_samples);
        multicorrelator_cpu.Carrier_wipeoff_multicorrelator_resampler(
            d_rem_carr_phase_rad,
            d_carrier_phase_step_rad, d_carrier_phase_rate_step_rad,
            static_cast<float>(d_rem_code_phase_chips) * static_cast<float>(d_code_samples_per_chip),
            static_cast<float>(d_code_phase_step_chips) * static_cast<float>(d_code_samples_per_chip),
            static_cast<float>(d_code_phase_rate_step_chips) * static_cast<float>(d_code_samples_per_chip),
            trk_parameters.vector_length);
        if (trk_parameters.track_pilot)
            {
                correlator_data_cpu.set_input_output_vectors(&d_Prompt_Data, input_samples);
                correlator_data_cpu.Carrier_wipeoff_multicorrelator_resampler(
                    d_rem_carr_phase_rad,
                    d_carrier_phase_step_rad, d_carrier_phase_rate_step_rad,
                    static_cast<float>(d_rem_code_phase_chips) * static_cast<float>(d_code_samples_per_chip),
                    static_cast<float>(d_code_phase_step_chips) * static_cast<float>(d_code_samples_per_chip),
                    static_cast<float>(d_code_phase_rate_step_chips) * static_cast<float>(d_code_samples_per_chip),
                    trk_parameters.vector_length);
            }
    }

    //! (:914-973)
    void run_dll_pll()
    {
        if (d_cloop)
            d_carr_phase_error_hz = pll_cloop_two_quadrant_atan(d_P_accu) / PI_2;
        else
            d_carr_phase_error_hz = pll_four_quadrant_atan(d_P_accu) / PI_2;
        if ((d_pull_in_transitory == true and trk_parameters.enable_fll_pull_in == true) or trk_parameters.enable_fll_steady_state)
            {
                d_carr_freq_error_hz = fll_four_quadrant_atan(d_P_accu_old, d_P_accu, 0, d_current_correlation_time_s) / PI_2;
                d_P_accu_old = d_P_accu;
                if ((d_pull_in_transitory == true and trk_parameters.enable_fll_pull_in == true))
                    d_carr_error_filt_hz = d_carrier_loop_filter.get_carrier_error(d_carr_freq_error_hz, 0, d_current_correlation_time_s);
                else
                    d_carr_error_filt_hz = d_carrier_loop_filter.get_carrier_error(d_carr_freq_error_hz, d_carr_phase_error_hz, d_current_correlation_time_s);
            }
        else
            {
                d_carr_error_filt_hz = d_carrier_loop_filter.get_carrier_error(0, d_carr_phase_error_hz, d_current_correlation_time_s);
            }
        d_carrier_doppler_hz = d_carr_error_filt_hz;
        if (d_veml)
            d_code_error_chips = dll_nc_vemlp_normalized(d_VE_accu, d_E_accu, d_L_accu, d_VL_accu);
        else
            d_code_error_chips = dll_nc_e_minus_l_normalized(d_E_accu, d_L_accu);
        d_code_error_filt_chips = d_code_loop_filter.apply(d_code_error_chips);
        d_code_freq_chips = (1.0 + (d_carrier_doppler_hz / d_signal_carrier_freq)) * d_code_chip_rate - d_code_error_filt_chips;
    }

    //! (:976-995)
    void clear_tracking_vars()
    {
        std::fill(d_correlator_outs.begin(), d_correlator_outs.end(), gr_complex(0.0, 0.0));
        if (trk_parameters.track_pilot) d_Prompt_Data = gr_complex(0.0, 0.0);
        d_P_accu_old = gr_complex(0.0, 0.0);
        d_current_symbol = 0;
        d_Prompt_circular_buffer.clear();
        d_carr_phase_error_hz = 0.0;
        d_carr_freq_error_hz = 0.0;
        d_carr_error_filt_hz = 0.0;
        d_code_error_chips = 0.0;
        d_code_error_filt_chips = 0.0;
        d_carrier_phase_rate_step_rad = 0.0;
        d_code_phase_rate_step_chips = 0.0;
        d_carr_ph_history.clear();
        d_code_ph_history.clear();
    }

    //! rate smoother used by both NCOs when high_dyn is set (:1016-1033, :1047-1064): mean of the
    //! newer half minus mean of the older half of a 2*smoother_length history, per sample
    double smoothed_rate(std::deque<std::pair<double, double>>& hist, double value, double samples, double current)
    {
        hist.push_back(std::pair<double, double>(value, samples));
        if (hist.size() > d_carr_ph_history_cap) hist.pop_front();
        if (hist.size() < d_carr_ph_history_cap) return current;
        double tmp_cp1 = 0.0, tmp_cp2 = 0.0, tmp_samples = 0.0;
        const unsigned int sl = trk_parameters.smoother_length;
        for (unsigned int k = 0; k < sl; k++)
            {
                tmp_cp1 += hist[k].first;
                tmp_cp2 += hist[sl * 2 - k - 1].first;
                tmp_samples += hist[sl * 2 - k - 1].second;
            }
        tmp_cp1 /= static_cast<double>(sl);
        tmp_cp2 /= static_cast<double>(sl);
        return (tmp_cp2 - tmp_cp1) / tmp_samples;
    }

    //! (:998-1070)
    void update_tracking_vars()
    {
        T_chip_seconds = 1.0 / d_code_freq_chips;
        T_prn_seconds = T_chip_seconds * static_cast<double>(d_code_length_chips);
        T_prn_samples = T_prn_seconds * trk_parameters.fs_in;
        K_blk_samples = T_prn_samples + d_rem_code_phase_samples;
        d_current_prn_length_samples = static_cast<int32_t>(std::floor(K_blk_samples));
        d_carrier_phase_step_rad = PI_2 * d_carrier_doppler_hz / trk_parameters.fs_in;
        if (trk_parameters.high_dyn)
            d_carrier_phase_rate_step_rad = smoothed_rate(d_carr_ph_history, d_carrier_phase_step_rad, static_cast<double>(d_current_prn_length_samples), d_carrier_phase_rate_step_rad);
        const double n = static_cast<double>(d_current_prn_length_samples);
        d_rem_carr_phase_rad += static_cast<float>(d_carrier_phase_step_rad * n + 0.5 * d_carrier_phase_rate_step_rad * n * n);
        d_rem_carr_phase_rad = std::fmod(d_rem_carr_phase_rad, PI_2);
        d_acc_carrier_phase_rad -= (d_carrier_phase_step_rad * n + 0.5 * d_carrier_phase_rate_step_rad * n * n);
        d_code_phase_step_chips = d_code_freq_chips / trk_parameters.fs_in;
        if (trk_parameters.high_dyn)
            d_code_phase_rate_step_chips = smoothed_rate(d_code_ph_history, d_code_phase_step_chips, n, d_code_phase_rate_step_chips);
        d_rem_code_phase_samples = K_blk_samples - n;
        d_rem_code_phase_chips = d_code_freq_chips * d_rem_code_phase_samples / trk_parameters.fs_in;
    }

    //! log_data (:1128-1250): one packed 96-byte record per epoch, the layout
    //! src/utils/matlab/libs/dll_pll_veml_read_tracking_dump.m and tracking_dump_reader.cc read
    void log_data(bool integrating)
    {
        if (!d_dump_file.is_open()) return;
        auto put = [&](const void* p, size_t n) { d_dump_file.write(reinterpret_cast<const char*>(p), n); };
        float f[7];
        f[0] = d_veml ? std::abs(d_VE_accu) : 0.0f;
        f[1] = std::abs(d_E_accu);
        f[2] = std::abs(d_P_accu);
        f[3] = std::abs(d_L_accu);
        f[4] = d_veml ? std::abs(d_VL_accu) : 0.0f;
        if (integrating && d_extend_correlation_symbols_count > 0)
            {
                // partial sums are scaled to the full integration length (:1176-1191)
                const float scale_factor = static_cast<float>(trk_parameters.extend_correlation_symbols) / static_cast<float>(d_extend_correlation_symbols_count);
                for (int i = 0; i < 5; i++) f[i] *= scale_factor;
            }
        const gr_complex p = trk_parameters.track_pilot ? d_Prompt_Data : d_correlator_outs[d_veml ? 2 : 1];
        f[5] = interchange_iq ? p.imag() : p.real();  // prompt I
        f[6] = interchange_iq ? p.real() : p.imag();  // prompt Q
        put(f, sizeof f);
        uint64_t stamp = d_sample_counter + static_cast<uint64_t>(d_current_prn_length_samples);
        put(&stamp, sizeof stamp);
        float g[12];
        g[0] = d_acc_carrier_phase_rad;
        g[1] = d_carrier_doppler_hz;
        g[2] = d_carrier_phase_rate_step_rad * trk_parameters.fs_in * trk_parameters.fs_in / PI_2;
        g[3] = d_code_freq_chips;
        g[4] = d_code_phase_rate_step_chips * trk_parameters.fs_in * trk_parameters.fs_in;
        g[5] = d_carr_phase_error_hz;
        g[6] = d_carr_error_filt_hz;
        g[7] = d_code_error_chips;
        g[8] = d_code_error_filt_chips;
        g[9] = d_CN0_SNV_dB_Hz;
        g[10] = d_carrier_lock_test;
        g[11] = d_rem_code_phase_samples;
        put(g, sizeof g);
        double stamp_d = static_cast<double>(d_sample_counter + d_current_prn_length_samples);
        put(&stamp_d, sizeof stamp_d);
        uint32_t prn_ = d_acquisition_gnss_synchro->PRN;
        put(&prn_, sizeof prn_);
    }

    //! (:839-878)
    bool cn0_and_tracking_lock_status(double coh_integration_time_s)
    {
        if (d_cn0_estimation_counter < trk_parameters.cn0_samples)
            {
                d_Prompt_buffer[d_cn0_estimation_counter] = d_P_accu;
                d_cn0_estimation_counter++;
                return true;
            }
        d_cn0_estimation_counter = 0;
        d_CN0_SNV_dB_Hz = cn0_svn_estimator(d_Prompt_buffer.data(), trk_parameters.cn0_samples, coh_integration_time_s);
        d_carrier_lock_test = carrier_lock_detector(d_Prompt_buffer.data(), trk_parameters.cn0_samples);
        if (!d_pull_in_transitory)
            {
                if (d_carrier_lock_test < d_carrier_lock_threshold or d_CN0_SNV_dB_Hz < trk_parameters.cn0_min)
                    d_carrier_lock_fail_counter++;
                else if (d_carrier_lock_fail_counter > 0)
                    d_carrier_lock_fail_counter--;
            }
        if (d_carrier_lock_fail_counter > trk_parameters.max_lock_fail)
            {
                d_events.push_back(3);  // 3 -> loss of lock
                d_carrier_lock_fail_counter = 0;
                return false;
            }
        return true;
    }

    const double PI_2 = 6.283185307179586;  // GPS_TWO_PI

    Dll_Pll_Conf trk_parameters;
    std::mutex d_setlock;
    std::string signal_type;
    bool d_veml = false;
    bool d_cloop = true;
    bool d_pull_in_transitory = true;
    uint32_t d_channel = 0;
    Gnss_Synchro* d_acquisition_gnss_synchro = nullptr;
    double d_signal_carrier_freq = 0.0;
    double d_code_period = 0.0;
    double d_code_chip_rate = 0.0;
    uint32_t d_code_length_chips = 0;
    uint32_t d_code_samples_per_chip = 0;
    int32_t d_symbols_per_bit = 1;
    int32_t d_correlation_length_ms = 1;
    int32_t d_n_correlator_taps = 3;
    int32_t d_state = 0;
    std::vector<float> d_tracking_code;
    std::vector<float> d_local_code_shift_chips;
    std::vector<gr_complex> d_correlator_outs;
    Hip_Multicorrelator_Real_Codes multicorrelator_cpu;
    Hip_Multicorrelator_Real_Codes correlator_data_cpu;  // pilot tracking: prompt of the data component
    std::vector<float> d_data_code;
    gr_complex d_Prompt_Data;
    bool d_secondary = false;
    bool interchange_iq = false;
    bool d_enable_extended_integration = false;
    std::string d_secondary_code_string;
    std::vector<int32_t> d_preambles_symbols;
    std::deque<gr_complex> d_Prompt_circular_buffer;  // capacity secondary_code_length
    std::deque<float> d_symbol_history;               // capacity preamble_length_symbols
    uint32_t d_current_symbol = 0;
    int32_t d_extend_correlation_symbols_count = 0;
    float d_bit_sync_min_time_s = 10.0f;
    gr_complex d_VE_accu, d_E_accu, d_P_accu, d_P_accu_old, d_L_accu, d_VL_accu;
    Tracking_loop_filter d_code_loop_filter;
    Tracking_FLL_PLL_filter d_carrier_loop_filter;
    double d_acq_code_phase_samples = 0.0;
    double d_acq_carrier_doppler_hz = 0.0;
    uint64_t d_acq_sample_stamp = 0;
    double d_current_correlation_time_s = 0.0;
    double d_carr_phase_error_hz = 0.0, d_carr_freq_error_hz = 0.0, d_carr_error_filt_hz = 0.0;
    double d_code_error_chips = 0.0, d_code_error_filt_chips = 0.0;
    double d_code_freq_chips = 0.0;
    double d_carrier_doppler_hz = 0.0;
    double d_acc_carrier_phase_rad = 0.0;
    double d_rem_code_phase_chips = 0.0;
    double d_rem_code_phase_samples = 0.0;
    double d_code_phase_step_chips = 0.0, d_code_phase_rate_step_chips = 0.0;
    double d_carrier_phase_step_rad = 0.0, d_carrier_phase_rate_step_rad = 0.0;
    float d_rem_carr_phase_rad = 0.0f;
    std::deque<std::pair<double, double>> d_carr_ph_history, d_code_ph_history;
    size_t d_carr_ph_history_cap = 20;
    double T_chip_seconds = 0.0, T_prn_seconds = 0.0, T_prn_samples = 0.0, K_blk_samples = 0.0;
    int32_t d_current_prn_length_samples = 0;
    uint64_t d_sample_counter = 0;
    int32_t d_cn0_estimation_counter = 0;
    int32_t d_carrier_lock_fail_counter = 0;
    double d_carrier_lock_test = 1.0;
    double d_CN0_SNV_dB_Hz = 0.0;
    double d_carrier_lock_threshold = 0.85;
    std::vector<gr_complex> d_Prompt_buffer;
    std::vector<int> d_events;
    std::ofstream d_dump_file;
};

#endif  // GNSSCORR_HIP_DLL_PLL_VEML_TRACKING_H_
