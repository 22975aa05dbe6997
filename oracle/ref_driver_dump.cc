// ref_driver_dump.cc -- C names for the reference's own reader of the tracking dump record
// (src/tests/unit-tests/signal-processing-blocks/libs/tracking_dump_reader.{h,cc}; the record is written by
// dll_pll_veml_tracking::log_data, src/algorithms/tracking/gnuradio_blocks/dll_pll_veml_tracking.cc:1196-1243).
// TEST INFRASTRUCTURE: linked with the reference's tracking_dump_reader.cc, compiled from where it lies, into
// oracle/_ref/libref_dump.so (oracle/Makefile `ref`); nothing of the reference is copied here.
#include "tracking_dump_reader.h"
#include <cstdint>
#include <string>

extern "C" {

// number of records Tracking_Dump_Reader::num_epochs() reports for `path` (-1: cannot open)
long long ref_dump_num_epochs(const char* path)
{
    Tracking_Dump_Reader r;
    if (!r.open_obs_file(std::string(path))) return -1;
    return (long long)r.num_epochs();
}

// reads up to max_records records with read_binary_obs(); per record 20 floats (file order, abs_VE .. aux1 without the integers) into
// f32[n][20], PRN_start_sample_count into u64[n], aux2 into f64[n], PRN into u32[n]; returns the number read
long long ref_dump_read(const char* path, long long max_records, float* f32, uint64_t* u64, double* f64, unsigned int* u32)
{
    Tracking_Dump_Reader r;
    if (!r.open_obs_file(std::string(path))) return -1;
    const long long n_file = (long long)r.num_epochs();
    long long n = 0;
    while (n < max_records && n < n_file && r.read_binary_obs())
        {
            float* o = f32 + 20 * n;
            o[0] = r.abs_VE;
            o[1] = r.abs_E;
            o[2] = r.abs_P;
            o[3] = r.abs_L;
            o[4] = r.abs_VL;
            o[5] = r.prompt_I;
            o[6] = r.prompt_Q;
            o[7] = r.acc_carrier_phase_rad;
            o[8] = r.carrier_doppler_hz;
            o[9] = r.carrier_doppler_rate_hz_s;
            o[10] = r.code_freq_chips;
            o[11] = r.code_freq_rate_chips;
            o[12] = r.carr_error_hz;
            o[13] = r.carr_error_filt_hz;
            o[14] = r.code_error_chips;
            o[15] = r.code_error_filt_chips;
            o[16] = r.CN0_SNV_dB_Hz;
            o[17] = r.carrier_lock_test;
            o[18] = r.aux1;
            o[19] = 0.0f;
            u64[n] = r.PRN_start_sample_count;
            f64[n] = r.aux2;
            u32[n] = r.PRN;
            n++;
        }
    return n;
}
}
