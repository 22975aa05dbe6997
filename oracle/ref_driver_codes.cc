/*
 * ref_driver_codes.cc -- C-linkage callers for the reference's PRN generators
 * (test infrastructure).  gps_sdr_signal_processing.cc, beidou_b1i_signal_processing.cc, glonass_l1_signal_processing.cc,
 * gps_l2c_signal.cc, gps_l5_signal.cc and beidou_b3i_signal_processing.cc are compiled from /root/reference where they
 * lie (oracle/Makefile target `ref`); this file only gives them C names.
 */
#ifdef REF_BDS
#include "beidou_b1i_signal_processing.h"
#endif
#ifdef REF_GPS
#include "gps_sdr_signal_processing.h"
#endif
#ifdef REF_GLO
#include "glonass_l1_signal_processing.h"
#endif
#ifdef REF_WB
#include "beidou_b3i_signal_processing.h"
#include "gps_l2c_signal.h"
#include "gps_l5_signal.h"
#endif
#include <complex>
#include <cstdint>

extern "C" {
#ifdef REF_GPS
void ref_gps_l1_ca_code_gen_int(int32_t* dest, int32_t prn, uint32_t chip_shift)
{
    gps_l1_ca_code_gen_int(dest, prn, chip_shift);
}
void ref_gps_l1_ca_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift)
{
    gps_l1_ca_code_gen_complex_sampled(reinterpret_cast<std::complex<float>*>(dest), prn, fs, chip_shift);
}
#endif
#ifdef REF_GLO
void ref_glonass_l1_ca_code_gen_complex(float* dest, uint32_t chip_shift)
{
    glonass_l1_ca_code_gen_complex(reinterpret_cast<std::complex<float>*>(dest), chip_shift);
}
void ref_glonass_l1_ca_code_gen_complex_sampled(float* dest, int32_t fs, uint32_t chip_shift)
{
    glonass_l1_ca_code_gen_complex_sampled(reinterpret_cast<std::complex<float>*>(dest), fs, chip_shift);
}
#endif
#ifdef REF_WB
// gps_l2c_signal.cc, gps_l5_signal.cc, beidou_b3i_signal_processing.cc (10.23 / 0.5115 Mcps signals)
void ref_gps_l2c_m_code_gen_float(float* dest, uint32_t prn) { gps_l2c_m_code_gen_float(dest, prn); }
void ref_gps_l2c_m_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs)
{
    gps_l2c_m_code_gen_complex_sampled(reinterpret_cast<std::complex<float>*>(dest), prn, fs);
}
void ref_gps_l5i_code_gen_float(float* dest, uint32_t prn) { gps_l5i_code_gen_float(dest, prn); }
void ref_gps_l5q_code_gen_float(float* dest, uint32_t prn) { gps_l5q_code_gen_float(dest, prn); }
void ref_gps_l5i_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs)
{
    gps_l5i_code_gen_complex_sampled(reinterpret_cast<std::complex<float>*>(dest), prn, fs);
}
void ref_gps_l5q_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs)
{
    gps_l5q_code_gen_complex_sampled(reinterpret_cast<std::complex<float>*>(dest), prn, fs);
}
void ref_beidou_b3i_code_gen_int(int32_t* dest, int32_t prn, uint32_t chip_shift) { beidou_b3i_code_gen_int(dest, prn, chip_shift); }
void ref_beidou_b3i_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift)
{
    beidou_b3i_code_gen_complex_sampled(reinterpret_cast<std::complex<float>*>(dest), prn, fs, chip_shift);
}
#endif
#ifdef REF_BDS
void ref_beidou_b1i_code_gen_int(int32_t* dest, int32_t prn, uint32_t chip_shift)
{
    beidou_b1i_code_gen_int(dest, prn, chip_shift);
}
void ref_beidou_b1i_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift)
{
    beidou_b1i_code_gen_complex_sampled(reinterpret_cast<std::complex<float>*>(dest), prn, fs, chip_shift);
}
#endif
}
