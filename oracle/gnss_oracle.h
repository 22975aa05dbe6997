/*
 * gnss_oracle.h -- CPU restatement of the reference's acquisition + tracking
 * correlator hot path (GNSS-SDR: zhufengGNSS/gnss-sdr-1).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the
 * __graft_entry__.smoke() check and bench.py's cpu_baseline leg may call it.
 * The product path (libgnsscorr.so, HIP) never links or loads this file.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - code NCO / resampler chip indices (real, high-dynamics, complex-chip and
 *     int16-chip resamplers), PRN generators (GPS L1 C/A, BeiDou B1I, GLONASS
 *     L1 C/A), running-phase sincos, argmax:  PINNED against the reference's own
 *     sources compiled into oracle/_ref (oracle/Makefile target `ref`).
 *   - PCPS acquisition:              PINNED by the reference's own known-answer
 *     tests (GPS_L1_CA_ID_1_Fs_4Msps_2ms.dat -> 524 samples / 1680 Hz, Galileo
 *     E1 file -> 2920 samples / -632 Hz) and by two real captures the reference
 *     ships: the NT1065 GLONASS L1 file of its tracking tests (the search lands
 *     on the hand-over they hard-code, 1343 samples / -2750 Hz) and the GSoC 2012
 *     Galileo E1 file with its MATLAB analysis (PRN 11 / 12 delays exact; the
 *     listed peak magnitudes, noise floors and their ratios reproduced to ~2e-3,
 *     tests/test_oracle_golden.py); exact grid values (FFTW rounding) are not
 *     reproducible and are PARITY UNPINNED beyond those tests.
 *   - rotator + dot-product accumulate (E/P/L values): PARITY UNPINNED -- the
 *     reference kernel header needs the Mako-generated <volk_gnsssdr/volk_gnsssdr.h>
 *     which cannot be generated here, and no reference test stores E/P/L values.
 *     The restatement follows the generic protokernel statement by statement.
 *
 * All arithmetic is IEEE float32 without FMA contraction (build with
 * -ffp-contract=off), matching a generic (non-FMA) build of the reference.
 */
#ifndef GNSS_ORACLE_H
#define GNSS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- tracking: code NCO ------------------------------------------------- */

/* volk_gnsssdr_32f_xn_resampler_32f_xn_generic
 * (kernels/volk_gnsssdr/volk_gnsssdr_32f_xn_resampler_32f_xn.h:77-94).
 * idx_out (optional, may be NULL): n_taps*N chip indices; res_out (optional):
 * n_taps*N resampled code values, tap-major. */
void orc_resampler(int32_t* idx_out, float* res_out, const float* code,
    float rem_code_phase_chips, float code_phase_step_chips,
    const float* shifts_chips, uint32_t code_length_chips, int n_taps, uint32_t N);

/* volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn_generic
 * (…high_dynamics_resampler_32f_xn.h:81-107), including the unsigned n*n wrap
 * and the sample-shifted copies of tap 0 for taps >= 1. */
void orc_resampler_high_dyn(int32_t* idx_out, float* res_out, const float* code,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    const float* shifts_chips, uint32_t code_length_chips, int n_taps, uint32_t N);

/* ---- tracking: carrier wipe-off + dot products ---------------------------- */

/* volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn_generic
 * (…32fc_32f_rotator_dot_prod_32fc_xn.h:81-113).  result: n_taps complex
 * (interleaved re,im); in: N complex; phase (in/out): 2 floats; in_a: tap-major
 * n_taps*N floats with row stride `lda`. */
void orc_rotator_dot_prod(float* result, const float* in, const float phase_inc[2],
    float phase[2], const float* in_a, uint32_t lda, int n_taps, uint32_t N);

/* volk_gnsssdr_32fc_32f_high_dynamic_rotator_dot_prod_32fc_xn_generic
 * (…high_dynamic_rotator_dot_prod_32fc_xn.h:82-116), libm cpowf as there. */
void orc_rotator_dot_prod_high_dyn(float* result, const float* in, const float phase_inc[2],
    const float phase_inc_rate[2], float phase[2], const float* in_a, uint32_t lda,
    int n_taps, uint32_t N);

/* Cpu_Multicorrelator_Real_Codes::Carrier_wipeoff_multicorrelator_resampler
 * (tracking/libs/cpu_multicorrelator_real_codes.cc:129-152, 7-argument form;
 * high_dyn selects the high-dynamics resampler + rotator as the class flag
 * does).  scratch: n_taps*N floats.  corr_out: n_taps complex. */
void orc_multicorrelator(float* corr_out, const float* sig_in, const float* code,
    uint32_t code_length_chips, const float* shifts_chips, int n_taps,
    float rem_carrier_phase_rad, float phase_step_rad, float phase_rate_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    uint32_t N, int high_dyn, float* scratch);

/* Cpu_Multicorrelator_16sc::Carrier_wipeoff_multicorrelator_resampler (cpu_multicorrelator_16sc.cc:78-103):
 * lv_16sc_t input, chips and output ((re, im) int16 interleaved), saturating int16 accumulation in sample
 * order.  n_taps <= 16.  exact_sums (optional): 2*n_taps unsaturated 32-bit sums of the same products. */
void orc_multicorrelator_16sc(int16_t* corr_out, const int16_t* sig_in, const int16_t* code_iq,
    uint32_t code_length_chips, const float* shifts_chips, int n_taps,
    float rem_carrier_phase_rad, float phase_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, uint32_t N, int32_t* exact_sums);

/* n_iter back-to-back orc_multicorrelator calls (timing helper of bench.py's threaded CPU baseline) */
void orc_multicorrelator_repeat(int n_iter, float* corr_out, const float* sig_in, const float* code,
    uint32_t code_length_chips, const float* shifts_chips, int n_taps,
    float rem_carrier_phase_rad, float phase_step_rad, float phase_rate_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
    uint32_t N, int high_dyn, float* scratch);

/* Cpu_Multicorrelator (complex chips; tracking/libs/cpu_multicorrelator.cc:103-130):
 * volk_gnsssdr_32fc_xn_resampler_32fc_xn_generic (…32fc_xn_resampler_32fc_xn.h:74-91), then
 * volk_gnsssdr_32fc_x2_rotator_dot_prod_32fc_xn_generic (…32fc_x2_rotator_dot_prod_32fc_xn.h:80-111).
 * code_iq: L (re, im) pairs; res_out / in_a: tap-major n_taps*N complex; scratch: 2*n_taps*N floats. */
void orc_resampler_cc(float* res_out, const float* code_iq, float rem_code_phase_chips,
    float code_phase_step_chips, const float* shifts_chips, uint32_t code_length_chips, int n_taps, uint32_t N);
void orc_rotator_dot_prod_cc(float* result, const float* in, const float phase_inc[2],
    float phase[2], const float* in_a, uint32_t lda, int n_taps, uint32_t N);
void orc_multicorrelator_cc(float* corr_out, const float* sig_in, const float* code_iq,
    uint32_t code_length_chips, const float* shifts_chips, int n_taps,
    float rem_carrier_phase_rad, float phase_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, uint32_t N, float* scratch);

/* ---- PRN generators -------------------------------------------------------- */

/* gps_l1_ca_code_gen_int (algorithms/libs/gps_sdr_signal_processing.cc:37-116) */
void orc_gps_l1_ca_code(int32_t* dest /*1023*/, int32_t prn, uint32_t chip_shift);
/* gps_l1_ca_code_gen_complex_sampled (…:151-196); dest: interleaved complex,
 * returns samples per code */
int32_t orc_gps_l1_ca_code_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift);
/* beidou_b1i_code_gen_int (algorithms/libs/beidou_b1i_signal_processing.cc:37-112) */
/* glonass_l1_ca_code_gen_complex / _complex_sampled (algorithms/libs/glonass_l1_signal_processing.cc:37-153) */
void orc_glonass_l1_ca_code(int32_t* dest /*511*/, uint32_t chip_shift);
int32_t orc_glonass_l1_ca_code_sampled(float* dest, int32_t fs, uint32_t chip_shift);
void orc_beidou_b1i_code(int32_t* dest /*2046*/, int32_t prn, uint32_t chip_shift);
int32_t orc_beidou_b1i_code_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift);
/* resampler() (algorithms/libs/gnss_signal_processing.cc:161-182) */
void orc_code_resampler(const float* from, float* dest, float fs_in, float fs_out,
    uint32_t length_in, uint32_t length_out);
/* galileo_e1_code_gen_sinboc11_float (galileo_e1_signal_processing.cc:108-119)
 * from a primary code of 4092 chips (+1/-1) supplied by the caller (the
 * memory codes are ICD data, kept in tests/golden/galileo_e1_codes.bin). */
void orc_galileo_e1_sinboc11(float* dest /*8184*/, const int8_t* primary /*4092*/);
/* galileo_e1_code_gen_float_sampled (…:154-229) without secondary code;
 * cboc selects the 12 samples/chip CBOC(6,1,1/11) replica, is_e1c the sign of
 * the BOC(6,1) term.  Returns samples per code. */
int32_t orc_galileo_e1_code_sampled(float* dest, const int8_t* primary, int cboc, int is_e1c,
    int32_t fs, uint32_t chip_shift);

/* ---- acquisition ----------------------------------------------------------- */

/* volk_gnsssdr_s32f_sincos_32fc_generic (…s32f_sincos_32fc.h:405-415) */
void orc_sincos(float* out /*N complex*/, float phase_inc, float* phase, uint32_t N);
/* volk_gnsssdr_32f_index_max_32u_generic (…32f_index_max_32u.h:460-481) */
uint32_t orc_index_max(const float* src, uint32_t N);

/* Double-precision mixed-radix complex FFT (any N; O(N * sum of prime factors)).
 * inverse != 0 -> exp(+j...) kernel, unnormalised, as FFTW/gr::fft. */
void orc_fft(double* re, double* im, uint32_t N, int inverse);

typedef struct
{
    uint32_t fft_size;         /* d_fft_size (pcps_acquisition.cc:77-85,113-117) */
    uint32_t consumed_samples; /* d_consumed_samples */
    uint32_t effective_fft_size;
    uint32_t num_doppler_bins;
    int32_t doppler_max;
    int32_t doppler_step;
    int64_t fs_in;
    uint32_t samples_per_chip;
    float samples_per_code;
    int bit_transition_flag;
    int use_cfar;     /* d_use_CFAR_algorithm_flag after the max_dwells rule (:152-159) */
    uint32_t max_dwells;
    /* state */
    float* fft_codes;      /* fft_size complex: conj(FFT(code)) */
    float* wipeoffs;       /* num_bins * fft_size complex */
    float* magnitude_grid; /* num_bins * fft_size floats */
    float* tmp_buffer;     /* fft_size floats (d_tmp_buffer: starts zeroed) */
    uint32_t dwell_counter;
} orc_pcps;

typedef struct
{
    uint32_t indext;
    int32_t doppler;
    uint32_t doppler_index;
    float test_statistics;
    float mag;         /* grid maximum (first peak, raw) */
    float input_power; /* 0 when not computed */
    float second_peak; /* bug-compatible second peak (first_vs_second only) */
    float second_peak_fixed; /* second peak with the whole row copied */
    double acq_delay_samples;
    double acq_doppler_hz;
} orc_pcps_result;

/* pcps_acquisition ctor + init (pcps_acquisition.cc:63-190, 313-368) */
orc_pcps* orc_pcps_create(int64_t fs_in, uint32_t sampled_ms, uint32_t ms_per_code,
    float samples_per_ms, float samples_per_code, uint32_t samples_per_chip,
    uint32_t doppler_max, uint32_t doppler_step, uint32_t max_dwells,
    int bit_transition_flag, int use_cfar);
void orc_pcps_destroy(orc_pcps* p);
/* d_old_freq (intermediate frequency / GLONASS FDMA offset): regenerates the coarse wipe-off grid
 * (pcps_acquisition.cc:242-247, :371-380) */
void orc_pcps_set_frequency_offset(orc_pcps* p, int64_t old_freq);
/* pcps_acquisition::set_local_code (:239-274); code: consumed_samples (or
 * fft_size/2 with bit transition) complex samples */
void orc_pcps_set_local_code(orc_pcps* p, const float* code);
/* pcps_acquisition::acquisition_core (:668-770), one dwell; in: consumed_samples
 * complex.  Accumulates into the grid like the reference; call
 * orc_pcps_reset_grid() to start a new search. */
void orc_pcps_core(orc_pcps* p, const float* in, orc_pcps_result* out);
void orc_pcps_reset_grid(orc_pcps* p);

#ifdef __cplusplus
}
#endif
#endif
