/*
 * ref_driver.c -- caller for the parts of the REFERENCE that compile here from
 * their own sources (test infrastructure; see oracle/Makefile target `ref`).
 *
 * The volk_gnsssdr protokernels are `static inline` functions in headers; this
 * translation unit includes those headers WHERE THEY LIE under /root/reference
 * (-I paths only, nothing copied) and exports plain symbols that forward to
 * them.  Only kernels whose headers need nothing generated are included:
 *   volk_gnsssdr_32f_xn_resampler_32f_xn.h, …_high_dynamics_resampler_…,
 *   volk_gnsssdr_32fc_xn_resampler_32fc_xn.h, volk_gnsssdr_16ic_xn_resampler_16ic_xn.h,
 *   volk_gnsssdr_s32f_sincos_32fc.h, volk_gnsssdr_32f_index_max_32u.h.
 * The rotator/dot-product headers include the Mako-generated
 * <volk_gnsssdr/volk_gnsssdr.h>, which does not exist in this image, so they
 * are unbuildable here and are NOT part of this driver.
 */
#define LV_HAVE_GENERIC 1
#ifdef REF_WITH_AVX
#define LV_HAVE_AVX 1
#endif
#include <string.h>
#ifdef REF_WITH_AVX
#include <immintrin.h>
#endif
#include <volk_gnsssdr/volk_gnsssdr_common.h>
#include <volk_gnsssdr/volk_gnsssdr_complex.h>
#include "volk_gnsssdr_32f_xn_resampler_32f_xn.h"
#include "volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn.h"
#include "volk_gnsssdr_32fc_xn_resampler_32fc_xn.h"
#include "volk_gnsssdr_16ic_xn_resampler_16ic_xn.h"
#include "volk_gnsssdr_s32f_sincos_32fc.h"
#include "volk_gnsssdr_32f_index_max_32u.h"

void ref_resampler_generic(float** result, const float* local_code, float rem_code_phase_chips,
    float code_phase_step_chips, float* shifts_chips, unsigned int code_length_chips,
    int num_out_vectors, unsigned int num_points)
{
    volk_gnsssdr_32f_xn_resampler_32f_xn_generic(result, local_code, rem_code_phase_chips,
        code_phase_step_chips, shifts_chips, code_length_chips, num_out_vectors, num_points);
}

void ref_resampler_high_dyn_generic(float** result, const float* local_code, float rem_code_phase_chips,
    float code_phase_step_chips, float code_phase_rate_step_chips, float* shifts_chips,
    unsigned int code_length_chips, int num_out_vectors, unsigned int num_points)
{
    volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn_generic(result, local_code, rem_code_phase_chips,
        code_phase_step_chips, code_phase_rate_step_chips, shifts_chips, code_length_chips,
        num_out_vectors, num_points);
}

void ref_resampler_cc_generic(float** result /* interleaved complex rows */, const float* local_code_iq,
    float rem_code_phase_chips, float code_phase_step_chips, float* shifts_chips, unsigned int code_length_chips,
    int num_out_vectors, unsigned int num_points)
{
    volk_gnsssdr_32fc_xn_resampler_32fc_xn_generic((lv_32fc_t**)result, (const lv_32fc_t*)local_code_iq, rem_code_phase_chips,
        code_phase_step_chips, shifts_chips, code_length_chips, num_out_vectors, num_points);
}

void ref_resampler_16ic_generic(short** result /* interleaved (re, im) int16 rows */, const short* local_code_iq,
    float rem_code_phase_chips, float code_phase_step_chips, float* shifts_chips, unsigned int code_length_chips,
    int num_out_vectors, unsigned int num_points)
{
    volk_gnsssdr_16ic_xn_resampler_16ic_xn_generic((lv_16sc_t**)result, (const lv_16sc_t*)local_code_iq, rem_code_phase_chips,
        code_phase_step_chips, shifts_chips, code_length_chips, num_out_vectors, num_points);
}

#ifdef REF_WITH_AVX
void ref_resampler_u_avx(float** result, const float* local_code, float rem_code_phase_chips,
    float code_phase_step_chips, float* shifts_chips, unsigned int code_length_chips,
    int num_out_vectors, unsigned int num_points)
{
    volk_gnsssdr_32f_xn_resampler_32f_xn_u_avx(result, local_code, rem_code_phase_chips,
        code_phase_step_chips, shifts_chips, code_length_chips, num_out_vectors, num_points);
}
#endif

void ref_sincos_generic(float* out /* interleaved complex */, float phase_inc, float* phase, unsigned int num_points)
{
    volk_gnsssdr_s32f_sincos_32fc_generic((lv_32fc_t*)out, phase_inc, phase, num_points);
}

unsigned int ref_index_max_generic(const float* src, unsigned int num_points)
{
    uint32_t t = 0;
    volk_gnsssdr_32f_index_max_32u_generic(&t, src, num_points);
    return t;
}
