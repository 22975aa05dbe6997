/*
 * gnss_oracle.c -- CPU restatement of the reference hot path.  TEST
 * INFRASTRUCTURE ONLY (see gnss_oracle.h for the contract and the pinning
 * status of each function).  Build: see oracle/Makefile (-ffp-contract=off,
 * no -ffast-math: every float operation below rounds exactly once, in the
 * order the reference's generic kernels evaluate it).
 */
#include "gnss_oracle.h"
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* tracking: code NCO                                                         */
/* ------------------------------------------------------------------------- */

/* chip index of the generic resampler: (int)floor(step*(float)n + shift - rem)
 * then the reference's negative fix-up and modulo
 * (volk_gnsssdr_32f_xn_resampler_32f_xn.h:88-91). */
static inline int32_t wrap_chip(int32_t i, uint32_t L)
{
    if (i < 0) i += (int32_t)L * (abs(i) / (int32_t)L + 1);
    return i % (int32_t)L;
}

void orc_resampler(int32_t* idx_out, float* res_out, const float* code,
    float rem, float step, const float* shifts, uint32_t L, int n_taps, uint32_t N)
{
    for (int t = 0; t < n_taps; t++)
        {
            for (uint32_t n = 0; n < N; n++)
                {
                    float a = step * (float)n;
                    float b = a + shifts[t];
                    float c = b - rem;
                    int32_t i = wrap_chip((int32_t)floor(c), L);
                    if (idx_out) idx_out[(size_t)t * N + n] = i;
                    if (res_out) res_out[(size_t)t * N + n] = code[i];
                }
        }
}

void orc_resampler_high_dyn(int32_t* idx_out, float* res_out, const float* code,
    float rem, float step, float rate, const float* shifts, uint32_t L, int n_taps, uint32_t N)
{
    /* first correlator (…high_dynamics_resampler_32f_xn.h:88-96): the rate
     * term uses (float)(n * n) with UNSIGNED n, i.e. it wraps at 2^32 */
    int32_t* idx0 = (int32_t*)malloc(sizeof(int32_t) * (N ? N : 1));
    for (uint32_t n = 0; n < N; n++)
        {
            float a = step * (float)n;
            float r = rate * (float)(n * n);
            float b = a + r;
            float c = b + shifts[0];
            float d = c - rem;
            idx0[n] = wrap_chip((int32_t)floor(d), L);
        }
    for (uint32_t n = 0; n < N; n++)
        {
            if (idx_out) idx_out[n] = idx0[n];
            if (res_out) res_out[n] = code[idx0[n]];
        }
    /* adjacent correlators are sample-shifted copies of tap 0 (:99-106) */
    uint32_t shift_samples = 0;
    for (int t = 1; t < n_taps; t++)
        {
            shift_samples += (int)round((shifts[t] - shifts[t - 1]) / step);
            for (uint32_t n = 0; n < N; n++)
                {
                    uint32_t src = (n < N - shift_samples) ? n + shift_samples : n - (N - shift_samples);
                    if (idx_out) idx_out[(size_t)t * N + n] = idx0[src];
                    if (res_out) res_out[(size_t)t * N + n] = code[idx0[src]];
                }
        }
    free(idx0);
}

/* ------------------------------------------------------------------------- */
/* tracking: rotator + dot products                                           */
/* ------------------------------------------------------------------------- */

/* C99 complex float product as libgcc's __mulsc3 evaluates it for finite
 * operands: (ac - bd) + j(ad + bc), four products, one subtraction, one
 * addition, each rounded to float. */
static inline void cmulf(float* re, float* im, float a, float b, float c, float d)
{
    float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    *re = ac - bd;
    *im = ad + bc;
}

void orc_rotator_dot_prod(float* result, const float* in, const float phase_inc[2],
    float phase[2], const float* in_a, uint32_t lda, int n_taps, uint32_t N)
{
    float pr = phase[0], pi = phase[1];
    for (int t = 0; t < n_taps; t++) result[2 * t] = result[2 * t + 1] = 0.0f;
    for (uint32_t n = 0; n < N; n++)
        {
            float tr, ti;
            cmulf(&tr, &ti, in[2 * n], in[2 * n + 1], pr, pi); /* tmp32_1 = in * phase (:92) */
            if (n % 256 == 0)
                { /* (*phase) /= hypotf(re, im)  (:95-103) */
                    float h = hypotf(pr, pi);
                    pr = pr / h;
                    pi = pi / h;
                }
            float nr, ni;
            cmulf(&nr, &ni, pr, pi, phase_inc[0], phase_inc[1]); /* (*phase) *= phase_inc (:106) */
            pr = nr;
            pi = ni;
            for (int t = 0; t < n_taps; t++)
                { /* result += tmp32_1 * in_a[t][n]  (:107-111) */
                    float a = in_a[(size_t)t * lda + n];
                    result[2 * t] += tr * a;
                    result[2 * t + 1] += ti * a;
                }
        }
    phase[0] = pr;
    phase[1] = pi;
}

void orc_rotator_dot_prod_high_dyn(float* result, const float* in, const float phase_inc[2],
    const float phase_inc_rate[2], float phase[2], const float* in_a, uint32_t lda,
    int n_taps, uint32_t N)
{
    /* …high_dynamic_rotator_dot_prod_32fc_xn.h:82-116 */
    float pr = phase[0], pi = phase[1];
    float dr = pr, di = pi; /* phase_doppler */
    float complex rate = phase_inc_rate[0] + I * phase_inc_rate[1];
    for (int t = 0; t < n_taps; t++) result[2 * t] = result[2 * t + 1] = 0.0f;
    for (uint32_t n = 0; n < N; n++)
        {
            if (n % 256 == 0)
                {
                    float h = hypotf(pr, pi);
                    pr = pr / h;
                    pi = pi / h;
                }
            float tr, ti;
            cmulf(&tr, &ti, in[2 * n], in[2 * n + 1], pr, pi);
            float nr, ni;
            cmulf(&nr, &ni, dr, di, phase_inc[0], phase_inc[1]);
            dr = nr;
            di = ni;
            float complex pdr = cpowf(rate, (float)(n * n) + 0.0f * I); /* unsigned n*n */
            float h2 = hypotf(crealf(pdr), cimagf(pdr));
            float qr = crealf(pdr) / h2, qi = cimagf(pdr) / h2;
            cmulf(&pr, &pi, dr, di, qr, qi);
            for (int t = 0; t < n_taps; t++)
                {
                    float a = in_a[(size_t)t * lda + n];
                    result[2 * t] += tr * a;
                    result[2 * t + 1] += ti * a;
                }
        }
    phase[0] = pr;
    phase[1] = pi;
}

/* ---- complex-code correlator (Cpu_Multicorrelator) ------------------------ */

void orc_resampler_cc(float* res_out, const float* code_iq, float rem, float step, const float* shifts,
    uint32_t L, int n_taps, uint32_t N)
{
    /* volk_gnsssdr_32fc_xn_resampler_32fc_xn.h:74-91: the chip index expression is the one of the
     * real-code resampler (…32f_xn_resampler_32f_xn.h:77-94); only the gathered element is complex */
    for (int t = 0; t < n_taps; t++)
        {
            for (uint32_t n = 0; n < N; n++)
                {
                    int i = (int)floor(step * (float)n + shifts[t] - rem);
                    if (i < 0) i += (int)L * (abs(i) / L + 1);
                    i = i % L;
                    res_out[2 * ((size_t)t * N + n)] = code_iq[2 * i];
                    res_out[2 * ((size_t)t * N + n) + 1] = code_iq[2 * i + 1];
                }
        }
}

void orc_rotator_dot_prod_cc(float* result, const float* in, const float phase_inc[2],
    float phase[2], const float* in_a, uint32_t lda, int n_taps, uint32_t N)
{
    /* volk_gnsssdr_32fc_x2_rotator_dot_prod_32fc_xn.h:80-111 */
    float pr = phase[0], pi = phase[1];
    for (int t = 0; t < n_taps; t++) result[2 * t] = result[2 * t + 1] = 0.0f;
    for (uint32_t n = 0; n < N; n++)
        {
            float tr, ti;
            cmulf(&tr, &ti, in[2 * n], in[2 * n + 1], pr, pi); /* tmp32_1 = in * phase (:90) */
            if (n % 256 == 0)
                { /* (:93-101) */
                    float h = hypotf(pr, pi);
                    pr = pr / h;
                    pi = pi / h;
                }
            float nr, ni;
            cmulf(&nr, &ni, pr, pi, phase_inc[0], phase_inc[1]); /* (:104) */
            pr = nr;
            pi = ni;
            for (int t = 0; t < n_taps; t++)
                { /* result += tmp32_1 * in_a[t][n]  (:105-109) */
                    float cr, ci;
                    cmulf(&cr, &ci, tr, ti, in_a[2 * ((size_t)t * lda + n)], in_a[2 * ((size_t)t * lda + n) + 1]);
                    result[2 * t] += cr;
                    result[2 * t + 1] += ci;
                }
        }
    phase[0] = pr;
    phase[1] = pi;
}

void orc_multicorrelator_cc(float* corr_out, const float* sig_in, const float* code_iq,
    uint32_t L, const float* shifts, int n_taps,
    float rem_carr, float phase_step, float rem_code, float code_step, uint32_t N, float* scratch)
{
    /* cpu_multicorrelator.cc:103-130 */
    orc_resampler_cc(scratch, code_iq, rem_code, code_step, shifts, L, n_taps, N);
    float phase[2] = {cosf(rem_carr), -sinf(rem_carr)};
    float complex e = cexpf(0.0f - I * phase_step);
    float inc[2] = {crealf(e), cimagf(e)};
    orc_rotator_dot_prod_cc(corr_out, sig_in, inc, phase, scratch, N, n_taps, N);
}

void orc_multicorrelator(float* corr_out, const float* sig_in, const float* code,
    uint32_t L, const float* shifts, int n_taps,
    float rem_carr, float phase_step, float phase_rate_step,
    float rem_code, float code_step, float code_rate_step,
    uint32_t N, int high_dyn, float* scratch)
{
    /* cpu_multicorrelator_real_codes.cc:129-152 */
    if (high_dyn)
        orc_resampler_high_dyn(NULL, scratch, code, rem_code, code_step, code_rate_step, shifts, L, n_taps, N);
    else
        orc_resampler(NULL, scratch, code, rem_code, code_step, shifts, L, n_taps, N);
    float phase[2] = {cosf(rem_carr), -sinf(rem_carr)};
    /* std::exp(lv_32fc_t(0.0, -phase_step_rad)): libstdc++ -> cexpf(0 - j*step)
     * = (cosf(step), sinf(-step)) for a zero real part */
    float complex e = cexpf(0.0f - I * phase_step);
    float inc[2] = {crealf(e), cimagf(e)};
    if (high_dyn)
        {
            float complex er = cexpf(0.0f - I * phase_rate_step);
            float rate[2] = {crealf(er), cimagf(er)};
            orc_rotator_dot_prod_high_dyn(corr_out, sig_in, inc, rate, phase, scratch, N, n_taps, N);
        }
    else
        {
            orc_rotator_dot_prod(corr_out, sig_in, inc, phase, scratch, N, n_taps, N);
        }
}

/* ---- 16-bit correlator (Cpu_Multicorrelator_16sc) ---------------------------- */

static inline int16_t sat_adds16(int16_t x, int16_t y)
{
    /* saturation_arithmetic.h:30-38 */
    int32_t r = (int32_t)x + (int32_t)y;
    if (r < -32768) r = -32768;
    if (r > 32767) r = 32767;
    return (int16_t)r;
}

void orc_multicorrelator_16sc(int16_t* corr_out, const int16_t* sig_in, const int16_t* code_iq,
    uint32_t L, const float* shifts, int n_taps,
    float rem_carr, float phase_step, float rem_code, float code_step, uint32_t N, int32_t* exact_sums)
{
    /* cpu_multicorrelator_16sc.cc:78-103: volk_gnsssdr_16ic_xn_resampler_16ic_xn_generic
     * (…16ic_xn_resampler_16ic_xn.h:74-91, same chip index expression as the float resamplers), then
     * volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn_generic (…16ic_x2_rotator_dot_prod_16ic_xn.h:80-110).
     * exact_sums (optional, 2*n_taps): the same products summed without saturation. */
    float pr = cosf(rem_carr), pi = -sinf(rem_carr);
    float complex e = cexpf(0.0f - I * phase_step);
    const float ir = crealf(e), ii = cimagf(e);
    int16_t accr[16], acci[16];
    int32_t exr[16], exi[16];
    for (int t = 0; t < n_taps; t++) accr[t] = acci[t] = 0, exr[t] = exi[t] = 0;
    for (uint32_t n = 0; n < N; n++)
        {
            /* tmp32 = (float)in * phase; tmp16 = (int16_t)rintf(.)  (:92-94) */
            float tr, ti;
            cmulf(&tr, &ti, (float)sig_in[2 * n], (float)sig_in[2 * n + 1], pr, pi);
            const int16_t yr = (int16_t)rintf(tr), yi = (int16_t)rintf(ti);
            if (n % 256 == 0)
                { /* (:97-105) */
                    float h = hypotf(pr, pi);
                    pr = pr / h;
                    pi = pi / h;
                }
            float nr, ni;
            cmulf(&nr, &ni, pr, pi, ir, ii); /* (:107) */
            pr = nr;
            pi = ni;
            for (int t = 0; t < n_taps; t++)
                {
                    int i = (int)floor(code_step * (float)n + shifts[t] - rem_code);
                    if (i < 0) i += (int)L * (abs(i) / L + 1);
                    i = i % L;
                    const int cr = code_iq[2 * i], ci = code_iq[2 * i + 1];
                    /* lv_16sc_t tmp = tmp16 * in_a[n]: evaluated in int, stored as int16 (:110) */
                    const int16_t mr = (int16_t)(yr * cr - yi * ci);
                    const int16_t mi = (int16_t)(yr * ci + yi * cr);
                    accr[t] = sat_adds16(accr[t], mr); /* (:112) */
                    acci[t] = sat_adds16(acci[t], mi);
                    exr[t] += mr;
                    exi[t] += mi;
                }
        }
    for (int t = 0; t < n_taps; t++)
        {
            corr_out[2 * t] = accr[t];
            corr_out[2 * t + 1] = acci[t];
            if (exact_sums)
                {
                    exact_sums[2 * t] = exr[t];
                    exact_sums[2 * t + 1] = exi[t];
                }
        }
}

/* timing helper for bench.py's multi-thread CPU baseline: n_iter back-to-back calls without
 * returning to the interpreter (each call rewrites corr_out, so nothing is hoisted) */
void orc_multicorrelator_repeat(int n_iter, float* corr_out, const float* sig_in, const float* code,
    uint32_t L, const float* shifts, int n_taps,
    float rem_carr, float phase_step, float phase_rate_step,
    float rem_code, float code_step, float code_rate_step,
    uint32_t N, int high_dyn, float* scratch)
{
    for (int i = 0; i < n_iter; i++)
        {
            orc_multicorrelator(corr_out, sig_in, code, L, shifts, n_taps, rem_carr + 1e-3f * (float)i, phase_step, phase_rate_step,
                rem_code, code_step, code_rate_step, N, high_dyn, scratch);
            __asm__ volatile("" ::: "memory");
        }
}

/* ------------------------------------------------------------------------- */
/* PRN generators                                                             */
/* ------------------------------------------------------------------------- */

static inline int32_t aux_ceil(float x) { return (int32_t)(int64_t)(x + 1); } /* gps_sdr_signal_processing.cc:35 */

void orc_gps_l1_ca_code(int32_t* dest, int32_t prn, uint32_t chip_shift)
{
    /* gps_sdr_signal_processing.cc:37-116; G2 delays of IS-GPS-200 */
    enum { CL = 1023 };
    static const int32_t delays[51] = {5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470, 471, 472,
        473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862,
        145, 175, 52, 21, 237, 235, 886, 657, 634, 762, 355, 1012, 176, 603, 130, 359, 595, 68, 386};
    unsigned char G1[CL], G2[CL], r1[10], r2[10];
    int32_t prn_idx = (120 <= prn && prn <= 138) ? prn - 88 : prn - 1;
    if (prn_idx < 0 || prn_idx > 50) return;
    for (int i = 0; i < 10; i++) r1[i] = r2[i] = 1;
    for (int i = 0; i < CL; i++)
        {
            G1[i] = r1[0];
            G2[i] = r2[0];
            unsigned char f1 = r1[7] ^ r1[0];
            unsigned char f2 = (r2[8] + r2[7] + r2[4] + r2[2] + r2[1] + r2[0]) & 1;
            for (int k = 0; k < 9; k++)
                {
                    r1[k] = r1[k + 1];
                    r2[k] = r2[k + 1];
                }
            r1[9] = f1;
            r2[9] = f2;
        }
    uint32_t delay = CL - delays[prn_idx];
    delay += chip_shift;
    delay %= CL;
    for (uint32_t i = 0; i < CL; i++)
        {
            dest[i] = (G1[(i + chip_shift) % CL] ^ G2[delay]) ? 1 : -1;
            delay = (delay + 1) % CL;
        }
}

static int32_t sample_code(float* dest, const int32_t* chips, int32_t code_len, int32_t code_freq, int32_t fs)
{
    /* gps_sdr_signal_processing.cc:151-196 / beidou_b1i_signal_processing.cc:146-191 */
    int32_t spc = (int32_t)((double)fs / (double)(code_freq / code_len));
    float ts = 1.0 / (float)fs;
    float tc = 1.0 / (float)code_freq;
    for (int32_t i = 0; i < spc; i++)
        {
            float aux = (ts * (i + 1)) / tc;
            int32_t k = aux_ceil(aux) - 1;
            int32_t v = (i == spc - 1) ? chips[code_len - 1] : chips[k];
            dest[2 * i] = (float)v;
            dest[2 * i + 1] = 0.0f;
        }
    return spc;
}

int32_t orc_gps_l1_ca_code_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift)
{
    int32_t chips[1023];
    memset(chips, 0, sizeof chips);
    orc_gps_l1_ca_code(chips, (int32_t)prn, chip_shift);
    return sample_code(dest, chips, 1023, 1023000, fs);
}

void orc_glonass_l1_ca_code(int32_t* dest, uint32_t chip_shift)
{
    /* glonass_l1_signal_processing.cc:37-97: 9-stage register, all ones, output stage index 2, feedback 4 ^ 0 */
    enum { CL = 511 };
    unsigned char G1[CL], reg[9];
    for (int i = 0; i < 9; i++) reg[i] = 1;
    for (int i = 0; i < CL; i++)
        {
            G1[i] = reg[2];
            unsigned char fb = reg[4] ^ reg[0];
            for (int k = 0; k < 8; k++) reg[k] = reg[k + 1];
            reg[8] = fb;
        }
    for (uint32_t i = 0; i < CL; i++) dest[i] = G1[(i + chip_shift) % CL] ? 1 : -1;
}

int32_t orc_glonass_l1_ca_code_sampled(float* dest, int32_t fs, uint32_t chip_shift)
{
    /* glonass_l1_signal_processing.cc:103-153 */
    int32_t code[511];
    const int32_t code_freq = 511000, code_len = 511;
    const int32_t spc = (int32_t)((double)fs / (double)(code_freq / code_len));
    const float ts = 1.0 / (float)fs, tc = 1.0 / (float)code_freq;
    orc_glonass_l1_ca_code(code, chip_shift);
    for (int32_t i = 0; i < spc; i++)
        {
            float aux = (ts * (i + 1)) / tc;
            int32_t k = aux_ceil(aux) - 1;
            dest[2 * i] = (float)((i == spc - 1) ? code[code_len - 1] : code[k]);
            dest[2 * i + 1] = 0.0f;
        }
    return spc;
}

void orc_beidou_b1i_code(int32_t* dest, int32_t prn, uint32_t chip_shift)
{
    /* beidou_b1i_signal_processing.cc:37-112 */
    enum { CL = 2046 };
    static const int32_t phase1[37] = {1, 1, 1, 1, 1, 1, 1, 1, 2, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 6, 6, 6, 6, 8, 8, 8, 9, 9, 10};
    static const int32_t phase2[37] = {3, 4, 5, 6, 8, 9, 10, 11, 7, 4, 5, 6, 8, 9, 10, 11, 5, 6, 8, 9, 10, 11, 6, 8, 9, 10, 11, 8, 9, 10, 11, 9, 10, 11, 10, 11, 11};
    unsigned char G1[CL], G2[CL];
    unsigned char r1[11] = {0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0};
    unsigned char r2[11] = {0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0};
    int32_t prn_idx = prn - 1;
    if (prn_idx < 0 || prn_idx > 32) return;
    for (int i = 0; i < CL; i++)
        {
            G1[i] = r1[0];
            G2[i] = r2[-(phase1[prn_idx] - 11)] ^ r2[-(phase2[prn_idx] - 11)];
            unsigned char f1 = (r1[0] + r1[1] + r1[2] + r1[3] + r1[4] + r1[10]) & 1;
            unsigned char f2 = (r2[0] + r2[2] + r2[3] + r2[6] + r2[7] + r2[8] + r2[9] + r2[10]) & 1;
            for (int k = 0; k < 10; k++)
                {
                    r1[k] = r1[k + 1];
                    r2[k] = r2[k + 1];
                }
            r1[10] = f1;
            r2[10] = f2;
        }
    uint32_t delay = CL; /* "delays[prn_idx] * 0" in the reference (:86) */
    delay += chip_shift;
    delay %= CL;
    for (uint32_t i = 0; i < CL; i++)
        {
            dest[i] = (G1[(i + chip_shift) % CL] ^ G2[delay]) ? 1 : -1;
            delay = (delay + 1) % CL;
        }
}

int32_t orc_beidou_b1i_code_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift)
{
    int32_t chips[2046];
    memset(chips, 0, sizeof chips);
    orc_beidou_b1i_code(chips, (int32_t)prn, chip_shift);
    return sample_code(dest, chips, 2046, 2046000, fs);
}

void orc_code_resampler(const float* from, float* dest, float fs_in, float fs_out,
    uint32_t length_in, uint32_t length_out)
{
    /* gnss_signal_processing.cc:161-182 */
    const float t_in = 1 / fs_in;
    const float t_out = 1 / fs_out;
    for (uint32_t i = 0; i < length_out - 1; i++)
        {
            float aux = (t_out * (i + 1)) / t_in;
            uint32_t k = (uint32_t)(aux_ceil(aux) - 1);
            dest[i] = from[k];
        }
    dest[length_out - 1] = from[length_in - 1];
}

void orc_galileo_e1_sinboc11(float* dest, const int8_t* primary)
{
    /* galileo_e1_signal_processing.cc:108-119 */
    for (uint32_t i = 0; i < 4092; i++)
        {
            dest[2 * i] = (float)primary[i];
            dest[2 * i + 1] = -dest[2 * i];
        }
}

int32_t orc_galileo_e1_code_sampled(float* dest, const int8_t* primary, int cboc, int is_e1c,
    int32_t fs, uint32_t chip_shift)
{
    /* galileo_e1_signal_processing.cc:154-229, _secondary_flag = false */
    const int32_t code_freq = 1023000;
    const uint32_t CL = 4092;
    uint32_t spc = (uint32_t)((double)fs / ((double)code_freq / (double)CL));
    const int32_t samples_per_chip = cboc ? 12 : 2;
    const uint32_t delay = (((int32_t)CL - chip_shift) % (int32_t)CL) * spc / CL;
    uint32_t code_len = samples_per_chip * CL;
    float* sig = (float*)malloc(sizeof(float) * code_len);
    if (cboc)
        {
            /* galileo_e1_gen_float (:122-151) with sinboc(1,1) and sinboc(6,1)
             * at 12 samples per chip (:72-105) */
            const float alpha = sqrt(10.0 / 11.0);
            const float beta = sqrt(1.0 / 11.0);
            for (uint32_t i = 0; i < CL; i++)
                {
                    for (uint32_t j = 0; j < 12; j++)
                        {
                            int32_t s11 = (j < 6) ? primary[i] : -primary[i];
                            int32_t s61 = (j % 2 == 0) ? primary[i] : -primary[i];
                            if (is_e1c)
                                sig[i * 12 + j] = alpha * (float)s11 - beta * (float)s61;
                            else
                                sig[i * 12 + j] = alpha * (float)s11 + beta * (float)s61;
                        }
                }
        }
    else
        {
            for (uint32_t i = 0; i < CL; i++)
                {
                    sig[2 * i] = (float)primary[i];
                    sig[2 * i + 1] = (float)(-primary[i]);
                }
        }
    if (fs != samples_per_chip * code_freq)
        {
            float* rs = (float*)malloc(sizeof(float) * spc);
            orc_code_resampler(sig, rs, (float)(samples_per_chip * code_freq), (float)fs, code_len, spc);
            free(sig);
            sig = rs;
        }
    for (uint32_t i = 0; i < spc; i++) dest[(i + delay) % spc] = sig[i];
    free(sig);
    return (int32_t)spc;
}

/* ------------------------------------------------------------------------- */
/* acquisition                                                                */
/* ------------------------------------------------------------------------- */

void orc_sincos(float* out, float phase_inc, float* phase, uint32_t N)
{
    float p = *phase;
    for (uint32_t i = 0; i < N; i++)
        {
            out[2 * i] = cosf(p);
            out[2 * i + 1] = sinf(p);
            p += phase_inc;
        }
    *phase = p;
}

uint32_t orc_index_max(const float* src, uint32_t N)
{
    if (N == 0) return 0;
    float max = src[0];
    uint32_t index = 0;
    for (uint32_t i = 1; i < N; ++i)
        {
            if (src[i] > max)
                {
                    index = i;
                    max = src[i];
                }
        }
    return index;
}

/* Radix-p decimation-in-frequency Stockham autosort passes over the prime
 * factors of N, float64.  Stands in for FFTW (gr::fft::fft_complex), whose
 * float32 rounding is not reproducible; see the header for what that means
 * for parity. */
void orc_fft(double* re, double* im, uint32_t N, int inverse)
{
    if (N <= 1) return;
    double* wr = (double*)malloc(sizeof(double) * N);
    double* wi = (double*)malloc(sizeof(double) * N);
    double* yr = (double*)malloc(sizeof(double) * N);
    double* yi = (double*)malloc(sizeof(double) * N);
    const double sgn = inverse ? 1.0 : -1.0;
    for (uint32_t k = 0; k < N; k++)
        {
            double a = 2.0 * M_PI * (double)k / (double)N;
            wr[k] = cos(a);
            wi[k] = sgn * sin(a);
        }
    double *xr = re, *xi = im, *or_ = yr, *oi = yi;
    uint32_t n = N, s = 1;
    while (n > 1)
        {
            uint32_t p = 2;
            while (n % p) p++;
            uint32_t m = n / p;
            uint32_t tw_n = N / n; /* w_n^k = w_N^(k*N/n) */
            uint32_t tw_p = N / p;
            for (uint32_t q = 0; q < m; q++)
                {
                    for (uint32_t r = 0; r < s; r++)
                        {
                            for (uint32_t k = 0; k < p; k++)
                                {
                                    double ar = 0.0, ai = 0.0;
                                    for (uint32_t j = 0; j < p; j++)
                                        {
                                            uint32_t src = r + s * (q + m * j);
                                            uint32_t w = (uint32_t)(((uint64_t)j * k % p) * tw_p);
                                            ar += xr[src] * wr[w] - xi[src] * wi[w];
                                            ai += xr[src] * wi[w] + xi[src] * wr[w];
                                        }
                                    uint32_t w = (uint32_t)(((uint64_t)q * k) % n) * tw_n;
                                    uint32_t dst = r + s * (p * q + k);
                                    or_[dst] = ar * wr[w] - ai * wi[w];
                                    oi[dst] = ar * wi[w] + ai * wr[w];
                                }
                        }
                }
            double* t;
            t = xr, xr = or_, or_ = t;
            t = xi, xi = oi, oi = t;
            n = m;
            s *= p;
        }
    if (xr != re)
        {
            memcpy(re, xr, sizeof(double) * N);
            memcpy(im, xi, sizeof(double) * N);
        }
    free(wr);
    free(wi);
    free(yr);
    free(yi);
}

/* float32 in -> float64 FFT -> float32 out (interleaved complex) */
static void fft_c32(float* out, const float* in, uint32_t N, int inverse)
{
    double* re = (double*)malloc(sizeof(double) * N);
    double* im = (double*)malloc(sizeof(double) * N);
    for (uint32_t i = 0; i < N; i++)
        {
            re[i] = in[2 * i];
            im[i] = in[2 * i + 1];
        }
    orc_fft(re, im, N, inverse);
    for (uint32_t i = 0; i < N; i++)
        {
            out[2 * i] = (float)re[i];
            out[2 * i + 1] = (float)im[i];
        }
    free(re);
    free(im);
}

orc_pcps* orc_pcps_create(int64_t fs_in, uint32_t sampled_ms, uint32_t ms_per_code,
    float samples_per_ms, float samples_per_code, uint32_t samples_per_chip,
    uint32_t doppler_max, uint32_t doppler_step, uint32_t max_dwells,
    int bit_transition_flag, int use_cfar)
{
    orc_pcps* p = (orc_pcps*)calloc(1, sizeof(orc_pcps));
    /* pcps_acquisition.cc:77-85,113-117 */
    p->consumed_samples = (uint32_t)(sampled_ms * samples_per_ms * (bit_transition_flag ? 2 : 1));
    p->fft_size = (sampled_ms == ms_per_code) ? p->consumed_samples : p->consumed_samples * 2;
    if (bit_transition_flag)
        {
            p->fft_size = p->consumed_samples * 2;
            max_dwells = 1;
        }
    p->effective_fft_size = bit_transition_flag ? p->fft_size / 2 : p->fft_size;
    p->bit_transition_flag = bit_transition_flag;
    p->max_dwells = max_dwells;
    p->use_cfar = (max_dwells == 1) ? use_cfar : 0; /* :152-159 */
    p->fs_in = fs_in;
    p->samples_per_chip = samples_per_chip;
    p->samples_per_code = samples_per_code;
    p->doppler_max = (int32_t)doppler_max;
    p->doppler_step = (int32_t)doppler_step;
    /* init(): :326 */
    p->num_doppler_bins = (uint32_t)ceil((double)((int32_t)doppler_max - (int32_t)(-(int32_t)doppler_max)) / (double)doppler_step);
    p->fft_codes = (float*)calloc((size_t)p->fft_size * 2, sizeof(float));
    p->tmp_buffer = (float*)calloc(p->fft_size, sizeof(float));
    p->wipeoffs = (float*)malloc(sizeof(float) * 2 * (size_t)p->fft_size * p->num_doppler_bins);
    p->magnitude_grid = (float*)calloc((size_t)p->fft_size * p->num_doppler_bins, sizeof(float));
    for (uint32_t d = 0; d < p->num_doppler_bins; d++)
        {
            /* :355-356 + update_local_carrier (:296-310) */
            int32_t doppler = -(int32_t)doppler_max + (int32_t)doppler_step * (int32_t)d;
            float freq = (float)doppler; /* d_old_freq + doppler, passed as float */
            float phase_step_rad = (float)(6.283185307179586 * freq / (float)fs_in);
            float ph = 0.0f;
            orc_sincos(p->wipeoffs + 2 * (size_t)d * p->fft_size, -phase_step_rad, &ph, p->fft_size);
        }
    return p;
}

void orc_pcps_set_frequency_offset(orc_pcps* p, int64_t old_freq)
{
    /* set_local_code's FDMA branch (:242-247) -> update_grid_doppler_wipeoffs (:371-380): d_old_freq + doppler */
    for (uint32_t d = 0; d < p->num_doppler_bins; d++)
        {
            int32_t doppler = -p->doppler_max + p->doppler_step * (int32_t)d;
            float freq = (float)(old_freq + (int64_t)doppler);
            float phase_step_rad = (float)(6.283185307179586 * freq / (float)p->fs_in);
            float ph = 0.0f;
            orc_sincos(p->wipeoffs + 2 * (size_t)d * p->fft_size, -phase_step_rad, &ph, p->fft_size);
        }
}

void orc_pcps_destroy(orc_pcps* p)
{
    if (!p) return;
    free(p->fft_codes);
    free(p->tmp_buffer);
    free(p->wipeoffs);
    free(p->magnitude_grid);
    free(p);
}

void orc_pcps_set_local_code(orc_pcps* p, const float* code)
{
    /* :239-274 */
    float* buf = (float*)calloc((size_t)p->fft_size * 2, sizeof(float));
    if (p->bit_transition_flag)
        {
            uint32_t offset = p->fft_size / 2;
            memcpy(buf + 2 * (size_t)offset, code, sizeof(float) * 2 * offset);
        }
    else if (p->fft_size == p->consumed_samples)
        {
            memcpy(buf, code, sizeof(float) * 2 * p->consumed_samples);
        }
    else
        {
            memcpy(buf + 2 * (size_t)(p->fft_size - p->consumed_samples), code, sizeof(float) * 2 * p->consumed_samples);
        }
    fft_c32(p->fft_codes, buf, p->fft_size, 0);
    for (uint32_t i = 0; i < p->fft_size; i++) p->fft_codes[2 * i + 1] = -p->fft_codes[2 * i + 1];
    free(buf);
}

void orc_pcps_reset_grid(orc_pcps* p)
{
    memset(p->magnitude_grid, 0, sizeof(float) * (size_t)p->fft_size * p->num_doppler_bins);
    p->dwell_counter = 0;
}

static void grid_max(const orc_pcps* p, float* peak, uint32_t* index_doppler, uint32_t* index_time)
{
    /* shared head of both statistics (:575-585, :611-621) */
    *peak = 0.0f;
    *index_doppler = 0;
    *index_time = 0;
    for (uint32_t i = 0; i < p->num_doppler_bins; i++)
        {
            const float* row = p->magnitude_grid + (size_t)i * p->fft_size;
            uint32_t t = orc_index_max(row, p->fft_size);
            if (row[t] > *peak)
                {
                    *peak = row[t];
                    *index_doppler = i;
                    *index_time = t;
                }
        }
}

void orc_pcps_core(orc_pcps* p, const float* in_samples, orc_pcps_result* out)
{
    const uint32_t N = p->fft_size;
    float* in = (float*)calloc((size_t)N * 2, sizeof(float)); /* d_input_signal, zero padded (:680-688) */
    memcpy(in, in_samples, sizeof(float) * 2 * p->consumed_samples);
    float* a = (float*)malloc(sizeof(float) * 2 * N);
    float* b = (float*)malloc(sizeof(float) * 2 * N);
    float input_power = 0.0f;
    memset(out, 0, sizeof *out);
    p->dwell_counter++;
    if (p->use_cfar || p->bit_transition_flag)
        { /* :703-709 */
            for (uint32_t i = 0; i < N; i++) p->tmp_buffer[i] = in[2 * i] * in[2 * i] + in[2 * i + 1] * in[2 * i + 1];
            for (uint32_t i = 0; i < N; i++) input_power += p->tmp_buffer[i];
            input_power /= (float)N;
        }
    const uint32_t eff = p->effective_fft_size;
    const uint32_t offset = p->bit_transition_flag ? eff : 0;
    for (uint32_t d = 0; d < p->num_doppler_bins; d++)
        { /* :714-745 */
            const float* w = p->wipeoffs + 2 * (size_t)d * N;
            for (uint32_t i = 0; i < N; i++) cmulf(&a[2 * i], &a[2 * i + 1], in[2 * i], in[2 * i + 1], w[2 * i], w[2 * i + 1]);
            fft_c32(b, a, N, 0);
            for (uint32_t i = 0; i < N; i++) cmulf(&a[2 * i], &a[2 * i + 1], b[2 * i], b[2 * i + 1], p->fft_codes[2 * i], p->fft_codes[2 * i + 1]);
            fft_c32(b, a, N, 1);
            float* row = p->magnitude_grid + (size_t)d * N;
            if (p->dwell_counter == 1)
                {
                    for (uint32_t i = 0; i < eff; i++)
                        {
                            float re = b[2 * (i + offset)], im = b[2 * (i + offset) + 1];
                            row[i] = re * re + im * im;
                        }
                }
            else
                {
                    for (uint32_t i = 0; i < eff; i++)
                        {
                            float re = b[2 * (i + offset)], im = b[2 * (i + offset) + 1];
                            p->tmp_buffer[i] = re * re + im * im;
                        }
                    for (uint32_t i = 0; i < eff; i++) row[i] = row[i] + p->tmp_buffer[i];
                }
        }
    float peak;
    uint32_t index_doppler, index_time;
    grid_max(p, &peak, &index_doppler, &index_time);
    out->indext = index_time;
    out->doppler_index = index_doppler;
    out->doppler = -p->doppler_max + p->doppler_step * (int32_t)index_doppler;
    out->mag = peak;
    out->input_power = input_power;
    if (p->use_cfar)
        { /* max_to_input_power_statistic (:565-596) */
            float nf = (float)N * (float)N;
            float magt = peak / (nf * nf);
            out->test_statistics = magt / input_power;
        }
    else
        { /* first_vs_second_peak_statistic (:599-665) */
            int32_t e1 = (int32_t)index_time - (int32_t)p->samples_per_chip;
            int32_t e2 = (int32_t)index_time + (int32_t)p->samples_per_chip;
            if (e1 < 0)
                e1 = (int32_t)N + e1;
            else if (e2 >= (int32_t)N)
                e2 = e2 - (int32_t)N;
            const float* row = p->magnitude_grid + (size_t)index_doppler * N;
            /* corrected variant: whole row */
            float* full = (float*)malloc(sizeof(float) * N);
            memcpy(full, row, sizeof(float) * N);
            int32_t idx = e1;
            do
                {
                    full[idx] = 0.0f;
                    idx++;
                    if (idx == (int32_t)N) idx = 0;
                }
            while (idx != e2);
            out->second_peak_fixed = full[orc_index_max(full, N)];
            free(full);
            /* as written: memcpy of d_fft_size BYTES (:647) */
            memcpy(p->tmp_buffer, row, N);
            idx = e1;
            do
                {
                    p->tmp_buffer[idx] = 0.0f;
                    idx++;
                    if (idx == (int32_t)N) idx = 0;
                }
            while (idx != e2);
            out->second_peak = p->tmp_buffer[orc_index_max(p->tmp_buffer, N)];
            out->test_statistics = peak / out->second_peak;
        }
    /* :764-768 */
    out->acq_delay_samples = (double)fmodf((float)index_time, p->samples_per_code);
    out->acq_doppler_hz = (double)out->doppler;
    free(in);
    free(a);
    free(b);
}
