/*!
 * \file tracking_loop_maths.h
 * \brief Post-correlation scalar maths of the DLL/PLL tracking loop (host side; a few dozen
 * flops per millisecond per channel), with the reference's names and formulas:
 *   discriminators   src/algorithms/tracking/libs/tracking_discriminators.cc:41-128
 *   lock detectors   src/algorithms/tracking/libs/lock_detectors.cc:71-111
 *   Tracking_loop_filter      src/algorithms/tracking/libs/tracking_loop_filter.cc:74-245
 *   Tracking_FLL_PLL_filter   src/algorithms/tracking/libs/tracking_FLL_PLL_filter.cc:55-133
 *   Dll_Pll_Conf              src/algorithms/tracking/libs/dll_pll_conf.{h,cc}
 * Inside a GNSS-SDR build (GNSSCORR_WITH_GNSS_SDR) the tree's own headers are used instead.
 */
#ifndef GNSSCORR_TRACKING_LOOP_MATHS_H_
#define GNSSCORR_TRACKING_LOOP_MATHS_H_

#ifdef GNSSCORR_WITH_GNSS_SDR
#include "dll_pll_conf.h"
#include "lock_detectors.h"
#include "tracking_FLL_PLL_filter.h"
#include "tracking_discriminators.h"
#include "tracking_loop_filter.h"
#else

#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

class Dll_Pll_Conf
{
public:
    // defaults of Dll_Pll_Conf::Dll_Pll_Conf() (dll_pll_conf.cc:36-73)
    int fll_filter_order = 1;
    bool enable_fll_pull_in = false;
    bool enable_fll_steady_state = false;
    unsigned int pull_in_time_s = 2;
    int pll_filter_order = 3;
    int dll_filter_order = 2;
    double fs_in = 0.0;
    uint32_t vector_length = 0U;
    bool dump = false;
    bool dump_mat = true;
    std::string dump_filename = "./dll_pll_dump.dat";
    float pll_pull_in_bw_hz = 50.0f;
    float dll_pull_in_bw_hz = 3.0f;
    float fll_bw_hz = 35.0f;
    float pll_bw_hz = 35.0f;
    float dll_bw_hz = 2.0f;
    float pll_bw_narrow_hz = 5.0f;
    float dll_bw_narrow_hz = 0.75f;
    float early_late_space_chips = 0.5f;
    float very_early_late_space_chips = 0.5f;
    float early_late_space_narrow_chips = 0.1f;
    float very_early_late_space_narrow_chips = 0.1f;
    int32_t extend_correlation_symbols = 5;
    bool high_dyn = false;
    int32_t cn0_samples = 20;
    int32_t carrier_lock_det_mav_samples = 20;
    int32_t cn0_min = 25;
    int32_t max_lock_fail = 50;
    uint32_t smoother_length = 10;
    double carrier_lock_th = 0.85;
    bool track_pilot = false;
    char system = 'G';
    char signal[3] = {'1', 'C', 0};
};

using gr_complex_t = std::complex<float>;

// ---- discriminators (all outputs in radians / chips as in the reference) ----
inline double fll_four_quadrant_atan(gr_complex_t prompt_s1, gr_complex_t prompt_s2, double t1, double t2)
{
    const double dot = prompt_s1.real() * prompt_s2.real() + prompt_s1.imag() * prompt_s2.imag();
    const double cross = prompt_s1.real() * prompt_s2.imag() - prompt_s2.real() * prompt_s1.imag();
    return std::atan2(cross, dot) / (t2 - t1);
}

inline double pll_four_quadrant_atan(gr_complex_t prompt_s1)
{
    return static_cast<double>(std::atan2(prompt_s1.imag(), prompt_s1.real()));
}

inline double pll_cloop_two_quadrant_atan(gr_complex_t prompt_s1)
{
    if (prompt_s1.real() != 0.0) return static_cast<double>(std::atan(prompt_s1.imag() / prompt_s1.real()));
    return 0.0;
}

inline double dll_nc_e_minus_l_normalized(gr_complex_t early_s1, gr_complex_t late_s1)
{
    const double e = std::abs(early_s1), l = std::abs(late_s1);
    if (e + l == 0.0) return 0.0;
    return 0.5 * (e - l) / (e + l);
}

inline double dll_nc_vemlp_normalized(gr_complex_t very_early_s1, gr_complex_t early_s1, gr_complex_t late_s1, gr_complex_t very_late_s1)
{
    const double e = std::sqrt(std::norm(very_early_s1) + std::norm(early_s1));
    const double l = std::sqrt(std::norm(very_late_s1) + std::norm(late_s1));
    if (e + l == 0.0) return 0.0;
    return (e - l) / (e + l);
}

// ---- lock detectors ----
//! Signal-to-noise variance C/N0 estimator (lock_detectors.cc:71-90)
inline float cn0_svn_estimator(const gr_complex_t* Prompt_buffer, int length, double coh_integration_time_s)
{
    double Psig = 0.0, Ptot = 0.0;
    for (int i = 0; i < length; i++)
        {
            Psig += std::abs(static_cast<double>(Prompt_buffer[i].real()));
            Ptot += static_cast<double>(Prompt_buffer[i].imag()) * static_cast<double>(Prompt_buffer[i].imag()) + static_cast<double>(Prompt_buffer[i].real()) * static_cast<double>(Prompt_buffer[i].real());
        }
    Psig /= static_cast<double>(length);
    Psig = Psig * Psig;
    Ptot /= static_cast<double>(length);
    const double SNR = Psig / (Ptot - Psig);
    return static_cast<float>(10.0 * std::log10(SNR) - 10.0 * std::log10(coh_integration_time_s));
}

//! Narrow-band difference over narrow-band power (lock_detectors.cc:94-111)
inline float carrier_lock_detector(const gr_complex_t* Prompt_buffer, int length)
{
    float sum_I = 0.0f, sum_Q = 0.0f;
    for (int i = 0; i < length; i++)
        {
            sum_I += Prompt_buffer[i].real();
            sum_Q += Prompt_buffer[i].imag();
        }
    const float NBP = sum_I * sum_I + sum_Q * sum_Q;
    const float NBD = sum_I * sum_I - sum_Q * sum_Q;
    return NBD / NBP;
}

// ---- code loop filter: bilinear-transform IIR of order 1..3 ----
class Tracking_loop_filter
{
public:
    Tracking_loop_filter() { update_coefficients(); }
    Tracking_loop_filter(float update_interval, float noise_bandwidth, int loop_order = 2, bool include_last_integrator = false)
        : d_loop_order(loop_order), d_include_last_integrator(include_last_integrator), d_noise_bandwidth(noise_bandwidth), d_update_interval(update_interval)
    {
        update_coefficients();
    }

    float get_noise_bandwidth() const { return d_noise_bandwidth; }
    float get_update_interval() const { return d_update_interval; }
    bool get_include_last_integrator() const { return d_include_last_integrator; }
    int get_order() const { return d_loop_order; }

    void set_noise_bandwidth(float noise_bandwidth)
    {
        d_noise_bandwidth = noise_bandwidth;
        update_coefficients();
    }
    void set_update_interval(float update_interval)
    {
        d_update_interval = update_interval;
        update_coefficients();
    }
    void set_include_last_integrator(bool include_last_integrator)
    {
        d_include_last_integrator = include_last_integrator;
        update_coefficients();
    }
    void set_order(int loop_order)
    {
        if (loop_order < 1 || loop_order > 3) return;
        d_loop_order = loop_order;
        update_coefficients();
    }

    void initialize(float initial_output = 0.0)
    {
        for (int i = 0; i < HISTORY; i++)
            {
                d_inputs[i] = 0.0f;
                d_outputs[i] = initial_output;
            }
        d_current_index = HISTORY - 1;
    }

    //! y[n] = sum_i b_i x[n-i] + sum_i a_i y[n-1-i] on circular histories (tracking_loop_filter.cc:74-101)
    float apply(float current_input)
    {
        float result = 0.0f;
        for (size_t i = 0; i < d_output_coefficients.size(); ++i) result += d_output_coefficients[i] * d_outputs[(d_current_index + i) % HISTORY];
        d_current_index--;
        if (d_current_index < 0) d_current_index += HISTORY;
        d_inputs[d_current_index] = current_input;
        for (size_t i = 0; i < d_input_coefficients.size(); ++i) result += d_input_coefficients[i] * d_inputs[(d_current_index + i) % HISTORY];
        d_outputs[d_current_index] = result;
        return result;
    }

private:
    static const int HISTORY = 4;

    //! natural frequency from the noise bandwidth, then the bilinear transform of the analog loop
    //! (tracking_loop_filter.cc:104-245; Kaplan's a3 = 1.1, b3 = 2.4 for the third order)
    void update_coefficients()
    {
        const float T = d_update_interval;
        const float zeta = 1.0 / std::sqrt(2.0);
        float g1, g2, g3, wn;
        auto& b = d_input_coefficients;
        auto& a = d_output_coefficients;
        switch (d_loop_order)
            {
            case 1:
                wn = d_noise_bandwidth * 4.0;
                g1 = wn;
                if (d_include_last_integrator)
                    {
                        b = {static_cast<float>(g1 * T / 2.0), static_cast<float>(g1 * T / 2.0)};
                        a = {1.0f};
                    }
                else
                    {
                        b = {g1};
                        a = {};
                    }
                break;
            case 2:
                wn = d_noise_bandwidth * (8.0 * zeta) / (4.0 * zeta * zeta + 1.0);
                g1 = wn * wn;
                g2 = wn * 2.0 * zeta;
                if (d_include_last_integrator)
                    {
                        b = {static_cast<float>(T / 2.0 * (g1 * T / 2.0 + g2)), static_cast<float>(T * T / 2.0 * g1), static_cast<float>(T / 2.0 * (g1 * T / 2.0 - g2))};
                        a = {2.0f, -1.0f};
                    }
                else
                    {
                        b = {static_cast<float>(g1 * T / 2.0 + g2), static_cast<float>(g1 * T / 2.0 - g2)};
                        a = {1.0f};
                    }
                break;
            case 3:
                {
                    wn = d_noise_bandwidth / 0.7845;
                    const float a3 = 1.1, b3 = 2.4;
                    g1 = wn * wn * wn;
                    g2 = a3 * wn * wn;
                    g3 = b3 * wn;
                    if (d_include_last_integrator)
                        {
                            b = {static_cast<float>(T / 2.0 * (g3 + T / 2.0 * (g2 + T / 2.0 * g1))), static_cast<float>(T / 2.0 * (-g3 + T / 2.0 * (g2 + 3.0 * T / 2.0 * g1))),
                                static_cast<float>(T / 2.0 * (-g3 - T / 2.0 * (g2 - 3.0 * T / 2.0 * g1))), static_cast<float>(T / 2.0 * (g3 - T / 2.0 * (g2 - T / 2.0 * g1)))};
                            a = {3.0f, -3.0f, 1.0f};
                        }
                    else
                        {
                            b = {static_cast<float>(g3 + T / 2.0 * (g2 + T / 2.0 * g1)), static_cast<float>(g1 * T * T / 2.0 - 2.0 * g3), static_cast<float>(g3 + T / 2.0 * (-g2 + T / 2.0 * g1))};
                            a = {2.0f, -1.0f};
                        }
                    break;
                }
            default:
                break;
            }
    }

    int d_loop_order = 2;
    int d_current_index = 0;
    bool d_include_last_integrator = false;
    float d_noise_bandwidth = 15.0f;
    float d_update_interval = 0.001f;
    float d_inputs[HISTORY] = {0, 0, 0, 0};
    float d_outputs[HISTORY] = {0, 0, 0, 0};
    std::vector<float> d_input_coefficients;
    std::vector<float> d_output_coefficients;
};

// ---- carrier loop filter: PLL of order 2/3 with FLL assist of order 1/2 (Kaplan 2nd ed., fig. 5.x) ----
class Tracking_FLL_PLL_filter
{
public:
    void set_params(float fll_bw_hz, float pll_bw_hz, int order)
    {
        d_order = order;
        if (d_order == 3)
            {
                d_pll_b3 = 2.400f;
                d_pll_a3 = 1.100f;
                d_pll_a2 = 1.414f;
                d_pll_w0p = pll_bw_hz / 0.7845;
                d_pll_w0p2 = d_pll_w0p * d_pll_w0p;
                d_pll_w0p3 = d_pll_w0p2 * d_pll_w0p;
                d_pll_w0f = fll_bw_hz / 0.53;
                d_pll_w0f2 = d_pll_w0f * d_pll_w0f;
            }
        else
            {
                d_pll_a2 = 1.414f;
                d_pll_w0p = pll_bw_hz / 0.53;
                d_pll_w0p2 = d_pll_w0p * d_pll_w0p;
                d_pll_w0f = fll_bw_hz / 0.25;
            }
    }

    void initialize(float d_acq_carrier_doppler_hz)
    {
        if (d_order == 3)
            {
                d_pll_x = 2.0 * d_acq_carrier_doppler_hz;
                d_pll_w = 0;
            }
        else
            {
                d_pll_w = d_acq_carrier_doppler_hz;
                d_pll_x = 0;
            }
    }

    float get_carrier_error(float FLL_discriminator, float PLL_discriminator, float correlation_time_s)
    {
        float carrier_error_hz;
        if (d_order == 3)
            {
                d_pll_w = d_pll_w + correlation_time_s * (d_pll_w0p3 * PLL_discriminator + d_pll_w0f2 * FLL_discriminator);
                d_pll_x = d_pll_x + correlation_time_s * (0.5 * d_pll_w + d_pll_a2 * d_pll_w0f * FLL_discriminator + d_pll_a3 * d_pll_w0p2 * PLL_discriminator);
                carrier_error_hz = 0.5 * d_pll_x + d_pll_b3 * d_pll_w0p * PLL_discriminator;
            }
        else
            {
                const float pll_w_new = d_pll_w + PLL_discriminator * d_pll_w0p2 * correlation_time_s + FLL_discriminator * d_pll_w0f * correlation_time_s;
                carrier_error_hz = 0.5 * (pll_w_new + d_pll_w) + d_pll_a2 * d_pll_w0p * PLL_discriminator;
                d_pll_w = pll_w_new;
            }
        return carrier_error_hz;
    }

private:
    int d_order = 0;
    float d_pll_w = 0, d_pll_w0p3 = 0, d_pll_w0f2 = 0, d_pll_x = 0, d_pll_a2 = 0, d_pll_w0f = 0, d_pll_a3 = 0, d_pll_w0p2 = 0, d_pll_b3 = 0, d_pll_w0p = 0;
};

#endif  // GNSSCORR_WITH_GNSS_SDR
#endif  // GNSSCORR_TRACKING_LOOP_MATHS_H_
