/*!
 * \file hip_multicorrelator_real_codes.h
 * \brief Drop-in image of Cpu_Multicorrelator_Real_Codes backed by libgnsscorr.so (MI355X / HIP).
 *
 * Same method set, argument meaning, ownership and return values as the reference class
 * (src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.h:45-69): pointers are
 * retained, every bool method returns true when the GPU call succeeded (always, in the reference), the constructor leaves the high-dynamics
 * flag set (cpu_multicorrelator_real_codes.cc:49).  The only behavioural addition is
 * last_status(): the reference has no failure path, the GPU has (no device, HIP error);
 * a failed call returns false, ZEROES corr_out (the loop then sees no signal rather than the previous epoch's sums), logs to
 * stderr once and is visible there.
 *
 * In dll_pll_veml_tracking.h the change is one line:
 *     Cpu_Multicorrelator_Real_Codes multicorrelator_cpu;   ->   Hip_Multicorrelator_Real_Codes multicorrelator_cpu;
 */
#ifndef GNSSCORR_HIP_MULTICORRELATOR_REAL_CODES_H_
#define GNSSCORR_HIP_MULTICORRELATOR_REAL_CODES_H_

#include "gnsscorr.h"
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace gnsscorr
{
//! Process-wide context per device (created on first use).  $GNSSCORR_DEVICE selects the GPU (default 0).
inline gc_ctx *shared_context()
{
    static std::mutex mtx;
    static gc_ctx *ctx = nullptr;
    std::lock_guard<std::mutex> lk(mtx);
    if (ctx == nullptr)
        {
            int dev = 0;
            if (const char *e = std::getenv("GNSSCORR_DEVICE")) dev = std::atoi(e);
            if (gc_ctx_create(dev, &ctx) != GC_OK)
                {
                    std::fprintf(stderr, "gnsscorr: %s\n", gc_last_error());
                    ctx = nullptr;
                }
        }
    return ctx;
}
}  // namespace gnsscorr


class Hip_Multicorrelator_Real_Codes
{
public:
    Hip_Multicorrelator_Real_Codes() : d_corr(nullptr), d_status(GC_OK)
    {
        gc_ctx *ctx = gnsscorr::shared_context();
        d_status = ctx ? gc_correlator_create(ctx, &d_corr) : GC_ERR_NO_DEVICE;
    }

    ~Hip_Multicorrelator_Real_Codes()
    {
        if (d_corr != nullptr) gc_correlator_destroy(d_corr);
    }

    Hip_Multicorrelator_Real_Codes(const Hip_Multicorrelator_Real_Codes &) = delete;
    Hip_Multicorrelator_Real_Codes &operator=(const Hip_Multicorrelator_Real_Codes &) = delete;

    void set_high_dynamics_resampler(bool use_high_dynamics_resampler)
    {
        check(gc_correlator_set_high_dynamics_resampler(d_corr, use_high_dynamics_resampler ? 1 : 0));
    }

    bool init(int max_signal_length_samples, int n_correlators)
    {
        d_n_corr = n_correlators;
        return check(gc_correlator_init(d_corr, max_signal_length_samples, n_correlators));
    }

    bool set_local_code_and_taps(int code_length_chips, const float *local_code_in, float *shifts_chips)
    {
        return check(gc_correlator_set_local_code_and_taps(d_corr, code_length_chips, local_code_in, shifts_chips));
    }

    bool set_input_output_vectors(std::complex<float> *corr_out, const std::complex<float> *sig_in)
    {
        d_out = corr_out;
        return check(gc_correlator_set_input_output_vectors(d_corr, reinterpret_cast<float *>(corr_out), reinterpret_cast<const float *>(sig_in)));
    }

    //! The reference exposes this helper publicly; on the GPU the resampled replica is never
    //! materialised (it is fused into the correlation kernel), so this is a no-op kept for signature parity.
    void update_local_code(int /*correlator_length_samples*/, float /*rem_code_phase_chips*/, float /*code_phase_step_chips*/, float /*code_phase_rate_step_chips*/ = 0.0) {}

    bool Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float phase_rate_step_rad, float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips, int signal_length_samples)
    {
        return zero_on_failure(check(gc_correlator_carrier_wipeoff_multicorrelator_resampler(d_corr, rem_carrier_phase_in_rad, phase_step_rad, phase_rate_step_rad, rem_code_phase_chips, code_phase_step_chips, code_phase_rate_step_chips, signal_length_samples)));
    }

    bool Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips, int signal_length_samples)
    {
        return zero_on_failure(check(gc_correlator_carrier_wipeoff_multicorrelator_resampler_6(d_corr, rem_carrier_phase_in_rad, phase_step_rad, rem_code_phase_chips, code_phase_step_chips, code_phase_rate_step_chips, signal_length_samples)));
    }

    bool free()
    {
        return d_corr != nullptr ? check(gc_correlator_free(d_corr)) : false;
    }

    //! GC_OK, or the status of the last failed call (see gc_last_error()).
    gc_status last_status() const { return d_status; }

private:
    //! The reference's methods cannot fail and always return true; the GPU's can (no device, HIP error, bad state): the
    //! outcome of THIS call is returned and kept in last_status() -- not sticky, a later good call reads GC_OK again.
    bool check(gc_status s)
    {
        if (d_corr == nullptr) return false;  // construction failed: d_status keeps GC_ERR_NO_DEVICE
        if (s != GC_OK && d_status == GC_OK) std::fprintf(stderr, "%s: %s\n", "Hip_Multicorrelator_Real_Codes", gc_last_error());
        d_status = s;
        return s == GC_OK;
    }

    //! a failed correlation must not leave the previous epoch's values for the loop to track on
    bool zero_on_failure(bool ok)
    {
        if (!ok && d_out != nullptr)
            for (int t = 0; t < d_n_corr; t++) d_out[t] = std::complex<float>(0, 0);
        return ok;
    }

    gc_correlator *d_corr;
    gc_status d_status;
    std::complex<float> *d_out = nullptr;
    int d_n_corr = 0;
};

#endif /* GNSSCORR_HIP_MULTICORRELATOR_REAL_CODES_H_ */
