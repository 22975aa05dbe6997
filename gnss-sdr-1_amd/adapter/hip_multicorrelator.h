/*!
 * \file hip_multicorrelator.h
 * \brief Drop-in image of Cpu_Multicorrelator (complex chips) backed by libgnsscorr.so (MI355X / HIP).
 *
 * Same method set, argument meaning, ownership and return values as the reference class
 * (src/algorithms/tracking/libs/cpu_multicorrelator.h:46-64), which the GLONASS L1/L2 trackers
 * (glonass_l1_ca_dll_pll_tracking_cc.h, ..._c_aid_tracking_cc.h) and the GPS L1 C-Aid tracker
 * (gps_l1_ca_dll_pll_c_aid_tracking_cc.h) hold as `multicorrelator_cpu`: pointers are retained,
 * every bool method returns true when the GPU call succeeded (always, in the reference).  last_status() is the only addition (see
 * hip_multicorrelator_real_codes.h).
 */
#ifndef GNSSCORR_HIP_MULTICORRELATOR_H_
#define GNSSCORR_HIP_MULTICORRELATOR_H_

#include "hip_multicorrelator_real_codes.h"

class Hip_Multicorrelator
{
public:
    Hip_Multicorrelator() : d_corr(nullptr), d_status(GC_OK)
    {
        gc_ctx *ctx = gnsscorr::shared_context();
        d_status = ctx ? gc_correlator_create(ctx, &d_corr) : GC_ERR_NO_DEVICE;
    }

    ~Hip_Multicorrelator()
    {
        if (d_corr != nullptr) gc_correlator_destroy(d_corr);
    }

    Hip_Multicorrelator(const Hip_Multicorrelator &) = delete;
    Hip_Multicorrelator &operator=(const Hip_Multicorrelator &) = delete;

    bool init(int max_signal_length_samples, int n_correlators)
    {
        d_n_corr = n_correlators;
        return check(gc_correlator_init(d_corr, max_signal_length_samples, n_correlators));
    }

    bool set_local_code_and_taps(int code_length_chips, const std::complex<float> *local_code_in, float *shifts_chips)
    {
        return check(gc_correlator_set_local_code_and_taps_complex(d_corr, code_length_chips, reinterpret_cast<const float *>(local_code_in), shifts_chips));
    }

    bool set_input_output_vectors(std::complex<float> *corr_out, const std::complex<float> *sig_in)
    {
        d_out = corr_out;
        return check(gc_correlator_set_input_output_vectors(d_corr, reinterpret_cast<float *>(corr_out), reinterpret_cast<const float *>(sig_in)));
    }

    //! Kept for signature parity: the resampled replica is fused into the correlation kernel.
    void update_local_code(int /*correlator_length_samples*/, float /*rem_code_phase_chips*/, float /*code_phase_step_chips*/) {}

    bool Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float rem_code_phase_chips, float code_phase_step_chips, int signal_length_samples)
    {
        return zero_on_failure(check(gc_correlator_carrier_wipeoff_multicorrelator_resampler_5(d_corr, rem_carrier_phase_in_rad, phase_step_rad, rem_code_phase_chips, code_phase_step_chips, signal_length_samples)));
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
        if (s != GC_OK && d_status == GC_OK) std::fprintf(stderr, "%s: %s\n", "Hip_Multicorrelator", gc_last_error());
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

#endif /* GNSSCORR_HIP_MULTICORRELATOR_H_ */
