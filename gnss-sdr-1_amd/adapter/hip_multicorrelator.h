/*!
 * \file hip_multicorrelator.h
 * \brief Drop-in image of Cpu_Multicorrelator (complex chips) backed by libgnsscorr.so (MI355X / HIP).
 *
 * Same method set, argument meaning, ownership and return values as the reference class
 * (src/algorithms/tracking/libs/cpu_multicorrelator.h:46-64), which the GLONASS L1/L2 trackers
 * (glonass_l1_ca_dll_pll_tracking_cc.h, ..._c_aid_tracking_cc.h) and the GPS L1 C-Aid tracker
 * (gps_l1_ca_dll_pll_c_aid_tracking_cc.h) hold as `multicorrelator_cpu`: pointers are retained,
 * every bool method returns true.  last_status() is the only addition (see
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
        check(gc_correlator_init(d_corr, max_signal_length_samples, n_correlators));
        return true;
    }

    bool set_local_code_and_taps(int code_length_chips, const std::complex<float> *local_code_in, float *shifts_chips)
    {
        check(gc_correlator_set_local_code_and_taps_complex(d_corr, code_length_chips, reinterpret_cast<const float *>(local_code_in), shifts_chips));
        return true;
    }

    bool set_input_output_vectors(std::complex<float> *corr_out, const std::complex<float> *sig_in)
    {
        check(gc_correlator_set_input_output_vectors(d_corr, reinterpret_cast<float *>(corr_out), reinterpret_cast<const float *>(sig_in)));
        return true;
    }

    //! Kept for signature parity: the resampled replica is fused into the correlation kernel.
    void update_local_code(int /*correlator_length_samples*/, float /*rem_code_phase_chips*/, float /*code_phase_step_chips*/) {}

    bool Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float rem_code_phase_chips, float code_phase_step_chips, int signal_length_samples)
    {
        check(gc_correlator_carrier_wipeoff_multicorrelator_resampler_5(d_corr, rem_carrier_phase_in_rad, phase_step_rad, rem_code_phase_chips, code_phase_step_chips, signal_length_samples));
        return true;
    }

    bool free()
    {
        if (d_corr != nullptr) check(gc_correlator_free(d_corr));
        return true;
    }

    //! GC_OK, or the status of the last failed call (see gc_last_error()).
    gc_status last_status() const { return d_status; }

private:
    void check(gc_status s)
    {
        if (d_corr == nullptr) return;
        if (s != GC_OK)
            {
                if (d_status == GC_OK) std::fprintf(stderr, "Hip_Multicorrelator: %s\n", gc_last_error());
                d_status = s;
            }
    }

    gc_correlator *d_corr;
    gc_status d_status;
};

#endif /* GNSSCORR_HIP_MULTICORRELATOR_H_ */
