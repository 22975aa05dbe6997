/*!
 * \file hip_multicorrelator_16sc.h
 * \brief Drop-in image of Cpu_Multicorrelator_16sc (lv_16sc_t chips, input and output) backed by libgnsscorr.so.
 *
 * Same method set, argument meaning, ownership and return values as the reference class
 * (src/algorithms/tracking/libs/cpu_multicorrelator_16sc.h:44-67), held as `multicorrelator_cpu_16sc` by the
 * *_sc C-Aid trackers (gps_l1_ca_dll_pll_c_aid_tracking_sc.h, glonass_l1_ca_dll_pll_c_aid_tracking_sc.h, ...).
 * Arithmetic and its one difference from the reference's running saturating sum: see include/gnsscorr.h
 * (gc_correlator_set_local_code_and_taps_16sc) and DESIGN.md.
 */
#ifndef GNSSCORR_HIP_MULTICORRELATOR_16SC_H_
#define GNSSCORR_HIP_MULTICORRELATOR_16SC_H_

#include "hip_multicorrelator_real_codes.h"
#include <cstdint>

class Hip_Multicorrelator_16sc
{
public:
    typedef std::complex<int16_t> lv_16sc_t;  // volk_gnsssdr_complex.h:52

    Hip_Multicorrelator_16sc() : d_corr(nullptr), d_status(GC_OK)
    {
        gc_ctx *ctx = gnsscorr::shared_context();
        d_status = ctx ? gc_correlator_create(ctx, &d_corr) : GC_ERR_NO_DEVICE;
    }

    ~Hip_Multicorrelator_16sc()
    {
        if (d_corr != nullptr) gc_correlator_destroy(d_corr);
    }

    Hip_Multicorrelator_16sc(const Hip_Multicorrelator_16sc &) = delete;
    Hip_Multicorrelator_16sc &operator=(const Hip_Multicorrelator_16sc &) = delete;

    bool init(int max_signal_length_samples, int n_correlators)
    {
        d_n_corr = n_correlators;
        return check(gc_correlator_init(d_corr, max_signal_length_samples, n_correlators));
    }

    bool set_local_code_and_taps(int code_length_chips, const lv_16sc_t *local_code_in, float *shifts_chips)
    {
        return check(gc_correlator_set_local_code_and_taps_16sc(d_corr, code_length_chips, reinterpret_cast<const int16_t *>(local_code_in), shifts_chips));
    }

    bool set_input_output_vectors(lv_16sc_t *corr_out, const lv_16sc_t *sig_in)
    {
        d_out = corr_out;
        return check(gc_correlator_set_input_output_vectors_16sc(d_corr, reinterpret_cast<int16_t *>(corr_out), reinterpret_cast<const int16_t *>(sig_in)));
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
        if (s != GC_OK && d_status == GC_OK) std::fprintf(stderr, "%s: %s\n", "Hip_Multicorrelator_16sc", gc_last_error());
        d_status = s;
        return s == GC_OK;
    }

    //! a failed correlation must not leave the previous epoch's values for the loop to track on
    bool zero_on_failure(bool ok)
    {
        if (!ok && d_out != nullptr)
            for (int t = 0; t < d_n_corr; t++) d_out[t] = lv_16sc_t(0, 0);
        return ok;
    }

    gc_correlator *d_corr;
    gc_status d_status;
    lv_16sc_t *d_out = nullptr;
    int d_n_corr = 0;
};

#endif /* GNSSCORR_HIP_MULTICORRELATOR_16SC_H_ */
