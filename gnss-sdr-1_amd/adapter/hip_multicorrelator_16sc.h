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
        check(gc_correlator_init(d_corr, max_signal_length_samples, n_correlators));
        return true;
    }

    bool set_local_code_and_taps(int code_length_chips, const lv_16sc_t *local_code_in, float *shifts_chips)
    {
        check(gc_correlator_set_local_code_and_taps_16sc(d_corr, code_length_chips, reinterpret_cast<const int16_t *>(local_code_in), shifts_chips));
        return true;
    }

    bool set_input_output_vectors(lv_16sc_t *corr_out, const lv_16sc_t *sig_in)
    {
        check(gc_correlator_set_input_output_vectors_16sc(d_corr, reinterpret_cast<int16_t *>(corr_out), reinterpret_cast<const int16_t *>(sig_in)));
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
                if (d_status == GC_OK) std::fprintf(stderr, "Hip_Multicorrelator_16sc: %s\n", gc_last_error());
                d_status = s;
            }
    }

    gc_correlator *d_corr;
    gc_status d_status;
};

#endif /* GNSSCORR_HIP_MULTICORRELATOR_16SC_H_ */
