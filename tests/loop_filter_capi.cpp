// C names for the drop-in layer's own loop filters (gnss-sdr-1_amd/adapter/tracking_loop_maths.h: Tracking_FLL_PLL_filter;
// adapter/hip_glonass_ca_dll_pll_tracking.h: Tracking_2nd_DLL_filter / Tracking_2nd_PLL_filter), so that
// tests/test_loop_filter_pin.py can run the committed scripts of tests/golden/ref_loop_filters.npz (outputs of the REFERENCE's
// filters compiled in oracle/_ref) through them.  Host-only code, built by the test with g++.
#include "tracking_loop_maths.h"
#ifdef WITH_SECOND_ORDER
#include "hip_glonass_ca_dll_pll_tracking.h"
#endif

extern "C" {
int fp_run(int order, float fll_bw, float pll_bw, float narrow_bw, float doppler, int n_switch, int n, const float* fll, const float* pll,
    const float* T, float* out)
{
    Tracking_FLL_PLL_filter f;
    f.set_params(fll_bw, pll_bw, order);
    f.initialize(doppler);
    for (int k = 0; k < n; k++)
        {
            if (k == n_switch) f.set_params(fll_bw, narrow_bw, order);
            out[k] = f.get_carrier_error(fll[k], pll[k], T[k]);
        }
    return 0;
}
#ifdef WITH_SECOND_ORDER
int s2_run(int which, float bw, float pdi, float bw2, float pdi2, int n, const float* e, float* out)
{
    if (which == 0)
        {
            Tracking_2nd_DLL_filter f(pdi);
            f.set_DLL_BW(bw);
            f.initialize();
            for (int k = 0; k < n; k++)
                {
                    if (k == n / 2)
                        {
                            f.set_pdi(pdi2);
                            f.set_DLL_BW(bw2);
                        }
                    out[k] = f.get_code_nco(e[k]);
                }
        }
    else
        {
            Tracking_2nd_PLL_filter f(pdi);
            f.set_PLL_BW(bw);
            f.initialize();
            for (int k = 0; k < n; k++)
                {
                    if (k == n / 2)
                        {
                            f.set_pdi(pdi2);
                            f.set_PLL_BW(bw2);
                        }
                    out[k] = f.get_carrier_nco(e[k]);
                }
        }
    return 0;
}
#endif
}
