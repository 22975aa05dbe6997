/*
 * ref_driver_loop.cc -- C-linkage callers for the reference's loop filters that compile from their own
 * sources (test infrastructure).  tracking_FLL_PLL_filter.cc (the carrier loop filter of dll_pll_veml_tracking,
 * dll_pll_veml_tracking.cc:344,564,939-950,1752), tracking_2nd_DLL_filter.cc and tracking_2nd_PLL_filter.cc (the
 * code / carrier filters of the GLONASS and carrier-aided blocks) include nothing but their own headers; they are
 * compiled from /root/reference where they lie (oracle/Makefile target `ref` -> _ref/libref_loop.so).  This file
 * only gives their methods C names.
 */
#include "tracking_2nd_DLL_filter.h"
#include "tracking_2nd_PLL_filter.h"
#include "tracking_FLL_PLL_filter.h"

extern "C" {
void* ref_fll_pll_new() { return new Tracking_FLL_PLL_filter(); }
void ref_fll_pll_delete(void* f) { delete static_cast<Tracking_FLL_PLL_filter*>(f); }
void ref_fll_pll_set_params(void* f, float fll_bw_hz, float pll_bw_hz, int order)
{
    static_cast<Tracking_FLL_PLL_filter*>(f)->set_params(fll_bw_hz, pll_bw_hz, order);
}
void ref_fll_pll_initialize(void* f, float doppler_hz) { static_cast<Tracking_FLL_PLL_filter*>(f)->initialize(doppler_hz); }
float ref_fll_pll_get_carrier_error(void* f, float fll, float pll, float t)
{
    return static_cast<Tracking_FLL_PLL_filter*>(f)->get_carrier_error(fll, pll, t);
}

void* ref_dll2_new(float pdi) { return new Tracking_2nd_DLL_filter(pdi); }
void ref_dll2_delete(void* f) { delete static_cast<Tracking_2nd_DLL_filter*>(f); }
void ref_dll2_set_bw(void* f, float bw) { static_cast<Tracking_2nd_DLL_filter*>(f)->set_DLL_BW(bw); }
void ref_dll2_set_pdi(void* f, float pdi) { static_cast<Tracking_2nd_DLL_filter*>(f)->set_pdi(pdi); }
void ref_dll2_initialize(void* f) { static_cast<Tracking_2nd_DLL_filter*>(f)->initialize(); }
float ref_dll2_get_code_nco(void* f, float e) { return static_cast<Tracking_2nd_DLL_filter*>(f)->get_code_nco(e); }

void* ref_pll2_new(float pdi) { return new Tracking_2nd_PLL_filter(pdi); }
void ref_pll2_delete(void* f) { delete static_cast<Tracking_2nd_PLL_filter*>(f); }
void ref_pll2_set_bw(void* f, float bw) { static_cast<Tracking_2nd_PLL_filter*>(f)->set_PLL_BW(bw); }
void ref_pll2_set_pdi(void* f, float pdi) { static_cast<Tracking_2nd_PLL_filter*>(f)->set_pdi(pdi); }
void ref_pll2_initialize(void* f) { static_cast<Tracking_2nd_PLL_filter*>(f)->initialize(); }
float ref_pll2_get_carrier_nco(void* f, float e) { return static_cast<Tracking_2nd_PLL_filter*>(f)->get_carrier_nco(e); }
}
