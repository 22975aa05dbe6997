// loop_maths_selftest.cpp -- CPU-only check of tracking_loop_maths.h against the expected values of the
// reference's own unit test (src/tests/unit-tests/signal-processing-blocks/tracking/tracking_loop_filter_test.cc:
// impulse responses of the 1st/2nd/3rd order loop filters with and without the last integrator, bw = 5 Hz,
// T = 1 ms) and against closed forms of the discriminators / lock detectors.
#include "tracking_loop_maths.h"
#include <cmath>
#include <cstdio>
#include <vector>

static int g_fail = 0;
#define EXPECT_NEAR(a, b, tol)                                                                     \
    do                                                                                             \
        {                                                                                          \
            if (!(std::fabs((double)(a) - (double)(b)) <= (tol)))                                  \
                {                                                                                  \
                    std::printf("FAIL %s:%d: %s = %.9g, expected %.9g\n", __FILE__, __LINE__, #a, (double)(a), (double)(b)); \
                    g_fail++;                                                                      \
                }                                                                                  \
        }                                                                                          \
    while (0)

static void impulse(int order, bool last_integrator, const std::vector<float>& expected, double tol)
{
    Tracking_loop_filter f(0.001f, 5.0f, order, last_integrator);
    EXPECT_NEAR(f.get_noise_bandwidth(), 5.0f, 0);
    EXPECT_NEAR(f.get_update_interval(), 0.001f, 0);
    EXPECT_NEAR(f.get_order(), order, 0);
    const std::vector<float> sample_data = {0.0, 0.0, 1.0, 0.0, 0.0, 0.0};
    f.initialize(0.0);
    for (size_t i = 0; i < sample_data.size(); ++i) EXPECT_NEAR(f.apply(sample_data[i]), expected[i], tol);
}

int main()
{
    // TrackingLoopFilterTest.* expected outputs
    impulse(1, false, {0.0f, 0.0f, 20.0f, 0.0f, 0.0f, 0.0f}, 1e-5);  // i * g1, g1 = 4 * bw
    impulse(1, true, {0.0f, 0.0f, 0.01f, 0.02f, 0.02f, 0.02f}, 1e-4);
    impulse(2, false, {0.0f, 0.0f, 13.37778f, 0.0889f, 0.0889f, 0.0889f}, 1e-4);
    impulse(2, true, {0.0f, 0.0f, 0.006689f, 0.013422f, 0.013511f, 0.013600f}, 1e-4);
    impulse(3, false, {0.0f, 0.0f, 15.31877f, 0.04494f, 0.04520f, 0.04546f}, 1e-4);
    impulse(3, true, {0.0f, 0.0f, 0.007659f, 0.015341f, 0.015386f, 0.015432f}, 1e-4);

    // discriminators
    EXPECT_NEAR(pll_cloop_two_quadrant_atan(gr_complex_t(1.0f, 1.0f)), M_PI / 4, 1e-7);
    EXPECT_NEAR(pll_cloop_two_quadrant_atan(gr_complex_t(-1.0f, -1.0f)), M_PI / 4, 1e-7);  // insensitive to 180 degree flips
    EXPECT_NEAR(pll_cloop_two_quadrant_atan(gr_complex_t(0.0f, 1.0f)), 0.0, 0);
    EXPECT_NEAR(pll_four_quadrant_atan(gr_complex_t(-1.0f, -1.0f)), -3 * M_PI / 4, 1e-6);
    EXPECT_NEAR(fll_four_quadrant_atan(gr_complex_t(1.0f, 0.0f), gr_complex_t(0.0f, 1.0f), 0.0, 0.001), (M_PI / 2) / 0.001, 1e-3);
    EXPECT_NEAR(dll_nc_e_minus_l_normalized(gr_complex_t(3.0f, 4.0f), gr_complex_t(0.0f, 5.0f)), 0.0, 1e-7);
    EXPECT_NEAR(dll_nc_e_minus_l_normalized(gr_complex_t(3.0f, 0.0f), gr_complex_t(1.0f, 0.0f)), 0.25, 1e-7);
    EXPECT_NEAR(dll_nc_e_minus_l_normalized(gr_complex_t(0.0f, 0.0f), gr_complex_t(0.0f, 0.0f)), 0.0, 0);
    EXPECT_NEAR(dll_nc_vemlp_normalized(gr_complex_t(3, 0), gr_complex_t(4, 0), gr_complex_t(0, 0), gr_complex_t(0, 0)), 1.0, 1e-7);

    // lock detectors: constant prompt on the I axis -> NBD/NBP = 1; on the Q axis -> -1
    std::vector<gr_complex_t> p(20, gr_complex_t(100.0f, 0.0f));
    EXPECT_NEAR(carrier_lock_detector(p.data(), 20), 1.0, 1e-7);
    for (auto& v : p) v = gr_complex_t(0.0f, 100.0f);
    EXPECT_NEAR(carrier_lock_detector(p.data(), 20), -1.0, 1e-7);
    // SNV estimator: |I| = A, Q = +-s alternating: Psig = A^2, Ptot = A^2 + s^2 -> C/N0 = 10log10(A^2/s^2 / T)
    for (size_t i = 0; i < p.size(); i++) p[i] = gr_complex_t(100.0f, (i % 2) ? 10.0f : -10.0f);
    EXPECT_NEAR(cn0_svn_estimator(p.data(), 20, 0.001), 10.0 * std::log10(100.0) + 30.0, 1e-3);

    // carrier loop filter: zero discriminators keep the initial Doppler
    for (int order : {2, 3})
        {
            Tracking_FLL_PLL_filter f;
            f.set_params(35.0f, 50.0f, order);
            f.initialize(1234.0f);
            for (int i = 0; i < 5; i++) EXPECT_NEAR(f.get_carrier_error(0.0f, 0.0f, 0.001f), 1234.0, 1e-3);
            // a constant phase error integrates upwards
            float a = f.get_carrier_error(0.0f, 0.1f, 0.001f), b = f.get_carrier_error(0.0f, 0.1f, 0.001f);
            EXPECT_NEAR(b > a, 1, 0);
        }
    Dll_Pll_Conf c;
    EXPECT_NEAR(c.pll_filter_order, 3, 0);
    EXPECT_NEAR(c.cn0_samples, 20, 0);
    EXPECT_NEAR(c.carrier_lock_th, 0.85, 1e-12);
    std::printf(g_fail ? "%d FAILURES\n" : "loop maths self-test passed\n", g_fail);
    return g_fail ? 1 : 0;
}
