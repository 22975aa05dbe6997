// phase_segments_selftest.cpp -- csrc/acq_phase_segments.h against the plain sequential float32 running sum
// (volk_gnsssdr_s32f_sincos_32fc_generic's `_phase += phase_inc`), bit for bit, for the increments the acquisition uses
// (2 pi f / fs, Doppler grids with and without an FDMA / IF offset) and random ones incl. tie-prone values.  CPU only.
#include "acq_phase_segments.h"
#include <cstdio>
#include <cstring>
#include <random>

static int check(float inc, int n, int* max_segs)
{
    std::vector<AcqPhaseSeg> segs;
    const int ns = acq_phase_segments(inc, n, segs);
    if (ns > *max_segs) *max_segs = ns;
    if (segs[0].i0 != 0) return 1;
    volatile float p = 0.0f;
    for (int i = 0; i < n; i++)
        {
            const float want = p;
            const float got = acq_phase_from_segments(segs.data(), ns, i);
            if (std::memcmp(&want, &got, 4) != 0 && !(want == 0.0f && got == 0.0f))
                {
                    std::printf("FAIL inc %.9g sample %d: sequential %.9g, segments %.9g (%d segments)\n", inc, i, want, got, ns);
                    return 1;
                }
            p = p + inc;
        }
    return 0;
}

int main()
{
    int fails = 0, max_segs = 0, cases = 0;
    // the bench / test grids: fs 25 MHz, 6.625 MHz, 4 MHz, 8 MHz; Doppler -10 .. 10 kHz in 250 Hz; offsets 0, GLONASS channels, 1.25 MHz
    const double fss[] = {25e6, 6.625e6, 4e6, 8e6};
    const long long offs[] = {0, -4 * 562500LL, 3 * 562500LL, 6 * 562500LL, 1250000, -437500 * 7LL};
    for (double fs : fss)
        for (long long off : offs)
            for (int dop = -10000; dop <= 10000; dop += 250)
                {
                    const float freq = (float)(off + dop);
                    const float inc = -(float)(6.283185307179586 * freq / (float)fs);
                    fails += check(inc, 25000, &max_segs);
                    cases++;
                }
    const int max_grid = max_segs;
    // long blocks, tiny and huge increments, exact ties (increments with few mantissa bits), random mantissas and signs
    const float special[] = {0.0f, 1.0f, -1.0f, 0.5f, 1.5f, 3.0f, 1e-20f, -1e-20f, 1e-30f, 6.2831855f, 0.75f, 2.5f, 1.25e-3f, 3.0517578125e-5f, 1.00000012f, 1e10f};
    for (float inc : special)
        {
            fails += check(inc, 100000, &max_segs);
            cases++;
        }
    std::mt19937 rng(20261005);
    for (int k = 0; k < 3000; k++)
        {
            const int e = (int)(rng() % 40) - 30;
            uint32_t mant = rng() & 0x7fffff;
            if (k % 3 == 0) mant &= 0x7ff000;  // few mantissa bits: ties in many binades
            if (k % 7 == 0) mant = (mant & 0x7fff00) | 0x80;
            const float inc = std::ldexp(1.0f + (float)mant / 8388608.0f, e) * ((rng() & 1) ? 1.0f : -1.0f);
            fails += check(inc, 20000 + (int)(rng() % 20000), &max_segs);
            cases++;
        }
    std::printf(fails ? "%d FAILURES\n" : "phase segments: %d increments agree with the sequential float32 sum bit for bit (at most %d segments per row on the acquisition grids, %d overall)\n", fails ? fails : cases, max_grid, max_segs);
    return fails ? 1 : 0;
}
