// acq_phase_segments.h -- the float32 running phase of the Doppler wipe-off (volk_gnsssdr_s32f_sincos_32fc: `_phase += phase_inc`
// per sample, pcps_acquisition.cc:296-310) as a short list of ARITHMETIC PROGRESSIONS, bit for bit.
//
// p[0] = 0, p[i + 1] = fl(p[i] + inc) looks inherently sequential (25000 dependent additions per Doppler bin: 0.56 ms for one
// lane per bin on the GPU, whatever the number of bins).  But while p stays inside one binade [2^e, 2^(e+1)) every p[i] is a
// multiple of the binade's spacing u, so fl(p + inc) = p + rn(inc / u) u: the step is CONSTANT, except where inc / u ends in
// exactly one half -- round-to-nearest-even then makes the result an even multiple of u, and from the second such step on the
// step is constant as well.  So: take true float steps until two consecutive steps are equal and stay inside p's binade, then
// jump to the end of the binade in closed form (p[i + j] = p[i] + j d, exact in double), and repeat.  A bin needs ~100
// segments for 25000 samples; every sample's phase is then computed independently (one thread each) and equals the
// sequential sum exactly (tests/phase_segments_selftest.cpp compares with the plain loop for random increments).
// Host-only, header-only.  Built with -ffp-contract=off like the rest of the library (plain IEEE float additions).
#ifndef ACQ_PHASE_SEGMENTS_H
#define ACQ_PHASE_SEGMENTS_H
#include <cmath>
#include <cstdint>
#include <vector>

struct AcqPhaseSeg
{
    double p0;  // phase at sample i0
    double d;   // step per sample inside the segment
    int i0;     // first sample of the segment (the segment ends where the next one starts)
    int pad;
};

// appends the segments of one bin (samples 0 .. n - 1) to `out`; returns how many were appended
static inline int acq_phase_segments(float inc, int n, std::vector<AcqPhaseSeg>& out)
{
    const size_t before = out.size();
    volatile float p = 0.0f;  // volatile: every addition is rounded to float, whatever the host compiler would like to keep in registers
    int i = 0;
    while (i < n)
        {
            const float pc = p;
            volatile float p1 = pc + inc;
            volatile float p2 = p1 + inc;
            const double d1 = (double)p1 - (double)pc, d2 = (double)p2 - (double)p1;
            if ((float)p1 == pc)
                {
                    // the increment no longer changes p (inc == 0, or p has outgrown it): constant from here on
                    out.push_back(AcqPhaseSeg{(double)pc, 0.0, i, 0});
                    break;
                }
            int e0 = 0, e1 = 0, e2 = 0;
            (void)std::frexp(pc, &e0);
            (void)std::frexp((float)p1, &e1);
            (void)std::frexp((float)p2, &e2);
            if (pc != 0.0f && d1 == d2 && d1 != 0.0 && e0 == e1 && e1 == e2 && std::isfinite((float)p2))
                {
                    // |p| < 2^e0 throughout the binade; samples i .. i + steps hold p + j d1, all of magnitude <= 2^e0 (the boundary itself
                    // is representable at either spacing)
                    const double hi = std::ldexp(1.0, e0);
                    const double room = hi - std::fabs((double)pc);
                    long long steps = (long long)std::floor(room / std::fabs(d1));
                    if (steps > (long long)(n - 1 - i)) steps = n - 1 - i;
                    if (steps >= 2)
                        {
                            out.push_back(AcqPhaseSeg{(double)pc, d1, i, 0});
                            i += (int)steps;
                            p = (float)((double)pc + (double)steps * d1);  // exact: a multiple of the binade's spacing, not beyond its upper boundary
                            continue;
                        }
                }
            out.push_back(AcqPhaseSeg{(double)pc, 0.0, i, 0});  // one sample, one true float step
            i += 1;
            p = p1;
        }
    return (int)(out.size() - before);
}

// the phase of sample i from a bin's segments (what the device does per thread); segs sorted by i0, segs[0].i0 == 0
static inline float acq_phase_from_segments(const AcqPhaseSeg* segs, int n_segs, int i)
{
    int lo = 0, hi = n_segs - 1;
    while (lo < hi)
        {
            const int mid = (lo + hi + 1) >> 1;
            if (segs[mid].i0 <= i)
                lo = mid;
            else
                hi = mid - 1;
        }
    return (float)(segs[lo].p0 + (double)(i - segs[lo].i0) * segs[lo].d);
}
#endif
