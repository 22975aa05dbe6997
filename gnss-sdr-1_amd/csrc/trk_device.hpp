// trk_device.hpp -- device-side code of the tracking multicorrelator (shared by the batched open-loop
// kernel, trk_kernels.hip, and the closed-loop kernel, trk_closed_loop.hip).  See trk_kernels.hip for
// the reference citations and the design notes.
#ifndef TRK_DEVICE_HPP
#define TRK_DEVICE_HPP
#include "gnsscorr.h"
#include "trk_kernels.h"
#include <hip/hip_runtime.h>
#include <type_traits>

#ifndef TRK_THREADS
#define TRK_THREADS 256  // workgroup size of the open-loop kernel (512 / 1024 measured: no faster)
#endif
#define TRK_CHUNK (2 * TRK_THREADS)  // samples per workgroup iteration (2 per lane)
#define TRK_HDR_FLOATS (TRK_THREADS / 64 * 16)
// LDS header of a workgroup of `threads` threads: one (re, im) partial per wave and tap (GC_MAX_TAPS taps)
static constexpr __host__ __device__ int trk_hdr_floats(int threads) { return threads / 64 * 16; }
#define TRK_RESYNC 64  // iterations between exact re-evaluations of the carrier phase
#ifndef TRK_PF
#define TRK_PF 2  // chunks prefetched ahead of the one being processed (16-byte loads in flight per lane)
#endif
#ifndef TRK_NT
#define TRK_NT 1  // nontemporal IQ loads (0: plain): the stream is read once; with line-aligned chunks +5-7 % on the HBM-bound launch (alone: +1.5 %)
#endif
#ifndef TRK_PRELOAD
#define TRK_PRELOAD 0  // the first two chunks of a window are requested before the workgroup's prologue (0: at the start of the loop)
#endif
#ifndef TRK_ALIGN_PAIRS
#define TRK_ALIGN_PAIRS 8  // alignment, in sample pairs, of the address the chunks of a window are counted from (1 or 8)
#endif

static __device__ __forceinline__ int posmod(int i, int L)
{
    int r = i % L;
    return r < 0 ? r + L : r;
}

// chip index before wrapping, generic resampler order: ((step*n) + shift) - rem
static __device__ __forceinline__ int chip_index(float step, float nf, float shift, float rem)
{
    float a = step * nf;
    float b = a + shift;
    float c = b - rem;
    return (int)floorf(c);
}

// high-dynamics first tap: (((step*n) + rate*(float)(n*n)) + shift0) - rem, n*n in uint32
static __device__ __forceinline__ int chip_index_hd(float step, float rate, unsigned n, float shift0, float rem)
{
    float a = step * (float)n;
    float r = rate * (float)(n * n);
    float b = a + r;
    float c = b + shift0;
    float d = c - rem;
    return (int)floorf(d);
}

// sum over the 64 lanes of a wave with DPP row shifts / broadcasts (no LDS traffic);
// the total ends in lane 63
static __device__ __forceinline__ float wave_sum(float v)
{
#define GC_DPP_ADD(ctrl, row_mask, bank_mask)                                                                    \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, bank_mask, false))
    GC_DPP_ADD(0x111, 0xf, 0xf);  // row_shr:1
    GC_DPP_ADD(0x112, 0xf, 0xf);  // row_shr:2
    GC_DPP_ADD(0x114, 0xf, 0xe);  // row_shr:4
    GC_DPP_ADD(0x118, 0xf, 0xc);  // row_shr:8  -> lane 15 of every row holds the row sum
    GC_DPP_ADD(0x142, 0xa, 0xf);  // row_bcast:15
    GC_DPP_ADD(0x143, 0xc, 0xf);  // row_bcast:31 -> lane 63 holds the wave sum
#undef GC_DPP_ADD
    return v;
}

// the same reduction on 32-bit integers (exact, order-independent)
static __device__ __forceinline__ int wave_sum_i(int v)
{
#define GC_DPP_ADDI(ctrl, row_mask, bank_mask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, row_mask, bank_mask, false)
    GC_DPP_ADDI(0x111, 0xf, 0xf);
    GC_DPP_ADDI(0x112, 0xf, 0xf);
    GC_DPP_ADDI(0x114, 0xf, 0xe);
    GC_DPP_ADDI(0x118, 0xf, 0xc);
    GC_DPP_ADDI(0x142, 0xa, 0xf);
    GC_DPP_ADDI(0x143, 0xc, 0xf);
#undef GC_DPP_ADDI
    return v;
}

// exact carrier rotator for sample n: exp(j*(theta0 + n*dtheta [+ e(n)*drate]))
template <bool HDC>
static __device__ __forceinline__ void carrier_at(int n, double theta0, double dtheta, double drate,
    float& zr, float& zi)
{
    double th = fma((double)n, dtheta, theta0);
    if (HDC)
        {
            // the reference applies cpowf(rate, (n-1)^2) to sample n (n >= 1), with
            // the square taken in unsigned 32-bit arithmetic and converted to float
            unsigned m = (n > 0) ? (unsigned)(n - 1) : 0u;
            float e = (float)(m * m);
            th = fma((double)e, drate, th);
        }
    const double inv2pi = 0.15915494309189533577;
    const double twopi = 6.283185307179586477;
    double t = th * inv2pi;
    t -= rint(t);
    float ang = (float)(t * twopi);
    sincosf(ang, &zi, &zr);
}

#define GC_GLOBAL __attribute__((address_space(1)))
// native vector types: loads through an address-space-qualified pointer need plain (non-class) types
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef signed char i8x2 __attribute__((ext_vector_type(2)));
typedef signed char i8x4 __attribute__((ext_vector_type(4)));

// IQ sample formats in HBM (gc_iq_format).  Integer samples are converted with a plain cast, like the
// reference's volk_gnsssdr_16ic_convert_32fc / interleaved-byte adapters do before the float correlators,
// so the arithmetic downstream is identical; only the bytes per sample change (8 / 4 / 2).
template <int FMT>
struct IqFmt;
template <>
struct IqFmt<GC_IQ_F32>
{
    typedef f32x2 elem;
    typedef f32x4 pair;
    static __device__ __forceinline__ f32x4 cvt(pair v) { return v; }
    static __device__ __forceinline__ f32x2 cvt1(elem v) { return v; }
};
template <>
struct IqFmt<GC_IQ_I16>
{
    typedef i16x2 elem;
    typedef i16x4 pair;
    static __device__ __forceinline__ f32x4 cvt(pair v) { return f32x4{(float)v.x, (float)v.y, (float)v.z, (float)v.w}; }
    static __device__ __forceinline__ f32x2 cvt1(elem v) { return f32x2{(float)v.x, (float)v.y}; }
};
template <>
struct IqFmt<GC_IQ_I8>
{
    typedef i8x2 elem;
    typedef i8x4 pair;
    static __device__ __forceinline__ f32x4 cvt(pair v) { return f32x4{(float)v.x, (float)v.y, (float)v.z, (float)v.w}; }
    static __device__ __forceinline__ f32x2 cvt1(elem v) { return f32x2{(float)v.x, (float)v.y}; }
};

// (int)floorf(x) in one instruction (v_floor_f32 + v_cvt_i32_f32 otherwise)
static __device__ __forceinline__ int floor_to_int(float x)
{
    int i;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(x));
    return i;
}

// one lane's 16-byte piece (two samples) of chunk c: uniform chunk base + constant per-lane offset, SGPR-base global load
template <int FMT, int THREADS, bool NT>
static __device__ __forceinline__ f32x4 trk_load_chunk(const GC_GLOBAL typename IqFmt<FMT>::elem* __restrict__ base, int c)
{
    typedef typename IqFmt<FMT>::pair pair_t;
    const GC_GLOBAL char* p = reinterpret_cast<const GC_GLOBAL char*>(base) + (size_t)c * (THREADS * sizeof(pair_t)) + threadIdx.x * sizeof(pair_t);
    if (NT) return IqFmt<FMT>::cvt(__builtin_nontemporal_load(reinterpret_cast<const GC_GLOBAL pair_t*>(p)));
    return IqFmt<FMT>::cvt(*reinterpret_cast<const GC_GLOBAL pair_t*>(p));
}

// Main loop over the chunks [c0, c1) of one (channel, epoch, slice).
//   WINDOWED: table[] holds code[(lo + k) mod L], indices need no wrap
//   else    : table[] holds code[0..L), indices are wrapped with the reference's modulo
//   CC      : the code table holds complex chips (re, im interleaved): Cpu_Multicorrelator's
//             32fc_xn_resampler_32fc_xn + 32fc_x2_rotator_dot_prod_32fc_xn pair (plain mode only)
//   SC16    : Cpu_Multicorrelator_16sc arithmetic (volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn): the rotated
//             sample is rounded to int16, multiplied with an int16 complex chip (wrapping to int16 like the
//             reference's lv_16sc_t product) and accumulated in 32-bit integers; the table holds one
//             (re16, im16) pair per 4-byte word and the accumulators carry integer bit patterns
//   THREADS : workgroup size (a multiple of 64); a chunk is 2*THREADS samples
//   DATA    : pilot tracking (dll_pll_veml_tracking.cc:899-910): a second, prompt-only correlator on the data
//             component's replica, slaved to the pilot prompt tap (same shift pointer, same NCO scalars, hence
//             the same chip index): table2 holds the data replica laid out like `table`; its sum goes to
//             accr[NTAPS] / acci[NTAPS] (plain float mode only)
//   PF      : full chunks kept in flight per lane ahead of the one being processed (2 everywhere: deeper queues were measured
//             no faster in the batched kernel and slower in the closed-loop kernel)
template <int NTAPS, bool HDR, bool HDC, bool WINDOWED, int FMT, bool CC = false, bool SC16 = false, int THREADS = TRK_THREADS, bool DATA = false, int PF = TRK_PF, bool NT = (TRK_NT != 0), bool WHOLE = true>
static __device__ __forceinline__ void trk_loop(const GC_GLOBAL typename IqFmt<FMT>::elem* __restrict__ base, const float* __restrict__ table,
    int a, int N, int V, int c0, int c1, int lo, int L, float step, float rem, float rate,
    const float (&shifts)[NTAPS], const int (&tap_delay)[NTAPS], double theta0, double dtheta, double drate, float lnmod,
    float (&accr)[NTAPS + (DATA ? 1 : 0)], float (&acci)[NTAPS + (DATA ? 1 : 0)], const float* __restrict__ table2,
    int lc0, int lc1, f32x4 pre0, f32x4 pre1)
{
    // lc0, lc1: chunks [lc0, lc1) lie inside the channel's buffer as WHOLE chunks (the ragged first / last chunk of a window usually
    // does: its neighbours in the stream are there), so they are fetched like the others -- 16-byte loads, prefetched -- and the
    // samples outside the window zeroed in registers; pre0 / pre1: chunks c0 and c0 + 1 where they are such chunks, requested by
    // the caller before its prologue (window fill, carrier angles), which then overlaps their latency
    static_assert(!DATA || (!CC && !SC16), "the data-component correlator exists for real float replicas only");
    constexpr int CHUNK = 2 * THREADS;
    constexpr int PT = NTAPS / 2;  // prompt tap of E/P/L and VE/E/P/L/VL
    const int tid = threadIdx.x;
    const float* tl = table - (CC ? 2 * lo : lo);  // windowed lookups index with the unwrapped chip number
    const float* tl2 = DATA ? table2 - lo : nullptr;
    // one tap, one sample: acc += y * code[i]
    auto mac = [&](float yr, float yi, int i, float& ar, float& ai) {
        if (SC16)
            {
                // yr / yi already hold the rounded int16 sample (as integers in float registers' bits)
                const int bits = __float_as_int(WINDOWED ? tl[i] : table[posmod(i, L)]);
                const int cr = (short)(bits & 0xffff), ci = bits >> 16;
                const int sr = __float_as_int(yr), si = __float_as_int(yi);
                const int pr = (short)(sr * cr - si * ci);  // lv_16sc_t product: computed in int, stored as int16
                const int pi = (short)(sr * ci + si * cr);
                ar = __int_as_float(__float_as_int(ar) + pr);
                ai = __int_as_float(__float_as_int(ai) + pi);
            }
        else if (CC)
            {
                const f32x2 cv = WINDOWED ? reinterpret_cast<const f32x2*>(tl)[i] : reinterpret_cast<const f32x2*>(table)[posmod(i, L)];
                ar = fmaf(yr, cv.x, ar);
                ar = fmaf(-yi, cv.y, ar);
                ai = fmaf(yr, cv.y, ai);
                ai = fmaf(yi, cv.x, ai);
            }
        else
            {
                const float cv = WINDOWED ? tl[i] : table[posmod(i, L)];
                ar = fmaf(yr, cv, ar);
                ai = fmaf(yi, cv, ai);
            }
    };

    // per-chunk advance of the two per-lane rotators: exp(j*CHUNK*dtheta)
    float wr = 1.0f, wi = 0.0f;
    if (!HDC)
        {
            double t = (double)CHUNK * dtheta * 0.15915494309189533577;
            t -= rint(t);
            sincosf((float)(t * 6.283185307179586477), &wi, &wr);
        }
    // second sample of the lane's pair: one more step of the carrier
    float w1r = 1.0f, w1i = 0.0f;
    if (!HDC)
        {
            double t = dtheta * 0.15915494309189533577;
            t -= rint(t);
            sincosf((float)(t * 6.283185307179586477), &w1i, &w1r);
        }
    float z0r = 1.0f, z0i = 0.0f, z1r = 1.0f, z1i = 0.0f;

    // A chunk is "full" when every lane's two samples lie inside the window: plain 16-byte loads, no
    // masks, no clamps.  Only the first chunk (odd-aligned window) and the last one can be ragged; they
    // are processed outside the pipelined interior loop.
    auto chunk_is_full = [&](int c) { return (c > 0 || a == 0) && (c + 1) * CHUNK <= V; };
    // a chunk that may be fetched whole: every full chunk (it lies inside the window, mirror of a ring included), and the ragged ones
    // the caller found inside the buffer
    auto loadable = [&](int c) { return chunk_is_full(c) || (WHOLE && c >= lc0 && c < lc1); };
    auto load_full = [&](int c) -> f32x4 { return trk_load_chunk<FMT, THREADS, NT>(base, c); };
    // a whole chunk's piece with the samples outside the window zeroed
    auto mask_window = [&](int c, f32x4 x) -> f32x4 {
        const int v = c * CHUNK + tid * 2;
        const bool in0 = v >= a && v < V, in1 = v + 1 >= a && v + 1 < V;
        x.x = in0 ? x.x : 0.f;
        x.y = in0 ? x.y : 0.f;
        x.z = in1 ? x.z : 0.f;
        x.w = in1 ? x.w : 0.f;
        return x;
    };
    auto load_masked = [&](int c) -> f32x4 {
        const int v = c * CHUNK + tid * 2;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (v >= a && v < V)
            {
                f32x2 s = IqFmt<FMT>::cvt1(base[v]);
                x.x = s.x;
                x.y = s.y;
            }
        if (v + 1 >= a && v + 1 < V)
            {
                f32x2 s = IqFmt<FMT>::cvt1(base[v + 1]);
                x.z = s.x;
                x.w = s.y;
            }
        return x;
    };
    // exact re-evaluation of both rotators (every TRK_RESYNC chunks)
    auto resync = [&](int c) {
        if (!HDC)
            {
                const int v = c * CHUNK + tid * 2;
                carrier_at<false>(v - a, theta0, dtheta, 0.0, z0r, z0i);
                z1r = fmaf(z0r, w1r, -(z0i * w1i));
                z1i = fmaf(z0r, w1i, z0i * w1r);
            }
    };
    // one chunk of work, then one step of the rotators
    auto process = [&](auto full_tag, int c, const f32x4 xc) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int v = c * CHUNK + tid * 2;
        int n0 = v - a, n1 = v + 1 - a;
        if (!FULL)
            {
                // masked lanes carry zero input; keep their sample number inside the window (an EMPTY window with an
                // odd start still runs one all-masked chunk: sample 0 is what the LDS window was built for)
                const int nmax = max(N - 1, 0);
                n0 = min(max(n0, 0), nmax);
                n1 = min(max(n1, 0), nmax);
            }
        if (HDC)
            {
                carrier_at<true>(n0, theta0, dtheta, drate, z0r, z0i);
                carrier_at<true>(n1, theta0, dtheta, drate, z1r, z1i);
                // the reference's phase_doppler is never renormalised: its modulus drifts as
                // |phase_inc|^n; samples with n % 256 == 0 are renormalised before use
                float g0 = (n0 & 255) ? fmaf((float)n0, lnmod, 1.0f) : 1.0f;
                float g1 = (n1 & 255) ? fmaf((float)n1, lnmod, 1.0f) : 1.0f;
                z0r *= g0;
                z0i *= g0;
                z1r *= g1;
                z1i *= g1;
            }
        // ---- wipe-off: y = x * z ----
        float y0r, y0i, y1r, y1i;
        if (SC16)
            {
                // tmp32 = (float)x * phase with the reference's separate products (…16ic_x2_rotator_dot_prod_16ic_xn.h:93),
                // tmp16 = (int16_t)rintf(.) (:94)
                y0r = __int_as_float((int)(short)__float2int_rn(xc.x * z0r - xc.y * z0i));
                y0i = __int_as_float((int)(short)__float2int_rn(xc.x * z0i + xc.y * z0r));
                y1r = __int_as_float((int)(short)__float2int_rn(xc.z * z1r - xc.w * z1i));
                y1i = __int_as_float((int)(short)__float2int_rn(xc.z * z1i + xc.w * z1r));
            }
        else
            {
                y0r = fmaf(xc.x, z0r, -(xc.y * z0i));
                y0i = fmaf(xc.x, z0i, xc.y * z0r);
                y1r = fmaf(xc.z, z1r, -(xc.w * z1i));
                y1i = fmaf(xc.z, z1i, xc.w * z1r);
            }
        // ---- code NCO + E/P/L accumulation ----
        if (HDR)
            {
#pragma unroll
                for (int t = 0; t < NTAPS; t++)
                    {
                        // tap t at sample n reads tap 0 at sample (n + delay) wrapped at N (…:100-106)
                        int m0 = n0 + tap_delay[t];
                        int m1 = n1 + tap_delay[t];
                        if (t > 0)
                            {
                                m0 = (m0 >= N) ? m0 - N : m0;
                                m1 = (m1 >= N) ? m1 - N : m1;
                                m0 = min(max(m0, 0), max(N - 1, 0));
                                m1 = min(max(m1, 0), max(N - 1, 0));
                            }
                        const int i0 = chip_index_hd(step, rate, (unsigned)m0, shifts[0], rem);
                        const int i1 = chip_index_hd(step, rate, (unsigned)m1, shifts[0], rem);
                        mac(y0r, y0i, i0, accr[t], acci[t]);
                        mac(y1r, y1i, i1, accr[t], acci[t]);
                    }
                if constexpr (DATA)
                    {
                        // the data correlator is its own one-tap object: ITS first tap is the prompt shift, evaluated with the
                        // high-dynamics formula at the sample itself (no tap delay)
                        const int j0 = chip_index_hd(step, rate, (unsigned)n0, shifts[PT], rem);
                        const int j1 = chip_index_hd(step, rate, (unsigned)n1, shifts[PT], rem);
                        const float d0 = WINDOWED ? tl2[j0] : table2[posmod(j0, L)];
                        const float d1 = WINDOWED ? tl2[j1] : table2[posmod(j1, L)];
                        accr[NTAPS] = fmaf(y0r, d0, accr[NTAPS]);
                        acci[NTAPS] = fmaf(y0i, d0, acci[NTAPS]);
                        accr[NTAPS] = fmaf(y1r, d1, accr[NTAPS]);
                        acci[NTAPS] = fmaf(y1i, d1, acci[NTAPS]);
                    }
            }
        else
            {
                const float s0 = step * (float)n0, s1 = step * (float)n1;
#pragma unroll
                for (int t = 0; t < NTAPS; t++)
                    {
                        const int i0 = floor_to_int((s0 + shifts[t]) - rem);
                        const int i1 = floor_to_int((s1 + shifts[t]) - rem);
                        mac(y0r, y0i, i0, accr[t], acci[t]);
                        mac(y1r, y1i, i1, accr[t], acci[t]);
                        if constexpr (DATA)
                            if (t == PT)
                            {
                                const float d0 = WINDOWED ? tl2[i0] : table2[posmod(i0, L)];
                                const float d1 = WINDOWED ? tl2[i1] : table2[posmod(i1, L)];
                                accr[NTAPS] = fmaf(y0r, d0, accr[NTAPS]);
                                acci[NTAPS] = fmaf(y0i, d0, acci[NTAPS]);
                                accr[NTAPS] = fmaf(y1r, d1, accr[NTAPS]);
                                acci[NTAPS] = fmaf(y1i, d1, acci[NTAPS]);
                            }
                    }
            }
        if (!HDC)
            {
                // advance both rotators by one chunk
                const float t0 = fmaf(z0r, wr, -(z0i * wi));
                z0i = fmaf(z0r, wi, z0i * wr);
                z0r = t0;
                const float t1 = fmaf(z1r, wr, -(z1i * wi));
                z1i = fmaf(z1r, wi, z1i * wr);
                z1r = t1;
            }
    };

    if (c0 >= c1) return;
#if !TRK_PRELOAD
    if (loadable(c0)) pre0 = load_full(c0);
    if (c0 + 1 < c1 && loadable(c0 + 1)) pre1 = load_full(c0 + 1);
#endif
    int c = c0;
    // ---- ragged head (at most one chunk) ----
    if (!chunk_is_full(c))
        {
            resync(c);
            process(std::false_type{}, c, loadable(c) ? mask_window(c, pre0) : load_masked(c));
            ++c;
        }
    // ---- interior: full chunks, TRK_PF loads in flight ahead of the chunk being processed ----
    const int cf1 = (c < c1 && !chunk_is_full(c1 - 1)) ? c1 - 1 : c1;
    if (c < cf1)
        {
            // two loads in flight per lane; the loop is unrolled by two so that the buffers keep their
            // registers (a rotating queue would have to wait for the youngest load to move it)
            // prefetches past the end re-read the last chunk that may be fetched whole (the ragged tail where it is one: its lines
            // are then on their way when the tail asks for them; else the last full chunk; value unused): no branch
            // in the loop body, so the compiler can wait with vmcnt(1) and keep one load in flight
            const int clast = (cf1 < c1 && loadable(cf1)) ? cf1 : cf1 - 1;
            if constexpr (PF == 2)
                {
                    // chunks c0 and c0 + 1 were requested by the caller where they may be fetched whole
                    f32x4 x0 = (c == c0) ? pre0 : (loadable(c0 + 1) ? pre1 : load_full(c));
                    f32x4 x1 = (c == c0 && c0 + 1 < c1 && loadable(c0 + 1)) ? pre1 : load_full(min(c + 1, clast));
                    while (c < cf1)
                        {
                            if ((c - c0) % TRK_RESYNC == 0) resync(c);
                            const int gend = min(cf1, c0 + ((c - c0) / TRK_RESYNC + 1) * TRK_RESYNC);
                            while (c + 2 <= gend)
                                {
                                    const f32x4 xa = x0;
                                    x0 = load_full(min(c + 2, clast));
                                    process(std::true_type{}, c, xa);
                                    const f32x4 xb = x1;
                                    x1 = load_full(min(c + 3, clast));
                                    process(std::true_type{}, c + 1, xb);
                                    c += 2;
                                }
                            if (c < gend)
                                {
                                    // odd chunk left in this group: the buffers swap roles
                                    const f32x4 xa = x0;
                                    x0 = x1;
                                    x1 = load_full(min(c + 2, clast));
                                    process(std::true_type{}, c, xa);
                                    ++c;
                                }
                        }
                }
            else
                {
                    // the same scheme with PF buffers: unrolled by PF, leftovers of a group one chunk at a time (the queue shifts)
                    f32x4 x[PF];
#pragma unroll
                    for (int u = 0; u < PF; u++) x[u] = load_full(min(c + u, clast));
                    while (c < cf1)
                        {
                            if ((c - c0) % TRK_RESYNC == 0) resync(c);
                            const int gend = min(cf1, c0 + ((c - c0) / TRK_RESYNC + 1) * TRK_RESYNC);
                            while (c + PF <= gend)
                                {
#pragma unroll
                                    for (int u = 0; u < PF; u++)
                                        {
                                            const f32x4 xa = x[u];
                                            x[u] = load_full(min(c + PF + u, clast));
                                            process(std::true_type{}, c + u, xa);
                                        }
                                    c += PF;
                                }
                            while (c < gend)
                                {
                                    const f32x4 xa = x[0];
#pragma unroll
                                    for (int u = 0; u + 1 < PF; u++) x[u] = x[u + 1];
                                    x[PF - 1] = load_full(min(c + PF, clast));
                                    process(std::true_type{}, c, xa);
                                    ++c;
                                }
                        }
                }
        }
    // ---- ragged tail (at most one chunk) ----
    if (c < c1)
        {
            if ((c - c0) % TRK_RESYNC == 0) resync(c);
            // (a window of two ragged chunks and nothing between them finds its tail in pre1)
            process(std::false_type{}, c, loadable(c) ? mask_window(c, (c == c0 + 1 && PF == 2) ? pre1 : load_full(c)) : load_masked(c));
        }
}

// One (channel, epoch, slice): builds the LDS code window, streams the IQ window, reduces the tap sums.
// Every thread of the 256-thread workgroup must call it; the sum of tap `tid` is returned to the threads
// with tid < NTAPS (others get 0).  lds: trk_hdr_floats(THREADS) + lds_table_floats floats of dynamic LDS.
// With SC16 the returned pair holds the two 32-bit integer sums as bit patterns.
// With DATA (pilot tracking) cd.code2 is the data component's replica: its prompt sum is returned to tid == NTAPS and the
// window needs 2x the LDS floats.
#ifdef GNSSCORR_EXPERIMENTS
#include "trk_chips.hpp"
#else
// the chip-domain loop is compiled only into experiments builds; its names are declared so that the CHIPS = false instantiations parse
#define TRK_SEG 512
#define TRK_CHIPS_WAVE_FLOATS 0
template <int NTAPS, bool WINDOWED, int THREADS, bool DATA, class B, class... A>
static __device__ void trk_loop_chips(B, A...);
#endif

// CHIPS: the plain float loop summed per chip (trk_chips.hpp); the caller provides THREADS / 64 * TRK_CHIPS_WAVE_FLOATS more
// floats of LDS behind the code window (rounded up to 16 bytes)
// NT: nontemporal IQ loads -- for launches that stream every window once from HBM; off where many workgroups re-read one stream
// from the caches over many epochs (the closed-loop kernel: 0.81 instead of 0.69 ms for 256 channels x 64 periods with the hint)
// resident: the caller keeps R[i] = code[i mod L], i < 2 L + TRK_RESIDENT_PAD (and, with DATA, the data replica's image right behind
// it) in the LDS table across calls -- the closed-loop kernel, one workgroup per channel for many code periods: nothing is filled
// here, a window [lo, hi] of up to L + TRK_RESIDENT_PAD chips is addressed inside the doubled image, longer ones through the first L
// entries with the modulo form.  Same chips, same sums as the filled window.
#define TRK_RESIDENT_PAD 64
static __device__ __forceinline__ int trk_resident_floats(int L) { return 2 * L + TRK_RESIDENT_PAD; }
template <int THREADS>
static __device__ __forceinline__ void trk_fill_resident(float* lds, const float* code, const float* code2, int L)
{
    float* table = lds + trk_hdr_floats(THREADS);
    const GC_GLOBAL float* c = (const GC_GLOBAL float*)code;
    const GC_GLOBAL float* c2 = (const GC_GLOBAL float*)code2;
    const int n = trk_resident_floats(L);
    for (int k = threadIdx.x; k < n; k += THREADS)
        {
            int i = k;
            i = (i >= L) ? i - L : i;
            i = (i >= L) ? i - L : i;
            table[k] = c[i];
            if (code2) table[n + k] = c2[i];
        }
}

// GTAB: the launch may have been given LESS LDS than the whole code table needs (the batched kernel sizes its window by what a slice of
// a nominal epoch touches: more workgroups per CU for long codes).  A record whose chips neither fit that window nor allow the whole
// table in LDS then reads the table from global memory with the modulo form (slow, correct, rare: a code step far off the nominal one)
template <int NTAPS, bool HDR, bool HDC, int FMT, bool CC = false, bool SC16 = false, int THREADS = TRK_THREADS, bool DATA = false, int PF = TRK_PF, bool CHIPS = false, bool NT = (TRK_NT != 0), bool WHOLE = true, bool GTAB = false>
static __device__ __forceinline__ float2 trk_epoch(const TrkChan& cd, const gc_epoch_params& p, int slice, int n_slices,
    int lds_table_floats, float* lds, int align_pairs = TRK_ALIGN_PAIRS, bool resident = false)
{
    // lds[0..HDRF): header (wave partials); then the code window
    static_assert(!CHIPS || (!HDR && !HDC && !CC && !SC16 && FMT == GC_IQ_F32), "the chip-domain loop exists for the plain float correlator");
    constexpr int CHUNK = CHIPS ? TRK_SEG : 2 * THREADS;  // unit of the slicing
    constexpr int HDRF = trk_hdr_floats(THREADS);
    float* table = lds + HDRF;
    const int tid = threadIdx.x;
    const int N = p.n_samples;
    const int L = cd.code_len;

    typedef typename IqFmt<FMT>::elem elem_t;
    const unsigned long long off = cd.ring_len ? p.sample_offset % cd.ring_len : p.sample_offset;  // (one 64-bit division per epoch)
    const elem_t* iq = static_cast<const elem_t*>(cd.iq) + off;
    // samples between the aligned address the chunks are counted from and the window's first one (they are masked out of the first
    // chunk): one pair keeps the pair loads legal, eight pairs of float samples (128 bytes) also keep every wave-instruction's 1 KiB on
    // eight cache lines instead of nine
    // (align_pairs = 1 for the level-1 calls: their staging keeps a window's 16-byte phase, and with it the summation order and the
    // bits of the result, whether the call runs alone or in a batch)
    const int a = (int)((reinterpret_cast<uintptr_t>(iq) / sizeof(elem_t)) & (2 * align_pairs - 1));
    // sample n lives at base[n + a].  IQ lives in HBM: global (not flat) loads
    const GC_GLOBAL elem_t* base = (const GC_GLOBAL elem_t*)(iq - a);
    const int V = N + a;
    const int n_chunks = (V + CHUNK - 1) / CHUNK;
    const int cps = (n_chunks + n_slices - 1) / n_slices;
    const int c0 = slice * cps;
    const int c1 = min(n_chunks, c0 + cps);
    // Chunks [lc0, lc1) lie inside the channel's buffer as whole chunks: the ragged ends of a window are then fetched like its
    // interior (trk_loop).  The first two chunks are requested HERE, before the prologue below, which hides their latency.
    int lc0 = 0, lc1 = 0;
    f32x4 pre0 = {0.f, 0.f, 0.f, 0.f}, pre1 = {0.f, 0.f, 0.f, 0.f};
    if (!CHIPS)
        {
            const unsigned long long end = cd.ring_len ? (unsigned long long)cd.ring_len : cd.n_iq;  // samples at cd.iq (a ring's mirror is not counted)
            const unsigned long long room = end > off ? end - off : 0;
            lc0 = off >= (unsigned long long)a ? 0 : 1;
            lc1 = (int)min((unsigned long long)n_chunks, (room + (unsigned long long)a) / CHUNK);
#if TRK_PRELOAD
            if (c0 < c1 && c0 >= lc0 && c0 < lc1) pre0 = trk_load_chunk<FMT, THREADS, NT>(base, c0);
            if (c0 + 1 < c1 && c0 + 1 >= lc0 && c0 + 1 < lc1) pre1 = trk_load_chunk<FMT, THREADS, NT>(base, c0 + 1);
#endif
        }

    const float step = p.code_phase_step_chips;
    const float rem = p.rem_code_phase_chips;
    const float rate = p.code_phase_rate_step_chips;

    float shifts[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; t++) shifts[t] = cd.shifts[t];

    // high-dynamics resampler: taps >= 1 are tap 0 delayed by whole samples
    int tap_delay[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; t++) tap_delay[t] = 0;
    if (HDR)
        {
            unsigned acc = 0;
#pragma unroll
            for (int t = 1; t < NTAPS; t++)
                {
                    acc += (unsigned)(int)rintf((shifts[t] - shifts[t - 1]) / step);
                    tap_delay[t] = (int)acc;
                }
        }

    // ---- carrier angles ----
    // theta0 only needs ~1e-7 rad (it is a constant rotation of the result): float atan2.  The
    // per-sample increment is multiplied by up to 1e5 samples: float64, from the series of atan(t),
    // t = im/re, whenever |t| is small (any Doppler below ~2 % of the sampling rate), atan2 otherwise.
    double theta0, dtheta, drate = 0.0;
    float lnmod = 0.0f;
    {
        theta0 = (double)atan2f(p.phase0_im, p.phase0_re);
        auto small_angle = [](float im, float re) -> double {
            const double t = (double)im / (double)re;
            if (re > 0.0f && fabs(t) < 0.03)
                {
                    const double t2 = t * t;
                    return t * (1.0 + t2 * (-1.0 / 3.0 + t2 * (0.2 + t2 * (-1.0 / 7.0 + t2 * (1.0 / 9.0)))));
                }
            return atan2((double)im, (double)re);
        };
        dtheta = small_angle(p.phase_inc_im, p.phase_inc_re);
        if (HDC)
            {
                drate = small_angle(p.phase_rate_im, p.phase_rate_re);
                lnmod = (float)(0.5 * log((double)p.phase_inc_re * p.phase_inc_re + (double)p.phase_inc_im * p.phase_inc_im));
            }
    }

    // ---- code window in LDS ----
    // Sample numbers this slice touches (clamped lanes included): [n_lo, n_hi].
    int n_lo, n_hi;
    if (HDR)
        {
            n_lo = 0;
            n_hi = max(N - 1, 0);
        }
    else
        {
            n_lo = max(c0 * CHUNK - a, 0);
            n_hi = max(min(c1 * CHUNK - a, N) - 1, n_lo);
        }
    float smin = shifts[0], smax = shifts[0];
    if (!HDR)
        {
#pragma unroll
            for (int t = 1; t < NTAPS; t++)
                {
                    smin = fminf(smin, shifts[t]);
                    smax = fmaxf(smax, shifts[t]);
                }
        }
    int lo, hi;
    bool monotone;
    if (HDR)
        {
            lo = chip_index_hd(step, rate, (unsigned)n_lo, shifts[0], rem);
            hi = chip_index_hd(step, rate, (unsigned)n_hi, shifts[0], rem);
            if (DATA)
                {
                    // the data tap walks the same samples with the prompt shift instead of the first tap's
                    lo = min(lo, chip_index_hd(step, rate, (unsigned)n_lo, shifts[NTAPS / 2], rem));
                    hi = max(hi, chip_index_hd(step, rate, (unsigned)n_hi, shifts[NTAPS / 2], rem));
                }
            // float ops are monotone, so the index is monotone in n when both terms are
            monotone = (step > 0.0f) && (rate >= 0.0f) && ((unsigned long long)n_hi * n_hi < 0xffffffffull);
        }
    else
        {
            lo = chip_index(step, (float)n_lo, smin, rem);
            hi = chip_index(step, (float)n_hi, smax, rem);
            monotone = (step >= 0.0f);
        }
    const long long span_ll = (long long)hi - (long long)lo + 1;
    typedef typename std::conditional<CC, f32x2, float>::type chip_t;
    bool windowed = monotone && span_ll > 0 && span_ll * (long long)(sizeof(chip_t) / sizeof(float)) * (DATA ? 2 : 1) <= (long long)lds_table_floats;
    const GC_GLOBAL chip_t* code = (const GC_GLOBAL chip_t*)cd.code;
    chip_t* table_c = reinterpret_cast<chip_t*>(table);
    // data-component replica (pilot tracking): same layout, right behind the pilot's table
    constexpr int NACC = NTAPS + (DATA ? 1 : 0);
    float* table2 = DATA ? table + (windowed ? (int)span_ll : L) : nullptr;
    const GC_GLOBAL float* code2 = (const GC_GLOBAL float*)cd.code2;
    const bool table_fits = !GTAB || (long long)L * (long long)(sizeof(chip_t) / sizeof(float)) * (DATA ? 2 : 1) <= (long long)lds_table_floats;
    if (resident && !CC)
        {
            // the doubled image is in place: address the window inside it (chip - lo + cbase < 2 L + pad), or fall back to its
            // first L entries with the modulo form
            windowed = monotone && span_ll > 0 && span_ll <= (long long)L + TRK_RESIDENT_PAD;
            if (windowed) lo -= posmod(lo, L);
            if (DATA) table2 = table + trk_resident_floats(L);
        }
    else if (windowed)
        {
            const int span = (int)span_ll;
            const int cbase = posmod(lo, L);
            if (cbase + span <= 3 * L)
                {
                    for (int k = tid; k < span; k += THREADS)
                        {
                            int i = cbase + k;  // < 3L: two conditional subtractions instead of a division
                            i = (i >= L) ? i - L : i;
                            i = (i >= L) ? i - L : i;
                            table_c[k] = code[i];
                            if (DATA) table2[k] = code2[i];
                        }
                }
            else
                {
                    for (int k = tid; k < span; k += THREADS)
                        {
                            table_c[k] = code[(cbase + k) % L];
                            if (DATA) table2[k] = code2[(cbase + k) % L];
                        }
                }
        }
    else if (table_fits)
        {
            for (int k = tid; k < L; k += THREADS)
                {
                    table_c[k] = code[k];
                    if (DATA) table2[k] = code2[k];
                }
        }
    __syncthreads();

    float accr[NACC], acci[NACC];
#pragma unroll
    for (int t = 0; t < NACC; t++) accr[t] = acci[t] = 0.0f;

    if constexpr (CHIPS)
        {
            float* scratch = table + ((lds_table_floats + 3) & ~3);
            if (windowed)
                trk_loop_chips<NTAPS, true, THREADS, DATA>(base, table, table2, a, N, V, c0, c1, lo, L, step, rem, shifts, theta0, dtheta, scratch, accr, acci);
            else
                trk_loop_chips<NTAPS, false, THREADS, DATA>(base, table, table2, a, N, V, c0, c1, 0, L, step, rem, shifts, theta0, dtheta, scratch, accr, acci);
        }
    else if (windowed)
        trk_loop<NTAPS, HDR, HDC, true, FMT, CC, SC16, THREADS, DATA, PF, NT, WHOLE>(base, table, a, N, V, c0, c1, lo, L, step, rem, rate, shifts, tap_delay, theta0, dtheta, drate, lnmod, accr, acci, table2, lc0, lc1, pre0, pre1);
    else if (table_fits)
        trk_loop<NTAPS, HDR, HDC, false, FMT, CC, SC16, THREADS, DATA, PF, NT, WHOLE>(base, table, a, N, V, c0, c1, 0, L, step, rem, rate, shifts, tap_delay, theta0, dtheta, drate, lnmod, accr, acci, table2, lc0, lc1, pre0, pre1);
    else if constexpr (GTAB)
        trk_loop<NTAPS, HDR, HDC, false, FMT, CC, SC16, THREADS, DATA, PF, NT, WHOLE>(base, reinterpret_cast<const float*>(cd.code), a, N, V, c0, c1, 0, L, step, rem, rate, shifts, tap_delay, theta0, dtheta, drate, lnmod, accr, acci, cd.code2, lc0, lc1, pre0, pre1);

    // ---- reduction: lanes -> wave (shuffles) -> workgroup (LDS) ----
    __syncthreads();  // the code window has been consumed by every thread
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int t = 0; t < NACC; t++)
        {
            float sr, si;
            if (SC16)
                {
                    sr = __int_as_float(wave_sum_i(__float_as_int(accr[t])));
                    si = __int_as_float(wave_sum_i(__float_as_int(acci[t])));
                }
            else
                {
                    sr = wave_sum(accr[t]);
                    si = wave_sum(acci[t]);
                }
            if (lane == 63)
                {
                    lds[(wave * NACC + t) * 2 + 0] = sr;
                    lds[(wave * NACC + t) * 2 + 1] = si;
                }
        }
    __syncthreads();
    float2 r = make_float2(0.f, 0.f);
    if (tid < NACC)
        {
            float sr = 0.f, si = 0.f;
#pragma unroll
            for (int w = 0; w < THREADS / 64; w++)
                {
                    if (SC16)
                        {
                            sr = __int_as_float(__float_as_int(sr) + __float_as_int(lds[(w * NACC + tid) * 2 + 0]));
                            si = __int_as_float(__float_as_int(si) + __float_as_int(lds[(w * NACC + tid) * 2 + 1]));
                        }
                    else
                        {
                            sr += lds[(w * NACC + tid) * 2 + 0];
                            si += lds[(w * NACC + tid) * 2 + 1];
                        }
                }
            r = make_float2(sr, si);
        }
    return r;
}

#endif  // TRK_DEVICE_HPP
