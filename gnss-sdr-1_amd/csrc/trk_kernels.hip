// trk_kernels.hip -- batched tracking multicorrelator for gfx950 (MI355X).
//
// One kernel fuses what the reference does in two volk_gnsssdr passes per
// channel-epoch (cpu_multicorrelator_real_codes.cc:129-152):
//   code NCO    volk_gnsssdr_32f_xn_resampler_32f_xn_generic
//               (kernels/volk_gnsssdr/volk_gnsssdr_32f_xn_resampler_32f_xn.h:77-94)
//               / ..._high_dynamics_resampler_32f_xn_generic (…:81-107)
//   carrier NCO + E/P/L dot products
//               volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn_generic (…:81-113)
//               / ..._high_dynamic_rotator_dot_prod_32fc_xn_generic (…:82-116)
// The resampled replica (4*n_taps bytes per sample in the reference) is never
// materialised: chip indices are computed per sample with the generic kernel's
// float32 operation order (no FMA contraction in that expression: this file is
// built with -ffp-contract=off and uses fmaf() explicitly where fusing is
// wanted) and looked up in an LDS-resident window of the code table.
//
// Work decomposition: one 256-thread workgroup (4 wave64) per
// (channel, epoch, slice).  Each lane streams 16-byte (2 complex sample) loads,
// 1 KiB per wave-instruction, coalesced; per-lane partial sums for every tap are
// reduced with wave shuffles, then across the 4 waves through LDS.  No MFMA:
// 8 bytes of IQ per ~45 VALU operations is an HBM/VALU-bound complex MAC stream.
#include "gc_internal.h"
#include "trk_kernels.h"

#define TRK_THREADS 256
#define TRK_CHUNK 512  // samples per workgroup iteration (2 per lane)
#define TRK_HDR_FLOATS 64
#define TRK_RESYNC 32  // iterations between exact re-evaluations of the carrier phase

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

static __device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
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

template <int NTAPS, bool HDR, bool HDC>
__global__ __launch_bounds__(TRK_THREADS) void trk_multicorrelator_kernel(
    const TrkChan* __restrict__ chans, const gc_epoch_params* __restrict__ params,
    float2* __restrict__ out, float2* __restrict__ partial,
    int n_channels, int n_epochs, int n_slices, int lds_table_floats)
{
    extern __shared__ float lds[];
    // lds[0..63]: header (wave partials, broadcast doubles); lds[64..]: code window
    float* table = lds + TRK_HDR_FLOATS;
    double* hdr_d = reinterpret_cast<double*>(lds);  // [0]=theta0 [1]=dtheta [2]=drate [3]=lnmod

    // XCD-aware job mapping: workgroups are dealt round-robin over the 8 XCDs, so
    // blocks with equal (blockIdx % 8) share an L2.  All channels of one epoch
    // (which read the same IQ window when they share an RF stream) are given the
    // same residue.
    const int b = blockIdx.x;
    const int x = b & 7;
    int q = b >> 3;
    const int slice = q % n_slices;
    q /= n_slices;
    const int ch = q % n_channels;
    const int epoch = (q / n_channels) * 8 + x;
    if (epoch >= n_epochs) return;
    const int job = ch * n_epochs + epoch;

    const TrkChan cd = chans[ch];
    const gc_epoch_params p = params[job];
    const int tid = threadIdx.x;
    const int N = p.n_samples;
    const int L = cd.code_len;

    const float2* iq = cd.iq + p.sample_offset;
    const int a = (int)((reinterpret_cast<uintptr_t>(iq) >> 3) & 1);  // 1: window starts on the odd half of a 16-byte pair
    const float2* base = iq - a;                                       // 16-byte aligned; sample n lives at base[n + a]
    const int V = N + a;
    const int n_chunks = (V + TRK_CHUNK - 1) / TRK_CHUNK;
    const int cps = (n_chunks + n_slices - 1) / n_slices;
    const int c0 = slice * cps;
    const int c1 = min(n_chunks, c0 + cps);

    const float step = p.code_phase_step_chips;
    const float rem = p.rem_code_phase_chips;
    const float rate = p.code_phase_rate_step_chips;

    float shifts[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; t++) shifts[t] = cd.shifts[t];

    // high-dynamics resampler: taps >= 1 are tap 0 delayed by whole samples
    int tap_delay[NTAPS];
    if (HDR)
        {
            unsigned acc = 0;
            tap_delay[0] = 0;
#pragma unroll
            for (int t = 1; t < NTAPS; t++)
                {
                    acc += (unsigned)(int)rintf((shifts[t] - shifts[t - 1]) / step);
                    tap_delay[t] = (int)acc;
                }
        }

    // ---- carrier angles (one lane, double precision), broadcast through LDS ----
    if (tid == 0)
        {
            hdr_d[0] = atan2((double)p.phase0_im, (double)p.phase0_re);
            hdr_d[1] = atan2((double)p.phase_inc_im, (double)p.phase_inc_re);
            hdr_d[2] = HDC ? atan2((double)p.phase_rate_im, (double)p.phase_rate_re) : 0.0;
            hdr_d[3] = HDC ? 0.5 * log((double)p.phase_inc_re * p.phase_inc_re + (double)p.phase_inc_im * p.phase_inc_im) : 0.0;
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
            n_lo = max(c0 * TRK_CHUNK - a, 0);
            n_hi = max(min(c1 * TRK_CHUNK - a, N) - 1, n_lo);
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
    const bool windowed = monotone && span_ll > 0 && span_ll <= (long long)lds_table_floats;
    const float* code = cd.code;
    if (windowed)
        {
            const int span = (int)span_ll;
            const int cbase = posmod(lo, L);
            for (int k = tid; k < span; k += TRK_THREADS) table[k] = code[(cbase + k) % L];
        }
    else
        {
            for (int k = tid; k < L; k += TRK_THREADS) table[k] = code[k];
        }
    __syncthreads();
    const double theta0 = hdr_d[0], dtheta = hdr_d[1], drate = hdr_d[2];
    const float lnmod = (float)hdr_d[3];

    float accr[NTAPS], acci[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; t++) accr[t] = acci[t] = 0.0f;

    // per-chunk advance of the two per-lane rotators: exp(j*TRK_CHUNK*dtheta)
    float wr = 1.0f, wi = 0.0f;
    if (!HDC)
        {
            double t = (double)TRK_CHUNK * dtheta * 0.15915494309189533577;
            t -= rint(t);
            sincosf((float)(t * 6.283185307179586477), &wi, &wr);
        }

    float z0r = 1.0f, z0i = 0.0f, z1r = 1.0f, z1i = 0.0f;
    int since_sync = TRK_RESYNC;  // force an exact evaluation on the first iteration

    const float4* base4 = reinterpret_cast<const float4*>(base);
    int v = c0 * TRK_CHUNK + tid * 2;
    float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
    // prefetch first chunk
    if (c0 < c1)
        {
            if (v >= a && v + 1 < V)
                xn = base4[v >> 1];
            else
                {
                    if (v >= a && v < V)
                        {
                            float2 s = base[v];
                            xn.x = s.x;
                            xn.y = s.y;
                        }
                    if (v + 1 >= a && v + 1 < V)
                        {
                            float2 s = base[v + 1];
                            xn.z = s.x;
                            xn.w = s.y;
                        }
                }
        }

    for (int c = c0; c < c1; ++c)
        {
            const float4 xc = xn;
            const int vc = v;
            v += TRK_CHUNK;
            // prefetch the next chunk while this one is processed
            xn = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c + 1 < c1)
                {
                    if (v + 1 < V)  // v >= a always holds past the first chunk
                        xn = base4[v >> 1];
                    else if (v < V)
                        {
                            float2 s = base[v];
                            xn.x = s.x;
                            xn.y = s.y;
                        }
                }

            // sample numbers, clamped so that masked lanes (zero input) still index inside the window
            const int n0 = min(max(vc - a, 0), N - 1);
            const int n1 = min(max(vc + 1 - a, 0), N - 1);

            // ---- carrier ----
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
            else if (since_sync >= TRK_RESYNC)
                {
                    carrier_at<false>(vc - a, theta0, dtheta, 0.0, z0r, z0i);
                    carrier_at<false>(vc + 1 - a, theta0, dtheta, 0.0, z1r, z1i);
                    since_sync = 0;
                }
            since_sync++;

            // ---- wipe-off: y = x * z ----
            const float y0r = fmaf(xc.x, z0r, -(xc.y * z0i));
            const float y0i = fmaf(xc.x, z0i, xc.y * z0r);
            const float y1r = fmaf(xc.z, z1r, -(xc.w * z1i));
            const float y1i = fmaf(xc.z, z1i, xc.w * z1r);

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
                                    m0 = min(max(m0, 0), N - 1);
                                    m1 = min(max(m1, 0), N - 1);
                                }
                            int i0 = chip_index_hd(step, rate, (unsigned)m0, shifts[0], rem);
                            int i1 = chip_index_hd(step, rate, (unsigned)m1, shifts[0], rem);
                            float cv0, cv1;
                            if (windowed)
                                {
                                    cv0 = table[i0 - lo];
                                    cv1 = table[i1 - lo];
                                }
                            else
                                {
                                    cv0 = table[posmod(i0, L)];
                                    cv1 = table[posmod(i1, L)];
                                }
                            accr[t] = fmaf(y0r, cv0, accr[t]);
                            acci[t] = fmaf(y0i, cv0, acci[t]);
                            accr[t] = fmaf(y1r, cv1, accr[t]);
                            acci[t] = fmaf(y1i, cv1, acci[t]);
                        }
                }
            else
                {
                    const float nf0 = (float)n0, nf1 = (float)n1;
                    const float s0 = step * nf0, s1 = step * nf1;
#pragma unroll
                    for (int t = 0; t < NTAPS; t++)
                        {
                            int i0 = (int)floorf((s0 + shifts[t]) - rem);
                            int i1 = (int)floorf((s1 + shifts[t]) - rem);
                            float cv0, cv1;
                            if (windowed)
                                {
                                    cv0 = table[i0 - lo];
                                    cv1 = table[i1 - lo];
                                }
                            else
                                {
                                    cv0 = table[posmod(i0, L)];
                                    cv1 = table[posmod(i1, L)];
                                }
                            accr[t] = fmaf(y0r, cv0, accr[t]);
                            acci[t] = fmaf(y0i, cv0, acci[t]);
                            accr[t] = fmaf(y1r, cv1, accr[t]);
                            acci[t] = fmaf(y1i, cv1, acci[t]);
                        }
                }

            if (!HDC)
                {
                    // advance both rotators by one chunk
                    float t0 = fmaf(z0r, wr, -(z0i * wi));
                    z0i = fmaf(z0r, wi, z0i * wr);
                    z0r = t0;
                    float t1 = fmaf(z1r, wr, -(z1i * wi));
                    z1i = fmaf(z1r, wi, z1i * wr);
                    z1r = t1;
                }
        }

    // ---- reduction: lanes -> wave (shuffles) -> workgroup (LDS) ----
    __syncthreads();  // the header doubles have been consumed by every thread
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int t = 0; t < NTAPS; t++)
        {
            float sr = wave_sum(accr[t]);
            float si = wave_sum(acci[t]);
            if (lane == 0)
                {
                    lds[(wave * NTAPS + t) * 2 + 0] = sr;
                    lds[(wave * NTAPS + t) * 2 + 1] = si;
                }
        }
    __syncthreads();
    if (tid < NTAPS)
        {
            float sr = 0.f, si = 0.f;
#pragma unroll
            for (int w = 0; w < TRK_THREADS / 64; w++)
                {
                    sr += lds[(w * NTAPS + tid) * 2 + 0];
                    si += lds[(w * NTAPS + tid) * 2 + 1];
                }
            if (n_slices == 1)
                out[(size_t)job * NTAPS + tid] = make_float2(sr, si);
            else
                partial[((size_t)job * n_slices + slice) * NTAPS + tid] = make_float2(sr, si);
        }
}

// sums the per-slice partials in slice order (deterministic)
__global__ void trk_finish_kernel(const float2* __restrict__ partial, float2* __restrict__ out,
    int n_items /* jobs*n_taps */, int n_taps, int n_slices)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    int job = i / n_taps, t = i % n_taps;
    float sr = 0.f, si = 0.f;
    for (int s = 0; s < n_slices; s++)
        {
            float2 v = partial[((size_t)job * n_slices + s) * n_taps + t];
            sr += v.x;
            si += v.y;
        }
    out[i] = make_float2(sr, si);
}

// -----------------------------------------------------------------------------
// launcher
// -----------------------------------------------------------------------------
template <int NTAPS>
static hipError_t launch_ntaps(int mode, dim3 grid, size_t lds_bytes, hipStream_t st,
    const TrkChan* chans, const gc_epoch_params* params, float2* out, float2* partial,
    int n_channels, int n_epochs, int n_slices, int lds_table_floats)
{
    switch (mode)
        {
        case TRK_MODE_PLAIN:
            hipLaunchKernelGGL((trk_multicorrelator_kernel<NTAPS, false, false>), grid, dim3(TRK_THREADS), lds_bytes, st,
                chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats);
            break;
        case TRK_MODE_HD_RESAMPLER:
            hipLaunchKernelGGL((trk_multicorrelator_kernel<NTAPS, true, false>), grid, dim3(TRK_THREADS), lds_bytes, st,
                chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats);
            break;
        case TRK_MODE_HD_FULL:
            hipLaunchKernelGGL((trk_multicorrelator_kernel<NTAPS, true, true>), grid, dim3(TRK_THREADS), lds_bytes, st,
                chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats);
            break;
        default:
            return hipErrorInvalidValue;
        }
    return hipGetLastError();
}

hipError_t trk_launch(int n_taps, int mode, hipStream_t st, const TrkChan* chans,
    const gc_epoch_params* params, float2* out, float2* partial,
    int n_channels, int n_epochs, int n_slices, int lds_table_floats)
{
    const int epochs8 = (n_epochs + 7) / 8 * 8;
    dim3 grid((unsigned)((size_t)epochs8 * n_channels * n_slices));
    size_t lds_bytes = (size_t)(TRK_HDR_FLOATS + lds_table_floats) * sizeof(float);
    hipError_t e;
#define CASE(NT)                                                                                       \
    case NT:                                                                                           \
        e = launch_ntaps<NT>(mode, grid, lds_bytes, st, chans, params, out, partial, n_channels, n_epochs, \
            n_slices, lds_table_floats);                                                               \
        break;
    switch (n_taps)
        {
            CASE(1)
            CASE(2)
            CASE(3)
            CASE(4)
            CASE(5)
            CASE(6)
            CASE(7)
            CASE(8)
        default:
            return hipErrorInvalidValue;
        }
#undef CASE
    if (e != hipSuccess) return e;
    if (n_slices > 1)
        {
            int n_items = n_channels * n_epochs * n_taps;
            hipLaunchKernelGGL(trk_finish_kernel, dim3((n_items + 255) / 256), dim3(256), 0, st,
                partial, out, n_items, n_taps, n_slices);
            e = hipGetLastError();
        }
    return e;
}
