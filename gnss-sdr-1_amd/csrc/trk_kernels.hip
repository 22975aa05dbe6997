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
#include "trk_device.hpp"

#ifndef TRK_WAVES
#define TRK_WAVES 8  // minimum waves per SIMD the register allocator must leave room for (<= 64 VGPRs)
#endif
static __device__ __forceinline__ short sat16(int v) { return (short)min(max(v, -32768), 32767); }

#ifndef TRK_CHIPS_WAVES
#define TRK_CHIPS_WAVES 4  // the chip-domain loop holds a lane's eight samples and the next segment's loads
#endif
template <int NTAPS, bool HDR, bool HDC, int FMT, bool CC = false, bool SC16 = false, bool CHIPS = false>
__global__ __launch_bounds__(TRK_THREADS, CHIPS ? TRK_CHIPS_WAVES : TRK_WAVES) void trk_multicorrelator_kernel(
    const TrkChan* __restrict__ chans, const gc_epoch_params* __restrict__ params,
    float2* __restrict__ out, float2* __restrict__ partial,
    int n_channels, int n_epochs, int n_slices, int lds_table_floats, int align_pairs)
{
    extern __shared__ float lds[];

    // XCD-aware job mapping: workgroups are dealt round-robin over the 8 XCDs, so
    // blocks with equal (blockIdx % 8) share an L2.  All channels of one epoch
    // (which read the same IQ window when they share an RF stream) are given the
    // same residue.
    const int b = blockIdx.x;
    int slice, ch, epoch;
    if (n_epochs == 1)
        {
            // one epoch per channel (level-1 calls and their batches): no epoch dimension to spread over the XCDs
            slice = b % n_slices;
            ch = b / n_slices;
            epoch = 0;
        }
    else
        {
            const int x = b & 7;
            int q = b >> 3;
            slice = q % n_slices;
            q /= n_slices;
            ch = q % n_channels;
            epoch = (q / n_channels) * 8 + x;
            if (epoch >= n_epochs) return;
        }
    const int job = ch * n_epochs + epoch;

    const TrkChan cd = chans[ch];
    const gc_epoch_params p = params[job];
    const float2 r = trk_epoch<NTAPS, HDR, HDC, FMT, CC, SC16, TRK_THREADS, false, TRK_PF, CHIPS, (TRK_NT != 0), true, !CHIPS>(cd, p, slice, n_slices, lds_table_floats, lds, align_pairs);
    if (threadIdx.x < NTAPS)
        {
            if (n_slices == 1)
                {
                    if (SC16)  // lv_16sc_t results: the exact integer sums, saturated once
                        reinterpret_cast<short2*>(out)[(size_t)job * NTAPS + threadIdx.x] = make_short2(sat16(__float_as_int(r.x)), sat16(__float_as_int(r.y)));
                    else
                        out[(size_t)job * NTAPS + threadIdx.x] = r;
                }
            else
                partial[((size_t)job * n_slices + slice) * NTAPS + threadIdx.x] = r;
        }
}

// sums the per-slice partials in slice order (deterministic)
__global__ void trk_finish_kernel(const float2* __restrict__ partial, float2* __restrict__ out,
    int n_items /* jobs*n_taps */, int n_taps, int n_slices, int sc16)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    int job = i / n_taps, t = i % n_taps;
    if (sc16)
        {
            int sr = 0, si = 0;
            for (int s = 0; s < n_slices; s++)
                {
                    float2 v = partial[((size_t)job * n_slices + s) * n_taps + t];
                    sr += __float_as_int(v.x);
                    si += __float_as_int(v.y);
                }
            reinterpret_cast<short2*>(out)[i] = make_short2(sat16(sr), sat16(si));
            return;
        }
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
// Experiments build only: $GNSSCORR_TRK_LOOP = samples | chips selects the form of the plain float loop.  The chip-domain form
// (trk_chips.hpp) gives the same results and measured slower on MI355X in every mode of the bench
#ifdef GNSSCORR_EXPERIMENTS
static bool trk_chip_domain()
{
    static const bool on = [] {
        const char* e = gc_exp_env("GNSSCORR_TRK_LOOP");
        return e && e[0] == 'c';
    }();
    return on;
}
static size_t trk_chips_lds_bytes(int lds_table_floats)
{
    return (size_t)(TRK_HDR_FLOATS + ((lds_table_floats + 3) & ~3) + TRK_THREADS / 64 * TRK_CHIPS_WAVE_FLOATS) * sizeof(float);
}
#endif
// whether trk_launch's kernel for (mode, format) serves records that fit neither the launch's LDS window nor the whole table in it
// (the global-table path of trk_epoch): the host sizes a launch's LDS below the table only then
bool trk_small_window_ok(int mode, int iq_format)
{
#ifdef GNSSCORR_EXPERIMENTS
    if (mode == TRK_MODE_PLAIN && iq_format == GC_IQ_F32 && trk_chip_domain()) return false;  // the chip-domain loop keeps the whole table in LDS
#endif
    (void)mode;
    (void)iq_format;
    return true;
}

template <int NTAPS, int FMT>
static hipError_t launch_ntaps_fmt(int mode, dim3 grid, size_t lds_bytes, hipStream_t st,
    const TrkChan* chans, const gc_epoch_params* params, float2* out, float2* partial,
    int n_channels, int n_epochs, int n_slices, int lds_table_floats, int align_pairs)
{
    switch (mode)
        {
        case TRK_MODE_PLAIN:
#ifdef GNSSCORR_EXPERIMENTS
            if (FMT == GC_IQ_F32 && trk_chip_domain() && trk_chips_lds_bytes(lds_table_floats) <= 64 * 1024)
                {
                    hipLaunchKernelGGL((trk_multicorrelator_kernel<NTAPS, false, false, GC_IQ_F32, false, false, true>), grid, dim3(TRK_THREADS),
                        trk_chips_lds_bytes(lds_table_floats), st, chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats, align_pairs);
                    break;
                }
#endif
            hipLaunchKernelGGL((trk_multicorrelator_kernel<NTAPS, false, false, FMT>), grid, dim3(TRK_THREADS), lds_bytes, st,
                chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats, align_pairs);
            break;
        case TRK_MODE_HD_RESAMPLER:
            hipLaunchKernelGGL((trk_multicorrelator_kernel<NTAPS, true, false, FMT>), grid, dim3(TRK_THREADS), lds_bytes, st,
                chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats, align_pairs);
            break;
        case TRK_MODE_HD_FULL:
            hipLaunchKernelGGL((trk_multicorrelator_kernel<NTAPS, true, true, FMT>), grid, dim3(TRK_THREADS), lds_bytes, st,
                chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats, align_pairs);
            break;
        case TRK_MODE_COMPLEX_CODE:
            hipLaunchKernelGGL((trk_multicorrelator_kernel<NTAPS, false, false, FMT, true>), grid, dim3(TRK_THREADS), lds_bytes, st,
                chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats, align_pairs);
            break;
        case TRK_MODE_SC16:
            if (FMT != GC_IQ_I16) return hipErrorInvalidValue;  // lv_16sc_t in, lv_16sc_t out
            hipLaunchKernelGGL((trk_multicorrelator_kernel<NTAPS, false, false, GC_IQ_I16, false, true>), grid, dim3(TRK_THREADS), lds_bytes, st,
                chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats, align_pairs);
            break;
        default:
            return hipErrorInvalidValue;
        }
    return hipGetLastError();
}

template <int NTAPS>
static hipError_t launch_ntaps(int mode, int fmt, dim3 grid, size_t lds_bytes, hipStream_t st,
    const TrkChan* chans, const gc_epoch_params* params, float2* out, float2* partial,
    int n_channels, int n_epochs, int n_slices, int lds_table_floats, int align_pairs)
{
    switch (fmt)
        {
        case GC_IQ_F32:
            return launch_ntaps_fmt<NTAPS, GC_IQ_F32>(mode, grid, lds_bytes, st, chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats, align_pairs);
        case GC_IQ_I16:
            return launch_ntaps_fmt<NTAPS, GC_IQ_I16>(mode, grid, lds_bytes, st, chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats, align_pairs);
        case GC_IQ_I8:
            return launch_ntaps_fmt<NTAPS, GC_IQ_I8>(mode, grid, lds_bytes, st, chans, params, out, partial, n_channels, n_epochs, n_slices, lds_table_floats, align_pairs);
        default:
            return hipErrorInvalidValue;
        }
}

hipError_t trk_launch(int n_taps, int mode, int iq_format, hipStream_t st, const TrkChan* chans,
    const gc_epoch_params* params, float2* out, float2* partial,
    int n_channels, int n_epochs, int n_slices, int lds_table_floats, bool line_aligned)
{
    const int align_pairs = line_aligned ? TRK_ALIGN_PAIRS : 1;
    const int epochs8 = n_epochs == 1 ? 1 : (n_epochs + 7) / 8 * 8;
    dim3 grid((unsigned)((size_t)epochs8 * n_channels * n_slices));
    size_t lds_bytes = (size_t)(TRK_HDR_FLOATS + lds_table_floats) * sizeof(float);
    hipError_t e;
#define CASE(NT)                                                                                       \
    case NT:                                                                                           \
        e = launch_ntaps<NT>(mode, iq_format, grid, lds_bytes, st, chans, params, out, partial, n_channels, n_epochs, \
            n_slices, lds_table_floats, align_pairs);                                                  \
        break;
    switch (n_taps)
        {
#ifdef TRK_DEV_BUILD  // development builds: the bench's tap counts only (a full build takes minutes)
            CASE(3)
            CASE(5)
#else
            CASE(1)
            CASE(2)
            CASE(3)
            CASE(4)
            CASE(5)
            CASE(6)
            CASE(7)
            CASE(8)
#endif
        default:
            return hipErrorInvalidValue;
        }
#undef CASE
    if (e != hipSuccess) return e;
    if (n_slices > 1)
        {
            int n_items = n_channels * n_epochs * n_taps;
            hipLaunchKernelGGL(trk_finish_kernel, dim3((n_items + 255) / 256), dim3(256), 0, st,
                partial, out, n_items, n_taps, n_slices, mode == TRK_MODE_SC16 ? 1 : 0);
            e = hipGetLastError();
        }
    return e;
}
