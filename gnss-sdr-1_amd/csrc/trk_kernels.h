// trk_kernels.h -- device-side descriptors and launcher of the tracking kernel.
#ifndef TRK_KERNELS_H
#define TRK_KERNELS_H
#include "gnsscorr.h"
#include <hip/hip_runtime.h>

#define GC_MAX_TAPS 8

// per-channel descriptor (device memory)
struct TrkChan
{
    const void* iq;            // IQ base of the channel's RF stream (HBM), samples in the batch's gc_iq_format
    unsigned long long n_iq;   // samples available at iq
    const float* code;         // code table (HBM): code_len floats, or code_len (re, im) pairs in TRK_MODE_COMPLEX_CODE
    int code_len;
    unsigned ring_len;         // 0: sample_offset is relative to iq; else iq is a gc_stream ring of ring_len samples and
                               // sample_offset an absolute sample number (the window is contiguous thanks to the mirror)
    float shifts[GC_MAX_TAPS]; // tap shifts in code samples
    const float* code2;        // closed loop, pilot tracking: replica of the data component (prompt-only correlator); else NULL
};

enum
{
    TRK_MODE_PLAIN = 0,        // resampler_32f_xn + rotator_dot_prod_32fc_xn
    TRK_MODE_HD_RESAMPLER = 1, // high-dynamics resampler + plain rotator (6-argument overload with the flag set)
    TRK_MODE_HD_FULL = 2,      // high-dynamics resampler + high-dynamic rotator
    TRK_MODE_COMPLEX_CODE = 3, // Cpu_Multicorrelator: resampler_32fc_xn + x2_rotator_dot_prod_32fc_xn (complex chips)
    TRK_MODE_SC16 = 4          // Cpu_Multicorrelator_16sc: 16ic_xn_resampler_16ic_xn + 16ic_x2_rotator_dot_prod_16ic_xn
                               // (GC_IQ_I16 input, one (re16, im16) word per chip, `out` receives n_taps short2)
};

// Enqueues the multicorrelator for n_channels x n_epochs jobs on `st`.
// lds_table_floats: capacity of the LDS code window in floats (>= longest code_len; twice that for complex chips).
// partial: workspace of n_channels*n_epochs*n_slices*n_taps float2 (n_slices > 1 only).
// line_aligned: the chunks of a window are counted from the 128-byte boundary below its first sample (whole cache lines per wave
// instruction: the streaming launches) instead of the 16-byte one (the level-1 calls, whose staging keeps a window's 16-byte phase so
// that a call gives the same bits alone and in a batch).
bool trk_small_window_ok(int mode, int iq_format);
hipError_t trk_launch(int n_taps, int mode, int iq_format, hipStream_t st, const TrkChan* chans,
    const gc_epoch_params* params, float2* out, float2* partial,
    int n_channels, int n_epochs, int n_slices, int lds_table_floats, bool line_aligned);

#endif
