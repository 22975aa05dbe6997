// acq_kernels.h -- device-side pieces of the PCPS acquisition engine.
#ifndef ACQ_KERNELS_H
#define ACQ_KERNELS_H
#include "gnsscorr.h"
#include <hip/hip_runtime.h>
#include "acq_phase_segments.h"

#define ACQ_MAX_FACTORS 12
#define ACQ_MAX_N1 64

// N-point FFT as N = N1 x N2: N2-point row FFTs in LDS, N1-point column DFTs in registers
struct AcqFftPlan
{
    int N, N1, N2;
    int n_fac;
    int fac[ACQ_MAX_FACTORS];  // radices of the N2-point row FFT, product = N2
    int tw_off[ACQ_MAX_FACTORS];  // offset of each stage's twiddle table inside the stage-twiddle array
    float2 w1[ACQ_MAX_N1];     // exp(-2*pi*j*k/N1), k < N1
};

// which array slice a cell reads: index = (cell / div) % mod
struct AcqCellMap
{
    int div, mod;
};

enum
{
    ACQ_EPI_COMPLEX = 0,  // store the transform (natural order)
    ACQ_EPI_COMPLEX_CONJ_PERM = 1,  // store conj(transform) in row-permuted order (code FFT)
    ACQ_EPI_PERM = 2,     // store the transform in row-permuted order (signal FFT)
    ACQ_EPI_MAG = 3,      // |.|^2 stored into the search grid (+ per-block row maxima): first dwell
    ACQ_EPI_MAG2 = 4,     // two dwells at once: Q holds [sat][2 * n_bins][N], the second n_bins being the next dwell; the grid gets
                          // |.|^2 of the first + |.|^2 of the second, written once
    ACQ_EPI_MAG2_ACC = 5, // the same on top of earlier dwells: (grid + first) + second
    ACQ_EPI_MAG_ACC = 6   // grid += |.|^2: later dwells
};

struct AcqMagArgs
{
    float* grid;          // [cell][fft_size]
    float* tmp;           // per-satellite d_tmp_buffer image [sat][fft_size] (may be null)
    float* blk_max_val;   // [cell][n_blocks]
    unsigned* blk_max_idx;
    int offset;           // first kept output sample (bit_transition: fft_size/2)
    int eff;              // kept samples per row
    int n_bins;           // cells per satellite
    int tmp_bin;          // bin whose single-dwell magnitudes are mirrored into tmp (accumulate only)
};

bool acq_plan_make(AcqFftPlan* plan, int N, size_t lds_limit_bytes);
size_t acq_rows_lds_bytes(const AcqFftPlan& plan);
// per-stage twiddles of the N2-point row FFT: stage f at out[tw_off[f] + (k-1)*m + q] = exp(-2*pi*j*q*k/n_f)
// (N2 - 1 entries in total), followed by the plain table exp(-2*pi*j*i/N2), i < N2 (generic radices)
void acq_stage_twiddles(const AcqFftPlan& plan, float2* out /* 2*N2 entries */);

// out[a*N2 + b] = in[a + N1*b] (* mul[a + N1*b]); in is zero beyond n_valid
hipError_t acq_launch_permute(hipStream_t st, const float2* in, const float2* mul, float2* out,
    const AcqFftPlan& plan, int n_valid, int n_arrays, size_t in_stride, size_t mul_stride, size_t out_stride);

// rows pass: Q[cell][k1][n2] = twiddle * FFT_N2(A[mapA(cell)][k1][.] * B[mapB(cell)][k1][.])
hipError_t acq_launch_rows(hipStream_t st, bool inverse, const AcqFftPlan& plan, int n_cells,
    const float2* A, AcqCellMap mapA, const float2* B, AcqCellMap mapB,
    float2* Q, const float2* wN2, const float2* wN);

// columns pass + epilogue
hipError_t acq_launch_cols(hipStream_t st, bool inverse, int epilogue, const AcqFftPlan& plan, int n_cells,
    const float2* Q, float2* out, const AcqMagArgs* mag);

int acq_cols_blocks(const AcqFftPlan& plan);

// The inverse row pass of one batch (arguments as acq_launch_rows, inverse) and the two-dwell column pass (epilogue ACQ_EPI_MAG2 /
// _MAG2_ACC, n_cells_cols grid cells read from Qc) of the previous batch in ONE launch; only for plans acq_rows_cols_fusable() accepts
// (N = 25 x 1000, the planar pair row kernel)
bool acq_rows_cols_fusable(const AcqFftPlan& plan);
hipError_t acq_launch_rows_cols(hipStream_t st, const AcqFftPlan& plan, int n_cells, const float2* A, AcqCellMap mapA, const float2* B,
    AcqCellMap mapB, float2* Q, const float2* wN2, const float2* wN, int epilogue, int n_cells_cols, const float2* Qc, const AcqMagArgs& mag);

// The whole inverse transform of every cell on chip (no inter-pass buffer), for plans acq_inv_fusable() accepts (N = 25 x 1000):
// grid cell (sat, bin) <- |IFFT(A[bin] * B[sat])|^2 (n_tr = 1), or that of A[bin] and then of A[n_bins + bin] on top (n_tr = 2: a
// dwell pair); `accumulate`: the grid holds earlier dwells and the first transform adds to them as well.  Same grid, scratch image
// and block maxima as acq_launch_rows + acq_launch_cols with the MAG / MAG_ACC epilogues.
bool acq_inv_fusable(const AcqFftPlan& plan);
hipError_t acq_launch_inv_fused(hipStream_t st, const AcqFftPlan& plan, int n_sats, int n_bins, int n_tr, bool accumulate, const float2* A, const float2* B,
    const float2* wN2, const float2* wN, const AcqMagArgs& mag, int n_cus);

// out[bin][(n % N1) * N2 + n / N1] = (cos, sin) of the float32 running phase of sample n (volk_gnsssdr_s32f_sincos_32fc): the table in the
// row-permuted layout the forward row pass reads, in one kernel from the rows' arithmetic-progression segments (acq_phase_segments.h):
// seg_off[bin] .. seg_off[bin + 1] index `segs`
hipError_t acq_launch_wipeoff_segments(hipStream_t st, const AcqPhaseSeg* segs, const int* seg_off, float2* out, int n_bins, const AcqFftPlan& plan);

// cshort / cbyte input block -> float complex (n samples)
hipError_t acq_launch_convert(hipStream_t st, int iq_format, const void* in, float2* out, int n);

hipError_t acq_launch_input_power(hipStream_t st, const float2* x, int n_valid, int N, float* out_power, float* tmp_all,
    int n_sats, size_t tmp_stride);

struct AcqFinalArgs
{
    const float* grid;
    float* tmp;  // [sat][fft_size]
    const float* blk_max_val;
    const unsigned* blk_max_idx;
    const float* input_power;
    gc_acq_result* results;
    float* part_val;     // [sat][ACQ_FINAL_PIECES][2]: second-peak candidates of the row pieces
    unsigned* part_cnt;  // [sat]: ticket counters, zero between launches
    int n_bins, n_blocks, fft_size;
    int doppler_max, doppler_step;
    int use_cfar;
    int samples_per_chip;
    float samples_per_code;
    int step_two;       // Doppler of a row follows the step-two formula (pcps_acquisition.cc:589-591)
    float center_step_two, doppler_step2;
    int n_bins_step2;
};
#define ACQ_FINAL_PIECES 8  // = ACQ_FINAL_SPLIT of acq_kernels.hip
hipError_t acq_launch_final(hipStream_t st, const AcqFinalArgs& a, int n_sats);

#endif
