// acq_kernels.hip -- PCPS acquisition kernels for gfx950 (MI355X).
//
// Reference path: pcps_acquisition::acquisition_core
// (src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.cc:668-770):
//   per Doppler bin  x*wipeoff -> FFT -> * conj(FFT(code)) -> IFFT -> |.|^2 -> grid (+=)
// followed by max_to_input_power_statistic (:565-596) or
// first_vs_second_peak_statistic (:599-665).
//
// FFT sizes are one code period of samples (4000, 25000, 100000 ...), not powers
// of two.  An N-point transform is split N = N1 x N2 (Cooley-Tukey "four step"):
//   rows    : N1 independent N2-point FFTs, each in LDS (Stockham autosort,
//             radix 8/5/4/3/2 + generic prime), one 256-thread workgroup per row
//   twiddle : w_N^(k1*n2), applied when the row leaves LDS
//   columns : N2 independent N1-point DFTs, each in the registers of one thread
// Rows consume the "row-permuted" layout P[a][b] = v[a + N1*b] and columns
// produce natural order, so every global access is a coalesced row.  The
// element-wise product with conj(FFT(code)) is fused into the row load and
// |.|^2 + non-coherent accumulation + row maximum into the column epilogue.
#include "acq_kernels.h"
#include "gc_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ACQ_THREADS 256

#ifndef ACQ_PK_CMUL
#define ACQ_PK_CMUL 0  // 1: complex multiplies as two hand-written packed-FP32 instructions.  Measured: the compiler already emits the same
                      // v_pk_mul_f32 + v_pk_fma_f32 pair from the plain form (identical counts in the row kernel), and the inline-asm version
                      // schedules worse (0.55 instead of 0.50 ms per search): kept as a knob
#endif
typedef float acq_f32x2 __attribute__((ext_vector_type(2)));
typedef float acq_f32x4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
#if ACQ_PK_CMUL
    // t = (-ay*by, ay*bx); r = (ax*bx + t.lo, ax*by + t.hi): the operand halves are picked with op_sel / op_sel_hi, the sign with neg_lo
    acq_f32x2 va = {a.x, a.y}, vb = {b.x, b.y}, t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(va), "v"(vb));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));
    return make_float2(r.x, r.y);
#else
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
#endif
}
static __device__ __forceinline__ float2 cmul_conj(float2 a, float2 b)  // a * conj(b)
{
#if ACQ_PK_CMUL
    // t = (ay*by, ay*bx); r = (ax*bx + t.lo, -ax*by + t.hi)
    acq_f32x2 va = {a.x, a.y}, vb = {b.x, b.y}, t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(va), "v"(vb));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));
    return make_float2(r.x, r.y);
#else
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y)));
#endif
}
static __device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
static __device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -j (forward) or +j (inverse)
template <bool INV>
static __device__ __forceinline__ float2 mul_mj(float2 a)
{
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// ---- small DFTs on registers (forward kernel exp(-j...), INV -> exp(+j...)) ----
template <bool INV>
static __device__ __forceinline__ void dft2(float2* a)
{
    float2 t = a[0];
    a[0] = cadd(t, a[1]);
    a[1] = csub(t, a[1]);
}
template <bool INV>
static __device__ __forceinline__ void dft3(float2* a)
{
    const float s60 = 0.86602540378443864676f;
    float2 t = cadd(a[1], a[2]);
    float2 d = csub(a[1], a[2]);
    float2 m = make_float2(fmaf(-0.5f, t.x, a[0].x), fmaf(-0.5f, t.y, a[0].y));
    float2 s = mul_mj<INV>(make_float2(s60 * d.x, s60 * d.y));
    a[0] = cadd(a[0], t);
    a[1] = cadd(m, s);
    a[2] = csub(m, s);
}
template <bool INV>
static __device__ __forceinline__ void dft4(float2* a)
{
    float2 t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]);
    float2 t2 = cadd(a[1], a[3]), t3 = mul_mj<INV>(csub(a[1], a[3]));
    a[0] = cadd(t0, t2);
    a[2] = csub(t0, t2);
    a[1] = cadd(t1, t3);
    a[3] = csub(t1, t3);
}
template <bool INV>
static __device__ __forceinline__ void dft5(float2* a)
{
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    float2 t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]);
    float2 t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
    float2 m1 = make_float2(a[0].x + c1 * t1.x + c2 * t2.x, a[0].y + c1 * t1.y + c2 * t2.y);
    float2 m2 = make_float2(a[0].x + c2 * t1.x + c1 * t2.x, a[0].y + c2 * t1.y + c1 * t2.y);
    float2 u1 = mul_mj<INV>(make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
    float2 u2 = mul_mj<INV>(make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
    a[0] = make_float2(a[0].x + t1.x + t2.x, a[0].y + t1.y + t2.y);
    a[1] = cadd(m1, u1);
    a[4] = csub(m1, u1);
    a[2] = cadd(m2, u2);
    a[3] = csub(m2, u2);
}
template <bool INV>
static __device__ __forceinline__ void dft8(float2* a)
{
    const float h = 0.70710678118654752440f;
    float2 e[4] = {a[0], a[2], a[4], a[6]};
    float2 o[4] = {a[1], a[3], a[5], a[7]};
    dft4<INV>(e);
    dft4<INV>(o);
    // w8^k * o[k]
    float2 o1 = INV ? make_float2(h * (o[1].x - o[1].y), h * (o[1].x + o[1].y))
                    : make_float2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
    float2 o2 = mul_mj<INV>(o[2]);
    float2 o3 = INV ? make_float2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y))
                    : make_float2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
    a[0] = cadd(e[0], o[0]);
    a[4] = csub(e[0], o[0]);
    a[1] = cadd(e[1], o1);
    a[5] = csub(e[1], o1);
    a[2] = cadd(e[2], o2);
    a[6] = csub(e[2], o2);
    a[3] = cadd(e[3], o3);
    a[7] = csub(e[3], o3);
}

// a * (c -+ j*s): multiplication by the constant exp(-+j*phi), c = cos(phi), s = sin(phi)
template <bool INV>
static __device__ __forceinline__ float2 cmul_const(float2 a, float c, float sn)
{
    return INV ? make_float2(fmaf(a.x, c, -(a.y * sn)), fmaf(a.y, c, a.x * sn))
               : make_float2(fmaf(a.x, c, a.y * sn), fmaf(a.y, c, -(a.x * sn)));
}
template <bool INV>
static __device__ __forceinline__ void dft10(float2* a)
{
    // 10 = 2 x 5 (decimation in time): X[k] = E[k] + w10^k O[k], X[k+5] = E[k] - w10^k O[k]
    float2 e[5] = {a[0], a[2], a[4], a[6], a[8]};
    float2 o[5] = {a[1], a[3], a[5], a[7], a[9]};
    dft5<INV>(e);
    dft5<INV>(o);
    o[1] = cmul_const<INV>(o[1], 0.80901699437494742410f, 0.58778525229247312917f);
    o[2] = cmul_const<INV>(o[2], 0.30901699437494742410f, 0.95105651629515357212f);
    o[3] = cmul_const<INV>(o[3], -0.30901699437494742410f, 0.95105651629515357212f);
    o[4] = cmul_const<INV>(o[4], -0.80901699437494742410f, 0.58778525229247312917f);
#pragma unroll
    for (int k = 0; k < 5; k++)
        {
            a[k] = cadd(e[k], o[k]);
            a[k + 5] = csub(e[k], o[k]);
        }
}
template <bool INV>
static __device__ __forceinline__ void dft16(float2* a)
{
    // 16 = 4 x 4: F_r = DFT4 of x[4*n1 + r]; X[k1 + 4*k2] = DFT4 over r of w16^(r*k1) F_r[k1]
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    float2 f[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
        {
            f[r][0] = a[r];
            f[r][1] = a[4 + r];
            f[r][2] = a[8 + r];
            f[r][3] = a[12 + r];
            dft4<INV>(f[r]);
        }
    // k1 = 0: no twiddles
    {
        float2 g[4] = {f[0][0], f[1][0], f[2][0], f[3][0]};
        dft4<INV>(g);
        a[0] = g[0];
        a[4] = g[1];
        a[8] = g[2];
        a[12] = g[3];
    }
    {
        float2 g[4] = {f[0][1], cmul_const<INV>(f[1][1], c1, s1), cmul_const<INV>(f[2][1], h, h), cmul_const<INV>(f[3][1], s1, c1)};
        dft4<INV>(g);
        a[1] = g[0];
        a[5] = g[1];
        a[9] = g[2];
        a[13] = g[3];
    }
    {
        float2 g[4] = {f[0][2], cmul_const<INV>(f[1][2], h, h), mul_mj<INV>(f[2][2]), cmul_const<INV>(f[3][2], -h, h)};
        dft4<INV>(g);
        a[2] = g[0];
        a[6] = g[1];
        a[10] = g[2];
        a[14] = g[3];
    }
    {
        float2 g[4] = {f[0][3], cmul_const<INV>(f[1][3], s1, c1), cmul_const<INV>(f[2][3], -h, h), cmul_const<INV>(f[3][3], -c1, -s1)};
        dft4<INV>(g);
        a[3] = g[0];
        a[7] = g[1];
        a[11] = g[2];
        a[15] = g[3];
    }
}

template <int R, bool INV>
static __device__ __forceinline__ void dftR(float2* a)
{
    if (R == 2) dft2<INV>(a);
    if (R == 3) dft3<INV>(a);
    if (R == 4) dft4<INV>(a);
    if (R == 5) dft5<INV>(a);
    if (R == 8) dft8<INV>(a);
    if (R == 10) dft10<INV>(a);
    if (R == 16) dft16<INV>(a);
}

// ---- one Stockham stage of the LDS row FFT ----
// current length n = R*m, stride s (product of the radices already done), N2 = n*s.
//   y[r + s*(R*q + k)] = w_n^(q*k) * sum_j x[r + s*(q + m*j)] * w_R^(j*k)
// twf: this stage's twiddles in global memory laid out [k-1][q] (unit stride across lanes; the
// table is a few KB shared by every workgroup, so it lives in L1/L2 and costs no LDS)
template <int R, bool INV>
static __device__ __forceinline__ void lds_stage(const float2* __restrict__ x, float2* __restrict__ y,
    const float2* __restrict__ twf, int N2, int m, int s)
{
    const int nb = N2 / R;
    for (int u = threadIdx.x; u < nb; u += blockDim.x)
        {
            const int r = u % s, q = u / s;
            float2 w[R];
#pragma unroll
            for (int k = 1; k < R; k++) w[k] = twf[(k - 1) * m + q];
            float2 a[R];
#pragma unroll
            for (int j = 0; j < R; j++) a[j] = x[r + s * (q + m * j)];
            dftR<R, INV>(a);
            const int base = r + s * R * q;
            y[base] = a[0];
#pragma unroll
            for (int k = 1; k < R; k++) y[base + s * k] = INV ? cmul_conj(a[k], w[k]) : cmul(a[k], w[k]);
        }
}

// generic (prime) radix: O(R^2) butterfly; twp = plain table exp(-2*pi*j*i/N2) in global memory
template <bool INV>
static __device__ void lds_stage_generic(const float2* __restrict__ x, float2* __restrict__ y,
    const float2* __restrict__ twp, int N2, int R, int m, int s)
{
    const int nb = N2 / R;
    const int wR = N2 / R;  // w_R^e = w_N2^(e*N2/R)
    for (int u = threadIdx.x; u < nb; u += blockDim.x)
        {
            const int r = u % s, q = u / s;
            for (int k = 0; k < R; k++)
                {
                    float2 acc = make_float2(0.f, 0.f);
                    for (int j = 0; j < R; j++)
                        {
                            float2 v = x[r + s * (q + m * j)];
                            float2 w = twp[((j * k) % R) * wR];
                            float2 p = INV ? cmul_conj(v, w) : cmul(v, w);
                            acc = cadd(acc, p);
                        }
                    float2 w = twp[q * k * s];
                    y[r + s * (R * q + k)] = INV ? cmul_conj(acc, w) : cmul(acc, w);
                }
        }
}

// ---- rows pass ----
template <bool INV>
__global__ __launch_bounds__(ACQ_THREADS) void acq_rows_kernel(AcqFftPlan plan,
    const float2* __restrict__ A, AcqCellMap mapA, const float2* __restrict__ B, AcqCellMap mapB,
    float2* __restrict__ Q, const float2* __restrict__ wN2, const float2* __restrict__ wN)
{
    extern __shared__ float2 sm[];
    const int N2 = plan.N2, N = plan.N;
    float2* buf0 = sm;
    float2* buf1 = sm + N2;
    const float2* twp = wN2 + N2;  // plain table behind the per-stage tables
    const int k1 = blockIdx.x;
    const int cell = blockIdx.y;
    const size_t row = (size_t)k1 * N2;
    const float2* a = A + (size_t)((cell / mapA.div) % mapA.mod) * N + row;
    if (B)
        {
            const float2* bb = B + (size_t)((cell / mapB.div) % mapB.mod) * N + row;
            for (int i = threadIdx.x; i < N2; i += blockDim.x) buf0[i] = cmul(a[i], bb[i]);
        }
    else
        {
            for (int i = threadIdx.x; i < N2; i += blockDim.x) buf0[i] = a[i];
        }
    __syncthreads();
    float2* src = buf0;
    float2* dst = buf1;
    int n = N2, s = 1;
    for (int f = 0; f < plan.n_fac; f++)
        {
            const int R = plan.fac[f];
            const int m = n / R;
            const float2* twf = wN2 + plan.tw_off[f];
            switch (R)
                {
                case 2: lds_stage<2, INV>(src, dst, twf, N2, m, s); break;
                case 3: lds_stage<3, INV>(src, dst, twf, N2, m, s); break;
                case 4: lds_stage<4, INV>(src, dst, twf, N2, m, s); break;
                case 5: lds_stage<5, INV>(src, dst, twf, N2, m, s); break;
                case 8: lds_stage<8, INV>(src, dst, twf, N2, m, s); break;
                case 10: lds_stage<10, INV>(src, dst, twf, N2, m, s); break;
                case 16: lds_stage<16, INV>(src, dst, twf, N2, m, s); break;
                default: lds_stage_generic<INV>(src, dst, twp, N2, R, m, s); break;
                }
            __syncthreads();
            float2* t = src;
            src = dst;
            dst = t;
            n = m;
            s *= R;
        }
    // leave LDS through the inter-pass twiddle w_N^(k1*n2); the table is stored as the matrix
    // T[k1][n2] so that a row reads it with unit stride
    float2* q = Q + (size_t)cell * N + row;
    const float2* wrow = wN + row;
    for (int i = threadIdx.x; i < N2; i += blockDim.x)
        {
            float2 v = src[i];
            if (k1 > 0)
                {
                    float2 w = wrow[i];
                    v = INV ? cmul_conj(v, w) : cmul(v, w);
                }
            q[i] = v;
        }
}

// ---- rows pass, packed form ----
// Several rows per 256-thread workgroup so that every stage has about 256*ITER butterflies to hand out
// (a single 1000-point row offers 100-200: half of the lanes would idle), radices up to 16 (1000 = 10*10*10:
// three stages instead of four), the first stage fed straight from global memory and the last one
// storing straight to global memory through the inter-pass twiddle (no staging copies), and in-place
// LDS between stages: a thread reads all of its butterflies of a stage into registers, the workgroup
// synchronises, then the outputs overwrite the same LDS rows (half the LDS of a ping-pong pair, so more
// workgroups per CU).  The kernel is a template of its stage list (radix and butterflies per thread of
// every stage): row length, strides and the first/last roles are compile-time constants, and each
// instantiation holds only its own stages, which keeps it at <= 128 VGPRs (one switch over all radices
// in one kernel made the register allocator spill).  acq_rows2_registry lists the instantiated
// stage lists; any other row length runs acq_rows_kernel.
#ifndef ACQ_ROWS2_WAVES
#define ACQ_ROWS2_WAVES 4  // waves per SIMD the register allocator must leave room for
#endif
#define ACQ_ROWS2_POINTS 20       // R * ITER: points a thread holds per stage
#define ACQ_ROWS2_LDS_BYTES 40000 // rows * N2 * 8 per workgroup: four workgroups per CU

struct AcqRows2Args
{
    const float2* A;
    AcqCellMap mapA;
    const float2* B;
    AcqCellMap mapB;
    float2* Q;
    const float2* wN2;
    const float2* wN;
    int n_rows;  // N1 * cells
    int rpw;     // rows per workgroup
    int n_bins;  // cells per satellite (= mapA.mod)
    int n_sats;  // satellites in this launch; cell = sat * n_bins + bin
    int n_groups;  // workgroups that have rows
    int sat_fastest;  // pair kernels: row order, 0 = (bin, sat, k1), 1 = (bin, k1, sat), 2 = (k1, bin, sat; planar kernel only)
    unsigned long long* ts;  // ACQ_ROWS3_DBG builds, $GNSSCORR_ACQ_DBG & 64: six s_memrealtime stamps per workgroup (100 MHz)
    int dbg;          // $GNSSCORR_ACQ_DBG, timing experiments on acq_rows3_kernel only (results are WRONG with any bit set):
                      // 1 = no global input loads, 2 = no global stores, 4 = no twiddle loads, 8 = no butterflies
};

// floor(a / b) for 0 <= a < 2^22 given inv_b = 1.0f / b
static __device__ __forceinline__ int fdiv(int a, float inv_b) { return (int)(((float)a + 0.5f) * inv_b); }

// one Stockham stage over the workgroup's rows: current length n = R*M, stride S, N2 = n*S
template <int R, int ITER, bool INV, int N2, int S, bool FIRST, bool LAST>
static __device__ __forceinline__ void rows2_stage(const AcqFftPlan& plan, const AcqRows2Args& g, float2* lds,
    int row0, int nrow, const float2* __restrict__ twf)
{
    constexpr int NB = N2 / R;       // butterflies per row
    constexpr int M = N2 / (S * R);  // sub-transform length after this stage
    const int N = plan.N, N1 = plan.N1;
    const int total = nrow * NB;
    const float inv_n1 = 1.0f / (float)N1, inv_ns = 1.0f / (float)g.n_sats;
    float2 a[ITER][R];
    int off[ITER];  // where the butterfly's outputs go (LDS index, or offset into Q for the last stage)
    int twi[ITER];  // where its twiddles start
    // ---- phase 1: gather the inputs ----
#pragma unroll
    for (int it = 0; it < ITER; it++)
        {
            const int v = threadIdx.x + it * ACQ_THREADS;
            off[it] = 0;
            twi[it] = 0;
            if (v < total)
                {
                    const int row = v / NB;
                    const int u = v - row * NB;
                    const int q = u / S;
                    const int r = u - q * S;
                    int cell = 0, k1 = 0, bin = 0, sat = 0;
                    if (FIRST || LAST)
                        {
                            // rows are handed out satellite-fastest: (bin, sat, k1) = digits of the row number,
                            // so that a run of workgroups shares few bins of A (see the kernel's block mapping)
                            const int rowid = row0 + row;
                            const int cl = fdiv(rowid, inv_n1);
                            k1 = rowid - cl * N1;
                            bin = fdiv(cl, inv_ns);
                            sat = cl - bin * g.n_sats;
                            cell = sat * g.n_bins + bin;
                        }
                    if (FIRST)
                        {
                            // S == 1: r = 0, q = u; element j of the butterfly is x[q + M*j]
                            const float2* ap = g.A + (size_t)bin * N + (size_t)k1 * N2 + q;
                            if (g.B)
                                {
                                    const float2* bp = g.B + (size_t)sat * N + (size_t)k1 * N2 + q;
#pragma unroll
                                    for (int j = 0; j < R; j++) a[it][j] = cmul(ap[M * j], bp[M * j]);
                                }
                            else
                                {
#pragma unroll
                                    for (int j = 0; j < R; j++) a[it][j] = ap[M * j];
                                }
                        }
                    else
                        {
                            const float2* x = lds + row * N2 + r + S * q;
#pragma unroll
                            for (int j = 0; j < R; j++) a[it][j] = x[S * M * j];
                        }
                    if (LAST)
                        {
                            // M == 1: q = 0; output k goes to n2 = r + S*k of the row
                            off[it] = cell * N + k1 * N2 + r;
                            twi[it] = k1 * N2 + r;
                        }
                    else
                        {
                            off[it] = row * N2 + r + S * R * q;
                            twi[it] = q;
                        }
                }
        }
    if (!FIRST) __syncthreads();  // every input of this stage has left LDS
    // ---- phase 2: butterflies, twiddles, scatter ----
#pragma unroll
    for (int it = 0; it < ITER; it++)
        {
            const int v = threadIdx.x + it * ACQ_THREADS;
            if (v < total)
                {
                    float2 tw[R];
                    if (LAST)
                        {
                            // inter-pass twiddle w_N^(k1*n2) (exactly 1 in row 0)
                            const float2* wp = g.wN + twi[it];
#pragma unroll
                            for (int k = 0; k < R; k++) tw[k] = wp[S * k];
                        }
                    else
                        {
                            tw[0] = make_float2(1.0f, 0.0f);
#pragma unroll
                            for (int k = 1; k < R; k++) tw[k] = twf[(k - 1) * M + twi[it]];
                        }
                    dftR<R, INV>(a[it]);
                    if (LAST)
                        {
                            float2* qp = g.Q + (size_t)off[it];
#pragma unroll
                            for (int k = 0; k < R; k++) qp[S * k] = INV ? cmul_conj(a[it][k], tw[k]) : cmul(a[it][k], tw[k]);
                        }
                    else
                        {
                            float2* y = lds + off[it];
                            y[0] = a[it][0];
#pragma unroll
                            for (int k = 1; k < R; k++) y[S * k] = INV ? cmul_conj(a[it][k], tw[k]) : cmul(a[it][k], tw[k]);
                        }
                }
        }
    if (!LAST) __syncthreads();  // outputs visible to the next stage
}

// ---- the same stage on PAIRS of adjacent butterflies -------------------------------------------------------------
// One thread owns butterflies u = 2p and 2p + 1 of a row.  What that buys (N2 = 1000 = 10 x 10 x 10, the row length of every
// 1 ms block from 2 to 25 Msps): every global access is 16 bytes per lane (first stage: A[q + M j] and A[q + 1 + M j] are
// neighbours; last stage: outputs r + S k and r + 1 + S k are neighbours), every LDS access of the middle stages is a
// ds_read/write_b128, and in the stages with S > 1 the two butterflies share their twiddles (same q).  Twiddles are not
// loaded one by one any more: w^k is built from w^1 by multiplications (k = 2 .. R-1; relative error < 1e-6, against a parity
// bar of 1e-4 of the peak), one or two loads per thread and stage instead of R - 1 (the profile of the plain version showed 116
// vector-memory instructions per thread, most of them 8-byte twiddle loads issued right before their use, and the waves
// waiting on them: SQ_WAIT_INST_ANY 57 % of the wave cycles at 40 % VALU utilisation).
// Conditions (checked at compile time): R even, N2 / R even, first stage M even, other stages S even.
#ifdef GNSSCORR_EXPERIMENTS  // the (re, im)-packed pair kernel: superseded by the planar one below (45.5 vs 43.1 us then), kept for A/B runs
template <int R, bool INV>
static __device__ __forceinline__ void tw_powers(float2 w, float2* tw)
{
    tw[0] = make_float2(1.0f, 0.0f);
    tw[1] = w;
#pragma unroll
    for (int k = 2; k < R; k++) tw[k] = (k % 2 == 0) ? cmul(tw[k / 2], tw[k / 2]) : cmul(tw[k - 1], w);
}
template <bool INV>
static __device__ __forceinline__ float2 tmul(float2 a, float2 w) { return INV ? cmul_conj(a, w) : cmul(a, w); }


template <int R, bool INV, int N2, int S, bool FIRST, bool LAST>
static __device__ __forceinline__ void rows2p_stage(const AcqFftPlan& plan, const AcqRows2Args& g, float2* lds,
    int row0, int nrow, const float2* __restrict__ twf)
{
    constexpr int NB = N2 / R;       // butterflies per row
    constexpr int M = N2 / (S * R);  // sub-transform length after this stage
    constexpr int NP = NB / 2;       // pairs per row
    static_assert(R % 2 == 0 && NB % 2 == 0 && (FIRST ? (M % 2 == 0 && S == 1) : S % 2 == 0) && !(FIRST && LAST), "pair stage: shape not supported");
    const int N = plan.N, N1 = plan.N1;
    const int p = threadIdx.x;
    const bool act = p < nrow * NP;
    float2 a0[R], a1[R];
    int row = 0, q = 0, r = 0, cell = 0, k1 = 0, bin = 0, sat = 0;
    float2 w0 = make_float2(1.f, 0.f), w1 = make_float2(1.f, 0.f), wd = make_float2(1.f, 0.f);
    if (act)
        {
            row = p / NP;
            const int u = 2 * (p - row * NP);
            q = u / S;
            r = u - q * S;
            if (FIRST || LAST)
                {
                    const float inv_n1 = 1.0f / (float)N1, inv_ns = 1.0f / (float)g.n_sats;
                    const int rowid = row0 + row;
                    if (g.sat_fastest)
                        {
                            // (bin, k1, sat): the rows of a workgroup are the SAME row of the signal spectrum for neighbouring
                            // satellites, so its A loads hit the CU's L1 after the first row's
                            const int bk = fdiv(rowid, inv_ns);
                            sat = rowid - bk * g.n_sats;
                            bin = fdiv(bk, inv_n1);
                            k1 = bk - bin * N1;
                        }
                    else
                        {
                            const int cl = fdiv(rowid, inv_n1);
                            k1 = rowid - cl * N1;
                            bin = fdiv(cl, inv_ns);
                            sat = cl - bin * g.n_sats;
                        }
                    cell = sat * g.n_bins + bin;
                }
            // twiddle seeds first: their latency hides behind the gather below
            if (LAST)
                {
                    // inter-pass twiddle w_N^(k1 * (r + S k)) = w_N^(k1 r) * (w_N^(k1 S))^k, for r and r + 1
                    const acq_f32x4 b = *reinterpret_cast<const acq_f32x4*>(g.wN + (size_t)k1 * N2 + r);
                    w0 = make_float2(b.x, b.y);
                    w1 = make_float2(b.z, b.w);
                    wd = g.wN[(size_t)k1 * N2 + S];
                }
            else if (FIRST)
                {
                    const acq_f32x4 b = *reinterpret_cast<const acq_f32x4*>(twf + q);  // w_n^q, w_n^(q+1)  (table row k = 1)
                    w0 = make_float2(b.x, b.y);
                    w1 = make_float2(b.z, b.w);
                }
            else
                w0 = twf[q];
            if (FIRST)
                {
                    const float2* ap = g.A + (size_t)bin * N + (size_t)k1 * N2 + q;
                    if (g.B)
                        {
                            const float2* bp = g.B + (size_t)sat * N + (size_t)k1 * N2 + q;
#pragma unroll
                            for (int j = 0; j < R; j++)
                                {
                                    const acq_f32x4 va = *reinterpret_cast<const acq_f32x4*>(ap + M * j);
                                    const acq_f32x4 vb = *reinterpret_cast<const acq_f32x4*>(bp + M * j);
                                    a0[j] = cmul(make_float2(va.x, va.y), make_float2(vb.x, vb.y));
                                    a1[j] = cmul(make_float2(va.z, va.w), make_float2(vb.z, vb.w));
                                }
                        }
                    else
                        {
#pragma unroll
                            for (int j = 0; j < R; j++)
                                {
                                    const acq_f32x4 va = *reinterpret_cast<const acq_f32x4*>(ap + M * j);
                                    a0[j] = make_float2(va.x, va.y);
                                    a1[j] = make_float2(va.z, va.w);
                                }
                        }
                }
            else
                {
                    const float2* x = lds + row * N2 + r + S * q;
#pragma unroll
                    for (int j = 0; j < R; j++)
                        {
                            const acq_f32x4 v = *reinterpret_cast<const acq_f32x4*>(x + S * M * j);
                            a0[j] = make_float2(v.x, v.y);
                            a1[j] = make_float2(v.z, v.w);
                        }
                }
        }
    if (!FIRST) __syncthreads();  // every input of this stage has left LDS
    if (act)
        {
            dftR<R, INV>(a0);
            dftR<R, INV>(a1);
            float2 tw[R];
            if (LAST)
                {
                    tw_powers<R, INV>(wd, tw);
                    float2* qp = g.Q + (size_t)cell * N + (size_t)k1 * N2 + r;
#pragma unroll
                    for (int k = 0; k < R; k++)
                        {
                            const float2 o0 = tmul<INV>(a0[k], cmul(w0, tw[k]));
                            const float2 o1 = tmul<INV>(a1[k], cmul(w1, tw[k]));
                            *reinterpret_cast<acq_f32x4*>(qp + S * k) = acq_f32x4{o0.x, o0.y, o1.x, o1.y};
                        }
                }
            else if (FIRST)
                {
                    // S == 1: butterfly q writes y[R q + k], k < R: 2 R contiguous elements for the pair
                    float2* y = lds + row * N2 + R * q;
                    tw_powers<R, INV>(w0, tw);
#pragma unroll
                    for (int k = 0; k < R; k += 2)
                        {
                            const float2 o0 = (k == 0) ? a0[0] : tmul<INV>(a0[k], tw[k]);
                            const float2 o1 = tmul<INV>(a0[k + 1], tw[k + 1]);
                            *reinterpret_cast<acq_f32x4*>(y + k) = acq_f32x4{o0.x, o0.y, o1.x, o1.y};
                        }
                    tw_powers<R, INV>(w1, tw);
#pragma unroll
                    for (int k = 0; k < R; k += 2)
                        {
                            const float2 o0 = (k == 0) ? a1[0] : tmul<INV>(a1[k], tw[k]);
                            const float2 o1 = tmul<INV>(a1[k + 1], tw[k + 1]);
                            *reinterpret_cast<acq_f32x4*>(y + R + k) = acq_f32x4{o0.x, o0.y, o1.x, o1.y};
                        }
                }
            else
                {
                    // the pair shares q, hence the twiddles; outputs r + S (R q + k) and the next element
                    tw_powers<R, INV>(w0, tw);
                    float2* y = lds + row * N2 + r + S * R * q;
#pragma unroll
                    for (int k = 0; k < R; k++)
                        {
                            const float2 o0 = (k == 0) ? a0[0] : tmul<INV>(a0[k], tw[k]);
                            const float2 o1 = (k == 0) ? a1[0] : tmul<INV>(a1[k], tw[k]);
                            *reinterpret_cast<acq_f32x4*>(y + S * k) = acq_f32x4{o0.x, o0.y, o1.x, o1.y};
                        }
                }
        }
    if (!LAST) __syncthreads();  // outputs visible to the next stage
}

template <bool INV, int N2, int S, int F, int... RS>
struct Rows2pRun;
template <bool INV, int N2, int S, int F>
struct Rows2pRun<INV, N2, S, F>
{
    static __device__ __forceinline__ void run(const AcqFftPlan&, const AcqRows2Args&, float2*, int, int) {}
};
template <bool INV, int N2, int S, int F, int R0, int... REST>
struct Rows2pRun<INV, N2, S, F, R0, REST...>
{
    static __device__ __forceinline__ void run(const AcqFftPlan& plan, const AcqRows2Args& g, float2* lds, int row0, int nrow)
    {
        rows2p_stage<R0, INV, N2, S, F == 0, sizeof...(REST) == 0>(plan, g, lds, row0, nrow, g.wN2 + plan.tw_off[F]);
        Rows2pRun<INV, N2, S * R0, F + 1, REST...>::run(plan, g, lds, row0, nrow);
    }
};
template <int... RS>
struct RowsLen;
template <>
struct RowsLen<>
{
    static constexpr int value = 1;
};
template <int R0, int... REST>
struct RowsLen<R0, REST...>
{
    static constexpr int value = R0 * RowsLen<REST...>::value;
};

#ifndef ACQ_ROWS2P_WAVES
#define ACQ_ROWS2P_WAVES 4
#endif
// radices RS...; rows per workgroup such that rows * N2 / (2 R) <= 256 threads for every stage
template <bool INV, int... RS>
__global__ __launch_bounds__(ACQ_THREADS, ACQ_ROWS2P_WAVES) void acq_rows2p_kernel(AcqFftPlan plan, AcqRows2Args g)
{
    extern __shared__ float2 sm[];
    const int per_xcd = gridDim.x >> 3;
    const int group = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (group >= g.n_groups) return;
    const int row0 = group * g.rpw;
    const int nrow = min(g.rpw, g.n_rows - row0);
    Rows2pRun<INV, RowsLen<RS...>::value, 1, 0, RS...>::run(plan, g, sm, row0, nrow);
}

// ---- N2 = 1000 = 10 x 10 x 10 on butterfly pairs, PLANAR and packed across the pair ----------------------------------------
#endif  // GNSSCORR_EXPERIMENTS

// Same decomposition as rows2p_stage (a thread owns butterflies u = 2p, 2p + 1 of a row; 16-byte global accesses; twiddles from
// seeds), but the registers hold (butterfly 0, butterfly 1) pairs of REAL parts and of IMAGINARY parts: every operation of the
// radix-10 butterfly is then one packed-FP32 instruction that serves both butterflies with no component shuffles.  The
// interleaved form above packs (re, im) of one value, which the +-j rotations and the constant twiddles of the DFT keep taking
// apart: its disassembly has 250 register moves among 1148 VALU instructions per wave.  LDS holds the rows as two planes (re,
// im) so that a pair's inputs and outputs of the LDS stages are 8-byte accesses of one plane each.
typedef float acq_pk2 __attribute__((ext_vector_type(2)));
// the planar complex arithmetic below is written once for a value type V: acq_pk2 (a pair of butterflies per thread, packed
// instructions) or float (one butterfly per thread)
static __device__ __forceinline__ acq_pk2 pfma(acq_pk2 a, acq_pk2 b, acq_pk2 c) { return __builtin_elementwise_fma(a, b, c); }
static __device__ __forceinline__ float pfma(float a, float b, float c) { return fmaf(a, b, c); }
template <class V>
static __device__ __forceinline__ V vsplat(float c);
template <>
__device__ __forceinline__ acq_pk2 vsplat<acq_pk2>(float c) { return acq_pk2{c, c}; }
template <>
__device__ __forceinline__ float vsplat<float>(float c) { return c; }
static __device__ __forceinline__ acq_pk2 psplat(float c) { return acq_pk2{c, c}; }
template <class V>
struct VC  // a complex value per butterfly held by the thread
{
    V r, i;
};
typedef VC<acq_pk2> PkC;
template <class V>
static __device__ __forceinline__ VC<V> pk_mul(VC<V> a, VC<V> b) { return VC<V>{pfma(a.r, b.r, -(a.i * b.i)), pfma(a.r, b.i, a.i * b.r)}; }
template <class V>
static __device__ __forceinline__ VC<V> pk_mul_conj(VC<V> a, VC<V> b) { return VC<V>{pfma(a.r, b.r, a.i * b.i), pfma(a.i, b.r, -(a.r * b.i))}; }
template <bool INV, class V>
static __device__ __forceinline__ VC<V> pk_tmul(VC<V> a, VC<V> w) { return INV ? pk_mul_conj(a, w) : pk_mul(a, w); }
template <class V>
static __device__ __forceinline__ VC<V> pk_add(VC<V> a, VC<V> b) { return VC<V>{a.r + b.r, a.i + b.i}; }
template <class V>
static __device__ __forceinline__ VC<V> pk_sub(VC<V> a, VC<V> b) { return VC<V>{a.r - b.r, a.i - b.i}; }
// a + (-j) b (forward) / a + j b (inverse), and the same with the opposite sign of b
template <bool INV, class V>
static __device__ __forceinline__ VC<V> pk_add_mj(VC<V> a, VC<V> b) { return INV ? VC<V>{a.r - b.i, a.i + b.r} : VC<V>{a.r + b.i, a.i - b.r}; }
template <bool INV, class V>
static __device__ __forceinline__ VC<V> pk_sub_mj(VC<V> a, VC<V> b) { return INV ? VC<V>{a.r + b.i, a.i - b.r} : VC<V>{a.r - b.i, a.i + b.r}; }
// a * exp(-+j phi) for a constant phi
template <bool INV, class V>
static __device__ __forceinline__ VC<V> pk_mul_const(VC<V> a, float c, float sn)
{
    const V cc = vsplat<V>(c), ss = vsplat<V>(sn);
    return INV ? VC<V>{pfma(a.r, cc, -(a.i * ss)), pfma(a.i, cc, a.r * ss)} : VC<V>{pfma(a.r, cc, a.i * ss), pfma(a.i, cc, -(a.r * ss))};
}
template <bool INV, class V>
static __device__ __forceinline__ void pk_dft5(VC<V>* a)
{
    const V c1 = vsplat<V>(0.30901699437494742410f), c2 = vsplat<V>(-0.80901699437494742410f);
    const V s1 = vsplat<V>(0.95105651629515357212f), s2 = vsplat<V>(0.58778525229247312917f);
    const VC<V> t1 = pk_add(a[1], a[4]), t2 = pk_add(a[2], a[3]), t3 = pk_sub(a[1], a[4]), t4 = pk_sub(a[2], a[3]);
    const VC<V> m1 = {pfma(c2, t2.r, pfma(c1, t1.r, a[0].r)), pfma(c2, t2.i, pfma(c1, t1.i, a[0].i))};
    const VC<V> m2 = {pfma(c1, t2.r, pfma(c2, t1.r, a[0].r)), pfma(c1, t2.i, pfma(c2, t1.i, a[0].i))};
    const VC<V> v1 = {pfma(s2, t4.r, s1 * t3.r), pfma(s2, t4.i, s1 * t3.i)};
    const VC<V> v2 = {pfma(s2, t3.r, -(s1 * t4.r)), pfma(s2, t3.i, -(s1 * t4.i))};
    a[0] = pk_add(a[0], pk_add(t1, t2));
    a[1] = pk_add_mj<INV>(m1, v1);
    a[4] = pk_sub_mj<INV>(m1, v1);
    a[2] = pk_add_mj<INV>(m2, v2);
    a[3] = pk_sub_mj<INV>(m2, v2);
}
template <bool INV, class V>
static __device__ __forceinline__ void pk_dft10(VC<V>* a)
{
    VC<V> e[5] = {a[0], a[2], a[4], a[6], a[8]};
    VC<V> o[5] = {a[1], a[3], a[5], a[7], a[9]};
    pk_dft5<INV>(e);
    pk_dft5<INV>(o);
    o[1] = pk_mul_const<INV>(o[1], 0.80901699437494742410f, 0.58778525229247312917f);
    o[2] = pk_mul_const<INV>(o[2], 0.30901699437494742410f, 0.95105651629515357212f);
    o[3] = pk_mul_const<INV>(o[3], -0.30901699437494742410f, 0.95105651629515357212f);
    o[4] = pk_mul_const<INV>(o[4], -0.80901699437494742410f, 0.58778525229247312917f);
#pragma unroll
    for (int k = 0; k < 5; k++)
        {
            a[k] = pk_add(e[k], o[k]);
            a[k + 5] = pk_sub(e[k], o[k]);
        }
}
// w^k, k < 10, from w (a seed per butterfly)
template <class V>
static __device__ __forceinline__ void pk_powers10(VC<V> w, VC<V>* tw)
{
    tw[0] = VC<V>{vsplat<V>(1.0f), vsplat<V>(0.0f)};
    tw[1] = w;
#pragma unroll
    for (int k = 2; k < 10; k++) tw[k] = (k % 2 == 0) ? pk_mul(tw[k / 2], tw[k / 2]) : pk_mul(tw[k - 1], w);
}

// The ten (re, im) input pairs of an LDS stage, x[100 j], as twenty ds_read_b64.  Written in assembly because the compiler merges
// neighbouring 8-byte LDS reads into ds_read2_b64, which the LDS serves at HALF the rate of two ds_read_b64 (8 cycles against 2 + 2
// per wave: MI355X_MICROARCH.md, LDS table) -- 80 of the ~415 LDS cycles a wave spends in this kernel.  One wait covers all twenty;
// the values pass through it as operands so that no use is scheduled above it.
static __device__ __forceinline__ void lds_read_pairs10(const float* xr, const float* xi, PkC* a)
{
    const unsigned ar = (unsigned)reinterpret_cast<uintptr_t>(xr), ai = (unsigned)reinterpret_cast<uintptr_t>(xi);  // low 32 bits of a flat LDS address = the LDS address
    // ONE statement: the twenty reads and their wait.  The compiler's own s_waitcnt insertion does not see LDS operations inside inline
    // assembly, so nothing (a register copy, a spill of an output) may be scheduled between issue and wait; early-clobber outputs keep
    // the address registers apart from the destinations
    asm volatile(
        "ds_read_b64 %0, %20 offset:0\n\t"
        "ds_read_b64 %1, %21 offset:0\n\t"
        "ds_read_b64 %2, %20 offset:400\n\t"
        "ds_read_b64 %3, %21 offset:400\n\t"
        "ds_read_b64 %4, %20 offset:800\n\t"
        "ds_read_b64 %5, %21 offset:800\n\t"
        "ds_read_b64 %6, %20 offset:1200\n\t"
        "ds_read_b64 %7, %21 offset:1200\n\t"
        "ds_read_b64 %8, %20 offset:1600\n\t"
        "ds_read_b64 %9, %21 offset:1600\n\t"
        "ds_read_b64 %10, %20 offset:2000\n\t"
        "ds_read_b64 %11, %21 offset:2000\n\t"
        "ds_read_b64 %12, %20 offset:2400\n\t"
        "ds_read_b64 %13, %21 offset:2400\n\t"
        "ds_read_b64 %14, %20 offset:2800\n\t"
        "ds_read_b64 %15, %21 offset:2800\n\t"
        "ds_read_b64 %16, %20 offset:3200\n\t"
        "ds_read_b64 %17, %21 offset:3200\n\t"
        "ds_read_b64 %18, %20 offset:3600\n\t"
        "ds_read_b64 %19, %21 offset:3600\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a[0].r), "=&v"(a[0].i), "=&v"(a[1].r), "=&v"(a[1].i), "=&v"(a[2].r), "=&v"(a[2].i), "=&v"(a[3].r), "=&v"(a[3].i), "=&v"(a[4].r), "=&v"(a[4].i), "=&v"(a[5].r), "=&v"(a[5].i), "=&v"(a[6].r), "=&v"(a[6].i), "=&v"(a[7].r), "=&v"(a[7].i), "=&v"(a[8].r), "=&v"(a[8].i), "=&v"(a[9].r), "=&v"(a[9].i)
        : "v"(ar), "v"(ai)
        : "memory");
}

#ifndef ACQ_ROWS3_DBG
#define ACQ_ROWS3_DBG 0  // 1: compile the phase-elimination switches of $GNSSCORR_ACQ_DBG in (profiles/tools/rows3_phases.sh builds with it)
#endif
#ifndef ACQ_ROWS3_NT
#define ACQ_ROWS3_NT 0  // 1: streaming stores of the inter-pass buffer
#endif
#ifndef ACQ_ROWS3_WAVES
#define ACQ_ROWS3_WAVES 4
#endif
#ifndef ACQ_ROWS3_STAGE2_QFAST
#define ACQ_ROWS3_STAGE2_QFAST 1  // lane -> pair mapping of the second stage (see there)
#endif
// slot `slot` of `slots_per_xcd` on XCD `xcd` (the launch decides which blocks those are)
template <bool INV>
static __device__ __forceinline__ void acq_rows3_body(const AcqFftPlan& plan, const AcqRows2Args& g, float2* sm, int xcd, int slot, int slots_per_xcd)
{
    constexpr int R = 10, N2 = 1000, NB = N2 / R, NP = NB / 2;
    float* pre = reinterpret_cast<float*>(sm);  // plane of real parts: rpw rows of N2
    float* pim = pre + g.rpw * N2;              // plane of imaginary parts
    // Each XCD owns one contiguous eighth of the row groups (blocks with equal blockIdx % 8 share an L2); a workgroup walks its
    // XCD's groups with the stride of the launch: once when the grid has a block per group, several times when the launch is
    // sized to the chip (persistent form: the stores of one group drain behind the loads of the next instead of at the wave's end)
    const int wg_per_xcd = slots_per_xcd;
    const int groups_per_xcd = (g.n_groups + 7) >> 3;
    const int N = plan.N, N1 = plan.N1;
    const int p = threadIdx.x;
#pragma nounroll
    for (int gi = slot; gi < groups_per_xcd; gi += wg_per_xcd)
    {
    const int group = xcd * groups_per_xcd + gi;
    if (group >= g.n_groups) break;
    const int row0 = group * g.rpw;
    const int nrow = min(g.rpw, g.n_rows - row0);
    const bool act = p < nrow * NP;
    const int row = act ? p / NP : 0;
    const int u = 2 * (p - row * NP);  // even butterfly of the pair, 0 .. 98
    int cell, k1, bin, sat;
    {
        const float inv_n1 = 1.0f / (float)N1, inv_ns = 1.0f / (float)g.n_sats;
        const int rowid = row0 + row;
        if (g.sat_fastest == 2)
            {
                const int kb = fdiv(rowid, inv_ns);
                sat = rowid - kb * g.n_sats;
                k1 = fdiv(kb, 1.0f / (float)g.n_bins);
                bin = kb - k1 * g.n_bins;
            }
        else if (g.sat_fastest)
            {
                const int bk = fdiv(rowid, inv_ns);
                sat = rowid - bk * g.n_sats;
                bin = fdiv(bk, inv_n1);
                k1 = bk - bin * N1;
            }
        else
            {
                const int cl = fdiv(rowid, inv_n1);
                k1 = rowid - cl * N1;
                bin = fdiv(cl, inv_ns);
                sat = cl - bin * g.n_sats;
            }
        cell = sat * g.n_bins + bin;
    }
#if ACQ_ROWS3_DBG
    if ((g.dbg & 64) && g.ts && p == 0) g.ts[(size_t)group * 8 + 0] = __builtin_readcyclecounter() * 0 + wall_clock64();
#endif
    PkC a[R], tw[R];
#ifndef ACQ_ROWS3_HOIST_TW
#define ACQ_ROWS3_HOIST_TW 0  // 1: the twiddle seeds of stages 2 and 3 are requested here, in front of the stage-1 inputs (loads return in order:
                              // whatever waits for the inputs has them too), instead of at the head of their stages.  Measured (round 4, same box,
                              // rocprofv3 means of 88 launches): 37.6 / 37.5 us against 36.9 / 38.1 us -- nothing: the 13 % that phase elimination
                              // attributes to "twiddle loads off" is the arithmetic on constant seeds, not the loads' latency.  Kept as a knob
#endif
    const int p2 = p - row * NP;  // pair of the row, 0 .. 49
    const int u2 = ACQ_ROWS3_STAGE2_QFAST ? 10 * (p2 % 10) + 2 * (p2 / 10) : u;  // this lane's pair in stage 2 (see there)
    float2 sd2 = make_float2(1.f, 0.f), d3 = make_float2(1.f, 0.f);
    acq_f32x4 b3 = acq_f32x4{1.f, 0.f, 1.f, 0.f};
    if (ACQ_ROWS3_HOIST_TW && act && !(ACQ_ROWS3_DBG && (g.dbg & 4)))
        {
            sd2 = g.wN2[plan.tw_off[1] + u2 / 10];                                              // w_100^q
            if (!(ACQ_ROWS3_DBG && (g.dbg & 16)))
                {
                    b3 = *reinterpret_cast<const acq_f32x4*>(g.wN + (size_t)k1 * N2 + u);       // w_N^(k1 r), w_N^(k1 (r + 1))
                    d3 = g.wN[(size_t)k1 * N2 + 100];                                           // w_N^(100 k1)
                }
        }
    // ---- stage 1: S = 1, M = 100; butterflies q = u, u + 1; inputs x[q + 100 j] from global memory (x the code spectrum) ----
    if (act)
        {
            const acq_f32x4 sd = (ACQ_ROWS3_DBG && (g.dbg & 4)) ? acq_f32x4{1.f, 0.f, 1.f, 0.f} : *reinterpret_cast<const acq_f32x4*>(g.wN2 + plan.tw_off[0] + u);  // w_1000^u, w_1000^(u+1)
            const float2* ap = g.A + (size_t)bin * N + (size_t)k1 * N2 + u;
            if (ACQ_ROWS3_DBG && (g.dbg & 1))
                {
#pragma unroll
                    for (int j = 0; j < R; j++) a[j] = PkC{acq_pk2{(float)(u + j), 1.0f}, acq_pk2{0.5f, (float)j}};
                }
            else if (g.B && !(ACQ_ROWS3_DBG && (g.dbg & 16)))
                {
                    const float2* bp = g.B + (size_t)sat * N + (size_t)k1 * N2 + u;
#pragma unroll
                    for (int j = 0; j < R; j++)
                        {
                            const acq_f32x4 va = *reinterpret_cast<const acq_f32x4*>(ap + 100 * j);
                            const acq_f32x4 vb = *reinterpret_cast<const acq_f32x4*>(bp + 100 * j);
                            a[j] = pk_mul(PkC{acq_pk2{va.x, va.z}, acq_pk2{va.y, va.w}}, PkC{acq_pk2{vb.x, vb.z}, acq_pk2{vb.y, vb.w}});
                        }
                }
            else
                {
#pragma unroll
                    for (int j = 0; j < R; j++)
                        {
                            const acq_f32x4 va = *reinterpret_cast<const acq_f32x4*>(ap + 100 * j);
                            a[j] = PkC{acq_pk2{va.x, va.z}, acq_pk2{va.y, va.w}};
                        }
                }
            if (!(ACQ_ROWS3_DBG && (g.dbg & 8))) pk_dft10<INV>(a);
            pk_powers10(PkC{acq_pk2{sd.x, sd.z}, acq_pk2{sd.y, sd.w}}, tw);
#pragma unroll
            for (int k = 1; k < R; k++) a[k] = pk_tmul<INV>(a[k], tw[k]);
            // butterfly q writes y[10 q + k]: the pair's 20 outputs are contiguous in each plane
            float* yr = pre + row * N2 + R * u;
            float* yi = pim + row * N2 + R * u;
            // as 16-byte stores: the lanes are 80 bytes apart, which 8-byte stores hit two-way bank conflicts with and 16-byte stores do not
            *reinterpret_cast<acq_f32x4*>(yr + 0) = acq_f32x4{a[0].r.x, a[1].r.x, a[2].r.x, a[3].r.x};
            *reinterpret_cast<acq_f32x4*>(yr + 4) = acq_f32x4{a[4].r.x, a[5].r.x, a[6].r.x, a[7].r.x};
            *reinterpret_cast<acq_f32x4*>(yr + 8) = acq_f32x4{a[8].r.x, a[9].r.x, a[0].r.y, a[1].r.y};
            *reinterpret_cast<acq_f32x4*>(yr + 12) = acq_f32x4{a[2].r.y, a[3].r.y, a[4].r.y, a[5].r.y};
            *reinterpret_cast<acq_f32x4*>(yr + 16) = acq_f32x4{a[6].r.y, a[7].r.y, a[8].r.y, a[9].r.y};
            *reinterpret_cast<acq_f32x4*>(yi + 0) = acq_f32x4{a[0].i.x, a[1].i.x, a[2].i.x, a[3].i.x};
            *reinterpret_cast<acq_f32x4*>(yi + 4) = acq_f32x4{a[4].i.x, a[5].i.x, a[6].i.x, a[7].i.x};
            *reinterpret_cast<acq_f32x4*>(yi + 8) = acq_f32x4{a[8].i.x, a[9].i.x, a[0].i.y, a[1].i.y};
            *reinterpret_cast<acq_f32x4*>(yi + 12) = acq_f32x4{a[2].i.y, a[3].i.y, a[4].i.y, a[5].i.y};
            *reinterpret_cast<acq_f32x4*>(yi + 16) = acq_f32x4{a[6].i.y, a[7].i.y, a[8].i.y, a[9].i.y};
        }
#if ACQ_ROWS3_DBG
    if ((g.dbg & 64) && g.ts && p == 0) g.ts[(size_t)group * 8 + 1] = __builtin_readcyclecounter() * 0 + wall_clock64();
#endif
    __syncthreads();
#if ACQ_ROWS3_DBG
    if ((g.dbg & 64) && g.ts && p == 0) g.ts[(size_t)group * 8 + 2] = __builtin_readcyclecounter() * 0 + wall_clock64();
#endif
    // ---- stage 2: S = 10, M = 10; butterflies u2 = 10 q + r, the pair shares q; inputs x[r + 10 q + 100 j] ----
    // Which pair of the row a lane takes in THIS stage is free (stage 1 wrote and stage 3 reads by position, not by owner).  With
    // consecutive lanes on consecutive r (q = lane / 5) the 8-byte stores of a 16-lane group fall into three 10-dword runs 100 dwords
    // apart, i.e. 4, 8 and 12 banks apart modulo 32: three-way conflicts, 12 instead of 4 LDS cycles per store instruction -- 160 of the
    // 195 conflict cycles per wave the counters show (SQ_LDS_BANK_CONFLICT / SQ_WAVES), the rest being the reads of waves that
    // straddle two rows.  With consecutive lanes on consecutive q (q = lane % 10, r = 2 (lane / 10)) the stores of a group are 2-dword
    // runs 100 dwords = 4 banks apart (two-way on the 9th and 10th), and the 8-byte reads are 10 dwords apart: all 64 banks once.
    {
        const int q = u2 / 10, r = u2 - 10 * q;
        float2 sd = sd2;
        if (act)
            {
                if (!ACQ_ROWS3_HOIST_TW && !(ACQ_ROWS3_DBG && (g.dbg & 4))) sd = g.wN2[plan.tw_off[1] + q];  // w_100^q
                const float* xr = pre + row * N2 + u2;
                const float* xi = pim + row * N2 + u2;
                lds_read_pairs10(xr, xi, a);
            }
        __syncthreads();  // every input of this stage has left LDS
        if (act)
            {
                if (!(ACQ_ROWS3_DBG && (g.dbg & 8))) pk_dft10<INV>(a);
                pk_powers10(PkC{psplat(sd.x), psplat(sd.y)}, tw);
                float* yr = pre + row * N2 + r + 100 * q;
                float* yi = pim + row * N2 + r + 100 * q;
                *reinterpret_cast<acq_pk2*>(yr) = a[0].r;
                *reinterpret_cast<acq_pk2*>(yi) = a[0].i;
#pragma unroll
                for (int k = 1; k < R; k++)
                    {
                        const PkC o = pk_tmul<INV>(a[k], tw[k]);
                        *reinterpret_cast<acq_pk2*>(yr + 10 * k) = o.r;
                        *reinterpret_cast<acq_pk2*>(yi + 10 * k) = o.i;
                    }
            }
        __syncthreads();
    }
#if ACQ_ROWS3_DBG
    if ((g.dbg & 64) && g.ts && p == 0) g.ts[(size_t)group * 8 + 3] = __builtin_readcyclecounter() * 0 + wall_clock64();
#endif
    // ---- stage 3: S = 100, M = 1; r = u; inputs x[r + 100 j]; outputs n2 = r + 100 k with the inter-pass twiddle ----
    if (act)
        {
            const acq_f32x4 b = ACQ_ROWS3_HOIST_TW ? b3 : (ACQ_ROWS3_DBG && (g.dbg & (4 | 16))) ? acq_f32x4{1.f, 0.f, 1.f, 0.f} : *reinterpret_cast<const acq_f32x4*>(g.wN + (size_t)k1 * N2 + u);  // w_N^(k1 r), w_N^(k1 (r + 1))
            const float2 d = ACQ_ROWS3_HOIST_TW ? d3 : (ACQ_ROWS3_DBG && (g.dbg & (4 | 16))) ? make_float2(1.f, 0.f) : g.wN[(size_t)k1 * N2 + 100];                                          // w_N^(100 k1)
            const float* xr = pre + row * N2 + u;
            const float* xi = pim + row * N2 + u;
            lds_read_pairs10(xr, xi, a);
            if (!(ACQ_ROWS3_DBG && (g.dbg & 8))) pk_dft10<INV>(a);
#if ACQ_ROWS3_DBG
            if (g.dbg & 16)
                {
                    // TIMING MOCK of a rows-LAST pass (columns first): no inter-pass twiddle, |.|^2 of the outputs, 8 bytes per k stored by
                    // every second cell (a dwell pair's sum written once): 2N bytes per transform instead of 8N
                    float* gp = reinterpret_cast<float*>(g.Q) + (size_t)(cell >> 1) * N + (size_t)k1 * N2 + u;
#pragma unroll
                    for (int k = 0; k < R; k++)
                        {
                            const acq_pk2 m = a[k].r * a[k].r + a[k].i * a[k].i;
                            if (!(cell & 1) || m.x == 1.2345e-33f) *reinterpret_cast<acq_pk2*>(gp + 100 * k) = m;
                        }
                }
            else
#endif
            {
            pk_powers10(PkC{psplat(d.x), psplat(d.y)}, tw);
            const PkC bb = {acq_pk2{b.x, b.z}, acq_pk2{b.y, b.w}};
#if ACQ_ROWS3_DBG
            if ((g.dbg & 64) && g.ts && p == 0) g.ts[(size_t)group * 8 + 4] = __builtin_readcyclecounter() * 0 + wall_clock64();
#endif
            float2* qp = g.Q + (size_t)cell * N + (size_t)k1 * N2 + u;
#pragma unroll
            for (int k = 0; k < R; k++)
                {
                    const PkC o = pk_tmul<INV>(a[k], k == 0 ? bb : pk_mul(bb, tw[k]));
#if ACQ_ROWS3_NT
                    if (!(ACQ_ROWS3_DBG && (g.dbg & 2)) || o.r.x == 1.2345e-33f) __builtin_nontemporal_store(acq_f32x4{o.r.x, o.i.x, o.r.y, o.i.y}, reinterpret_cast<acq_f32x4*>(qp + 100 * k));
#else
                    if (!(ACQ_ROWS3_DBG && (g.dbg & 2)) || o.r.x == 1.2345e-33f) *reinterpret_cast<acq_f32x4*>(qp + 100 * k) = acq_f32x4{o.r.x, o.i.x, o.r.y, o.i.y};
#endif
                }
            }
        }
#if ACQ_ROWS3_DBG
    if ((g.dbg & 64) && g.ts && p == 0) g.ts[(size_t)group * 8 + 5] = __builtin_readcyclecounter() * 0 + wall_clock64();
#endif
    __syncthreads();  // stage 3 has read the planes: the next group's stage 1 may overwrite them
#if ACQ_ROWS3_DBG
    if ((g.dbg & 64) && g.ts && p == 0) g.ts[(size_t)group * 8 + 6] = __builtin_readcyclecounter() * 0 + wall_clock64();
#endif

    }
}

#ifndef ACQ_ROWS3_THREADS
#define ACQ_ROWS3_THREADS 256  // 512: ten rows per workgroup (80 KB of LDS, two workgroups per CU), half as many workgroups to dispatch
#endif
template <bool INV>
__global__ __launch_bounds__(ACQ_ROWS3_THREADS, ACQ_ROWS3_WAVES) void acq_rows3_kernel(AcqFftPlan plan, AcqRows2Args g)
{
    extern __shared__ float2 sm[];
    acq_rows3_body<INV>(plan, g, sm, blockIdx.x & 7, blockIdx.x >> 3, gridDim.x >> 3);
}

// stage list: RI = R*16 + ITER per stage, run in order
template <bool INV, int N2, int S, int F, int... RI>
struct Rows2Run;
template <bool INV, int N2, int S, int F>
struct Rows2Run<INV, N2, S, F>
{
    static __device__ __forceinline__ void run(const AcqFftPlan&, const AcqRows2Args&, float2*, int, int) {}
};
template <bool INV, int N2, int S, int F, int RI0, int... REST>
struct Rows2Run<INV, N2, S, F, RI0, REST...>
{
    static __device__ __forceinline__ void run(const AcqFftPlan& plan, const AcqRows2Args& g, float2* lds, int row0, int nrow)
    {
        constexpr int R = RI0 / 16, ITER = RI0 % 16;
        rows2_stage<R, ITER, INV, N2, S, F == 0, sizeof...(REST) == 0>(plan, g, lds, row0, nrow, g.wN2 + plan.tw_off[F]);
        Rows2Run<INV, N2, S * R, F + 1, REST...>::run(plan, g, lds, row0, nrow);
    }
};
template <int... RI>
struct Rows2Len;
template <>
struct Rows2Len<>
{
    static constexpr int value = 1;
};
template <int RI0, int... REST>
struct Rows2Len<RI0, REST...>
{
    static constexpr int value = (RI0 / 16) * Rows2Len<REST...>::value;
};

template <bool INV, int... RI>
__global__ __launch_bounds__(ACQ_THREADS, ACQ_ROWS2_WAVES) void acq_rows2_kernel(AcqFftPlan plan, AcqRows2Args g)
{
    extern __shared__ float2 sm[];
    // XCD-aware block mapping: workgroups are dealt round-robin over the 8 XCDs, so the blocks with equal
    // (blockIdx % 8) share an L2.  Each XCD gets one contiguous eighth of the row groups: a few bins of the
    // signal spectrum x all code spectra of the launch (~3 MB at N = 25000) instead of everything.
    const int per_xcd = gridDim.x >> 3;
    const int group = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (group >= g.n_groups) return;
    const int row0 = group * g.rpw;
    const int nrow = min(g.rpw, g.n_rows - row0);
    Rows2Run<INV, Rows2Len<RI...>::value, 1, 0, RI...>::run(plan, g, sm, row0, nrow);
}

// instantiated stage lists (radix, butterflies per thread), as acq_rows2_config derives them for the row
// lengths the planner picks for 1/2/4/8 ms blocks at 2 ... 25 Msps (and the power-of-two sizes)
typedef void (*AcqRows2Fn)(AcqFftPlan, AcqRows2Args);
struct AcqRows2Entry
{
    int n_stages;
    int ri[4];
    AcqRows2Fn fwd, inv;
    AcqRows2Fn pair_fwd, pair_inv;  // the same stage list on butterfly pairs (rows2p_stage), where its shape conditions hold
};
#define R2(r, i) ((r) * 16 + (i))
#define ROWS2_ENTRY3(a, b, c) {3, {a, b, c, 0}, &acq_rows2_kernel<false, a, b, c>, &acq_rows2_kernel<true, a, b, c>, nullptr, nullptr}
#define ROWS2_ENTRY4(a, b, c, d) {4, {a, b, c, d}, &acq_rows2_kernel<false, a, b, c, d>, &acq_rows2_kernel<true, a, b, c, d>, nullptr, nullptr}
static const AcqRows2Entry acq_rows2_registry[] = {
    {3, {R2(10, 2), R2(10, 2), R2(10, 2), 0}, &acq_rows2_kernel<false, R2(10, 2), R2(10, 2), R2(10, 2)>, &acq_rows2_kernel<true, R2(10, 2), R2(10, 2), R2(10, 2)>,
        &acq_rows3_kernel<false>, &acq_rows3_kernel<true>},  // 1000: N = 2000 ... 25000 (planar pair kernel)
    ROWS2_ENTRY3(R2(10, 1), R2(10, 1), R2(10, 1)),            // 1000, at most 2 rows per workgroup
    ROWS2_ENTRY3(R2(16, 1), R2(16, 1), R2(4, 4)),             // 1024
    ROWS2_ENTRY4(R2(10, 2), R2(5, 4), R2(5, 4), R2(5, 4)),    // 1250: N = 2500, 6250, 12500
    ROWS2_ENTRY3(R2(16, 1), R2(16, 1), R2(5, 4)),             // 1280: N = 32000
    ROWS2_ENTRY3(R2(16, 1), R2(16, 1), R2(8, 2)),             // 2048
    ROWS2_ENTRY3(R2(16, 1), R2(10, 2), R2(10, 2)),            // 1600: N = 40000
    ROWS2_ENTRY4(R2(16, 1), R2(5, 4), R2(5, 4), R2(5, 4)),    // 2000: N = 50000
    ROWS2_ENTRY3(R2(16, 1), R2(16, 1), R2(10, 1)),            // 2560: N = 64000
    ROWS2_ENTRY3(R2(16, 1), R2(16, 1), R2(16, 1)),            // 4096
    ROWS2_ENTRY4(R2(16, 1), R2(10, 2), R2(5, 4), R2(5, 4)),   // 4000: N = 100000
    ROWS2_ENTRY4(R2(16, 1), R2(10, 2), R2(3, 4), R2(2, 8)),   // 960: N = 24000
    ROWS2_ENTRY4(R2(16, 1), R2(10, 2), R2(10, 2), R2(2, 8)),  // 3200: N = 80000
};
#undef ROWS2_ENTRY3
#undef ROWS2_ENTRY4

// rows per workgroup and per-stage butterflies per thread for the packed kernel; false: use acq_rows_kernel
static bool acq_rows2_config(const AcqFftPlan& plan, int* rpw_out, int* iters)
{
    const int radices_ok[] = {2, 3, 4, 5, 8, 10, 16};
    for (int f = 0; f < plan.n_fac; f++)
        {
            bool ok = false;
            for (int r : radices_ok) ok = ok || (plan.fac[f] == r);
            if (!ok) return false;
        }
    if ((size_t)plan.N2 * sizeof(float2) > 64 * 1024) return false;
    int rpw = (int)(ACQ_ROWS2_LDS_BYTES / ((size_t)plan.N2 * sizeof(float2)));
    if (rpw < 1) rpw = 1;
    if (rpw > 16) rpw = 16;
    static const int rpw_cap = [] {
        const char* e = gc_exp_env("GNSSCORR_ACQ_RPW");  // tuning knob: cap on the rows per workgroup
        return e ? std::atoi(e) : 0;
    }();
    if (rpw_cap > 0 && rpw > rpw_cap) rpw = rpw_cap;
    for (; rpw >= 1; rpw--)
        {
            bool fits = true;
            for (int f = 0; f < plan.n_fac && fits; f++)
                {
                    const int R = plan.fac[f], nb = plan.N2 / R;
                    const int need = (rpw * nb + ACQ_THREADS - 1) / ACQ_THREADS;
                    int it = 1;
                    while (it < need) it *= 2;
                    if (it * R > ACQ_ROWS2_POINTS || it > 8) fits = false;
                    iters[f] = it;
                }
            if (fits)
                {
                    *rpw_out = rpw;
                    return true;
                }
        }
    return false;
}

// ---- register-resident N1-point DFT (Stockham, fully unrolled) ----
template <int N>
struct PickRadix
{
    static constexpr int value = (N % 5 == 0) ? 5 : (N % 4 == 0) ? 4 : (N % 3 == 0) ? 3 : (N % 2 == 0) ? 2 : N;
};

template <int NCUR, int S, int NTOT, bool INV>
struct RegFft
{
    // x: input, y: scratch; returns pointer parity through the recursion: result in `x` if the
    // number of stages is even, else in `y` (resolved at compile time by result_in_x)
    static constexpr int R = PickRadix<NCUR>::value;
    static constexpr int M = NCUR / R;
    static __device__ __forceinline__ void run(float2* x, float2* y, const float2* w)
    {
#pragma unroll
        for (int q = 0; q < M; q++)
            {
#pragma unroll
                for (int r = 0; r < S; r++)
                    {
                        float2 a[R];
#pragma unroll
                        for (int j = 0; j < R; j++) a[j] = x[r + S * (q + M * j)];
                        dftR<R, INV>(a);
#pragma unroll
                        for (int k = 0; k < R; k++)
                            {
                                float2 v = a[k];
                                if (q * k != 0)
                                    {
                                        float2 tw = w[(q * k * S) % NTOT];
                                        v = INV ? cmul_conj(v, tw) : cmul(v, tw);
                                    }
                                y[r + S * (R * q + k)] = v;
                            }
                    }
            }
        RegFft<M, S * R, NTOT, INV>::run(y, x, w);
    }
    static constexpr bool result_in_first = !RegFft<M, S * R, NTOT, INV>::result_in_first;
};
template <int S, int NTOT, bool INV>
struct RegFft<1, S, NTOT, INV>
{
    static __device__ __forceinline__ void run(float2*, float2*, const float2*) {}
    static constexpr bool result_in_first = true;
};

struct MaxPair
{
    float v;
    unsigned i;
};
// reference semantics of volk_gnsssdr_32f_index_max_32u: the first maximum wins
static __device__ __forceinline__ MaxPair max_pair(MaxPair a, MaxPair b)
{
    bool take_b = (b.v > a.v) || (b.v == a.v && b.i < a.i);
    return take_b ? b : a;
}

#ifndef ACQ_COLS_WAVES
#define ACQ_COLS_WAVES 1  // minimum waves per SIMD the columns kernel is compiled for.  4 (128 registers, a fourth workgroup per CU) was
                          // measured: the 25-point column transform then spills 43 registers and a search takes 0.60 instead of 0.50 ms
#endif
#ifndef ACQ_COLS_NT
#define ACQ_COLS_NT 2  // bit 0: streaming loads of the inter-pass buffer (measured SLOWER, 0.381 vs 0.345 ms per search: much of it is served by
                       // the Infinity Cache and the hint gives that up); bit 1: streaming stores of the magnitude grid (0.340 vs 0.345 ms)
#endif
#ifndef ACQ_COLS_PAIR_WAVES
#define ACQ_COLS_PAIR_WAVES 3  // the two-dwell epilogue is held to the registers of the one-dwell kernel (168: three waves per SIMD)
#endif
// one 256-column block `xblk` (of n_xblk) of cell `cell`.  p1s: N1 x ACQ_THREADS floats of LDS (two-dwell epilogues only), svs: 2 x
// ACQ_THREADS / 64 words of LDS; a caller that loops over blocks puts a __syncthreads() between them
template <int N1, bool INV, int EPI>
static __device__ __forceinline__ void acq_cols_body(const AcqFftPlan& plan, const float2* __restrict__ Q, float2* __restrict__ out, const AcqMagArgs& mag,
    int xblk, int n_xblk, int cell, float* __restrict__ p1s, float* __restrict__ svs)
{
    const int N2 = plan.N2, N = plan.N;
    // Column of this thread.  The row-permuted epilogues (forward transforms) store natural index m = n2 + N2 k at (m % N1) N2 + m / N1:
    // with consecutive columns on consecutive lanes every lane writes its 8 bytes N2 elements away from its neighbour's.  Where
    // N2 is a multiple of N1 J (J = 256 / N1 columns groups per block) the block takes columns N1 j + r instead, lane = r J + j: then
    // m % N1 = r and m / N1 = j + ..., so J consecutive lanes store J consecutive elements (80-byte runs at N1 = 25) -- 13.6 -> ~6 us
    // for the 82 spectra of a dwell pair.  The loads of the inter-pass buffer become strided, but it was written just before and sits in L2.
    constexpr bool PERM_EPI = (EPI == ACQ_EPI_PERM || EPI == ACQ_EPI_COMPLEX_CONJ_PERM);
    constexpr int PJ = ACQ_THREADS / N1;
    const bool perm_map = PERM_EPI && N1 > 1 && PJ >= 4 && (N2 % (N1 * PJ)) == 0 && n_xblk * (N1 * PJ) == N2;
    int n2 = xblk * ACQ_THREADS + threadIdx.x;
    bool active = n2 < N2;
    if (perm_map)
        {
            const int t = threadIdx.x, r = t / PJ, j = t - r * PJ;
            active = t < N1 * PJ;
            n2 = active ? xblk * (N1 * PJ) + N1 * j + r : 0;
        }
    float2 v0[N1], v1[N1];
    // ACQ_EPI_MAG2: the rows pass ran over 2 * n_bins "bins" per satellite, the second half being the next dwell's spectra
    constexpr bool PAIR = (EPI == ACQ_EPI_MAG2 || EPI == ACQ_EPI_MAG2_ACC);
    const size_t qcell = PAIR ? (size_t)(cell / mag.n_bins) * (2 * mag.n_bins) + cell % mag.n_bins : (size_t)cell;
    const float2* q = Q + qcell * N;
    // the inter-pass buffer is read exactly once (here): streaming loads when ACQ_COLS_NT & 1; the magnitude grid is written once per
    // search: streaming stores when ACQ_COLS_NT & 2
    auto ldq = [](const float2* p) -> float2 {
#if ACQ_COLS_NT & 1
        if (EPI >= ACQ_EPI_MAG)
            {
                typedef float nt_f2 __attribute__((ext_vector_type(2)));
                const nt_f2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f2*>(p));
                return make_float2(v.x, v.y);
            }
#endif
        return *p;
    };
#pragma unroll
    for (int k = 0; k < N1; k++) v0[k] = active ? ldq(q + (size_t)k * N2 + n2) : make_float2(0.f, 0.f);
    RegFft<N1, 1, N1, INV>::run(v0, v1, plan.w1);
    float2* res = RegFft<N1, 1, N1, INV>::result_in_first ? v0 : v1;
    // first dwell's magnitudes wait in LDS (N1 x 256 floats: 25 KB of the 160) while the second column is transformed: in
    // registers they cost the third wave per SIMD (208 instead of 168), and so do the second column's loads hoisted above the first transform
    if (PAIR)
        {
#pragma unroll
            for (int k = 0; k < N1; k++) p1s[k * ACQ_THREADS + threadIdx.x] = res[k].x * res[k].x + res[k].y * res[k].y;
            __builtin_amdgcn_sched_barrier(0);
            const float2* q2 = q + (size_t)mag.n_bins * N;
#pragma unroll
            for (int k = 0; k < N1; k++) v0[k] = active ? ldq(q2 + (size_t)k * N2 + n2) : make_float2(0.f, 0.f);
            RegFft<N1, 1, N1, INV>::run(v0, v1, plan.w1);
        }

    if (EPI == ACQ_EPI_COMPLEX)
        {
            if (active)
                {
                    float2* o = out + (size_t)cell * N;
#pragma unroll
                    for (int k = 0; k < N1; k++) o[(size_t)k * N2 + n2] = res[k];
                }
        }
    else if (EPI == ACQ_EPI_PERM || EPI == ACQ_EPI_COMPLEX_CONJ_PERM)
        {
            // natural index m = n2 + N2*k  ->  row-permuted position (m % N1)*N2 + m / N1
            if (active)
                {
                    float2* o = out + (size_t)cell * N;
#pragma unroll
                    for (int k = 0; k < N1; k++)
                        {
                            int m = n2 + N2 * k;
                            float2 v = res[k];
                            if (EPI == ACQ_EPI_COMPLEX_CONJ_PERM) v.y = -v.y;
                            o[(size_t)(m % N1) * N2 + m / N1] = v;
                        }
                }
        }
    else
        {
            // |.|^2, non-coherent accumulation, per-block maximum of the grid row
            float* g = mag.grid + (size_t)cell * N;
            const int sat = cell / mag.n_bins, bin = cell % mag.n_bins;
            // the scratch image holds the LAST dwell's own magnitudes of one bin whenever that dwell accumulated (the second of a fused pair always does)
            float* tmp = (mag.tmp && (EPI == ACQ_EPI_MAG_ACC || PAIR) && bin == mag.tmp_bin) ? mag.tmp + (size_t)sat * N : nullptr;
            MaxPair best = {-1.0f, 0xffffffffu};
            if (active)
                {
                    // non-coherent accumulation: all previous grid values are fetched before the first store (the compiler
                    // cannot prove that g[idx(k)] and g[idx(k')] differ, so a load after a store would wait for it: N1 serial
                    // round trips made the accumulating launches 2.5x slower than the first dwell's)
                    constexpr bool HAS_PREV = (EPI == ACQ_EPI_MAG_ACC || EPI == ACQ_EPI_MAG2_ACC);  // whether the grid holds earlier dwells is part of the instantiation:
                    // the previous values are fetched up front (25 registers, the third instead of the fourth wave per SIMD) only where they exist
                    float prev[HAS_PREV ? N1 : 1];
#pragma unroll
                    for (int k = 0; k < (HAS_PREV ? N1 : 0); k++)
                        {
                            const int idx = n2 + N2 * k - mag.offset;
                            prev[k] = (idx >= 0 && idx < mag.eff) ? g[idx] : 0.0f;
                        }
#pragma unroll
                    for (int k = 0; k < N1; k++)
                        {
                            const int m = n2 + N2 * k;   // natural output index
                            const int idx = m - mag.offset;  // grid column
                            if (idx >= 0 && idx < mag.eff)
                                {
                                    float p = res[k].x * res[k].x + res[k].y * res[k].y;
                                    float val = p;
                                    if (PAIR)
                                        {
                                            // dwell 1: grid (+)= p1; dwell 2: grid += p, in the order two separate passes add
                                            const float first = p1s[k * ACQ_THREADS + threadIdx.x];
                                            val = (EPI == ACQ_EPI_MAG2_ACC ? prev[HAS_PREV ? k : 0] + first : first) + p;
                                            if (tmp) tmp[idx] = p;
                                        }
                                    else if (EPI == ACQ_EPI_MAG_ACC)
                                        {
                                            if (tmp) tmp[idx] = p;
                                            val = prev[HAS_PREV ? k : 0] + p;
                                        }
#if ACQ_COLS_NT & 2
                                    __builtin_nontemporal_store(val, &g[idx]);
#else
                                    g[idx] = val;
#endif
                                    MaxPair c = {val, (unsigned)idx};
                                    best = max_pair(best, c);
                                }
                        }
                }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                {
                    MaxPair o;
                    o.v = __shfl_down(best.v, off, 64);
                    o.i = __shfl_down(best.i, off, 64);
                    best = max_pair(best, o);
                }
            float* sv = svs;
            unsigned* si = reinterpret_cast<unsigned*>(svs + ACQ_THREADS / 64);
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            if (lane == 0)
                {
                    sv[wave] = best.v;
                    si[wave] = best.i;
                }
            __syncthreads();
            if (threadIdx.x == 0)
                {
                    MaxPair b = {sv[0], si[0]};
                    for (int w = 1; w < ACQ_THREADS / 64; w++)
                        {
                            MaxPair c = {sv[w], si[w]};
                            b = max_pair(b, c);
                        }
                    mag.blk_max_val[(size_t)cell * n_xblk + xblk] = b.v;
                    mag.blk_max_idx[(size_t)cell * n_xblk + xblk] = b.i;
                }
        }
}


template <int N1, bool INV, int EPI>
__global__ __launch_bounds__(ACQ_THREADS, (N1 <= 25 ? ((EPI == ACQ_EPI_MAG2 || EPI == ACQ_EPI_MAG2_ACC) ? ACQ_COLS_PAIR_WAVES : ACQ_COLS_WAVES) : 1)) void acq_cols_kernel(AcqFftPlan plan, const float2* __restrict__ Q,
    float2* __restrict__ out, AcqMagArgs mag)
{
    constexpr bool PAIR = (EPI == ACQ_EPI_MAG2 || EPI == ACQ_EPI_MAG2_ACC);
    // first dwell's magnitudes wait in LDS (N1 x 256 floats: 25 KB of the 160) while the second column is transformed
    __shared__ float p1[PAIR ? N1 * ACQ_THREADS : 1];
    __shared__ float sv[2 * (ACQ_THREADS / 64)];
    acq_cols_body<N1, INV, EPI>(plan, Q, out, mag, blockIdx.x, gridDim.x, blockIdx.y, p1, sv);
}

#if ACQ_ROWS3_DBG
// TIMING MOCK of a columns-FIRST inverse pass ($GNSSCORR_ACQ_DBG & 32, results meaningless): transform t of a batch = (spectrum t / n_sats,
// satellite t % n_sats); per column the 25 products A[1000 k + n2] * B[1000 k + n2] (two coalesced arrays), the 25-point inverse DFT in
// registers, the inter-pass twiddle from a third array, 8N bytes of complex results stored.  Operands, twiddles and results all live in
// the inter-pass buffer Q (any finite contents do: only the access pattern and the arithmetic are real).
__global__ __launch_bounds__(ACQ_THREADS, 3) void acq_cols_first_mock_kernel(AcqFftPlan plan, float2* __restrict__ Q, int n_sats, int n_spectra)
{
    constexpr int N1 = 25;
    const int N2 = plan.N2, N = plan.N;
    const int n2 = blockIdx.x * ACQ_THREADS + threadIdx.x;
    const int t = blockIdx.y;
    const int spec = t / n_sats, sat = t - spec * n_sats;
    const bool active = n2 < N2;
    const float2* A = Q + (size_t)spec * N;
    const float2* B = Q + (size_t)(n_spectra + sat) * N;
    const float2* W = Q + (size_t)(n_spectra + n_sats) * N;
    float2 v0[N1], v1[N1];
#pragma unroll
    for (int k = 0; k < N1; k++) v0[k] = active ? cmul(A[(size_t)k * N2 + n2], B[(size_t)k * N2 + n2]) : make_float2(0.f, 0.f);
    RegFft<N1, 1, N1, true>::run(v0, v1, plan.w1);
    float2* res = RegFft<N1, 1, N1, true>::result_in_first ? v0 : v1;
    if (active)
        {
            float2* o = Q + (size_t)(n_spectra + n_sats + 1 + t) * N;
#pragma unroll
            for (int k = 0; k < N1; k++) o[(size_t)k * N2 + n2] = cmul_conj(res[k], W[(size_t)k * N2 + n2]);
        }
}
#endif

#ifdef GNSSCORR_EXPERIMENTS  // two measured-slower forms of the inverse transform: a cell kept on its CU, and row / column roles in one launch
// -----------------------------------------------------------------------------------------------------------------------------
// Whole inverse transform of a cell ON CHIP (N = N1 x 1000, N1 <= 25): * conj(FFT(code)), IFFT, |.|^2 (+=), block maxima
// (pcps_acquisition.cc:724-739) with NO inter-pass buffer.  The two-pass form moves every cell's N complex values out to the
// inter-pass buffer and back (16 N bytes per cell: 262 of the 314 MB a batch of 656 cells moves) and its row pass waits on that
// traffic with four workgroups per CU; here a cell's values never leave the CU: one 512-thread workgroup per CU walks its cells,
// a thread owns TWO adjacent columns (N1 complex values each: 100 registers at N1 = 25), the 1000-point rows are transformed five
// at a time (one radix-10 butterfly per thread and stage, three stages through two alternating 40 KB LDS images, like
// acq_rows3_body with V = float), each finished row is handed to the column owners through the same image, and after the last
// row the thread transforms its two columns in registers (RegFft) and runs the column pass's epilogue.  HBM sees the grid only;
// spectra and code spectra (23 MB per search) are re-read from L2 / Infinity Cache.  The next round's 20 inputs per thread are
// requested one round ahead (40 registers), so with two waves per SIMD the loads are behind ~600 vector instructions.
// A cell of a dwell PAIR is two transforms by the same workgroup, the first stored (or added to earlier dwells), the second
// added on top: (grid + first) + second in the order two separate passes add, bit for bit, and each thread re-reads only what it
// wrote itself.
// -----------------------------------------------------------------------------------------------------------------------------
#define ACQ_FUSED_THREADS 512
#define ACQ_FUSED_RPR 5  // rows per round: 500 of the 512 threads hold one butterfly each
struct AcqFusedArgs
{
    const float2* A;   // spectra [spec][N], row-permuted
    const float2* B;   // code spectra [sat][N], row-permuted, conjugated
    const float2* wN2;
    const float2* wN;
    AcqMagArgs mag;
    int n_sats, n_bins;
    int n_tr;          // transforms per grid cell: 1, or 2 (dwell pair: spectrum `bin`, then spectrum n_bins + `bin`)
    int accumulate;    // the grid holds earlier dwells: the first transform adds too
    int n_cells;       // n_sats * n_bins
};
typedef VC<float> FC;

template <int N1>
struct FusedCols
{
    float2 a[N1], b[N1];  // columns 2t and 2t + 1, indexed by row k1
};

// one round: rows k1 = ROUND * 5 + i.  pa / pb: this round's inputs (already in registers); on return they hold the next round's
// (requested from nxtA / nxtB rows, or untouched when `more` is false)
template <int N1, int ROUND>
static __device__ __forceinline__ void acq_fused_round(const AcqFftPlan& plan, const AcqFusedArgs& g, float2* P, float2* Q, float2 (&pa)[10], float2 (&pb)[10],
    FusedCols<N1>& col, const float2* nxtA, const float2* nxtB, bool more, int t)
{
    constexpr int R = 10, N2 = 1000;
    // The thread's position is laundered through an empty asm once per round: the stage twiddles depend on it alone, and a compiler
    // that knows it hoists all 36 of them out of the item loop and then spills them (scratch reloads queue behind the input
    // prefetch in vmcnt order: measured 4x slower than the two-pass form).  Recomputing them costs ~64 instructions per round.
    asm volatile("" : "+v"(t));
    const bool act = t < ACQ_FUSED_RPR * 100;
    const int i = act ? t / 100 : 0, u = t - 100 * i;
    const int k1 = ROUND * ACQ_FUSED_RPR + i;
    const bool row_ok = act && k1 < N1;
    FC a[R], tw[R];
    // ---- stage 1: butterfly q = u of row i; inputs x[u + 100 j] * code; outputs y[10 u + k] ----
    if (row_ok)
        {
            const float2 sd = g.wN2[plan.tw_off[0] + u];  // w_1000^u
#pragma unroll
            for (int j = 0; j < R; j++) a[j] = pk_mul(FC{pa[j].x, pa[j].y}, FC{pb[j].x, pb[j].y});
            pk_dft10<true>(a);
            pk_powers10(FC{sd.x, sd.y}, tw);
            float2* y = P + i * N2 + R * u;
            y[0] = make_float2(a[0].r, a[0].i);
#pragma unroll
            for (int k = 1; k < R; k++)
                {
                    const FC o = pk_tmul<true>(a[k], tw[k]);
                    y[k] = make_float2(o.r, o.i);
                }
        }
    __syncthreads();
    // the next round's inputs: 20 loads that return while this round's stages 2 and 3 run
    if (more && act)
        {
#pragma unroll
            for (int j = 0; j < R; j++) pa[j] = nxtA[(size_t)i * N2 + u + 100 * j];
#pragma unroll
            for (int j = 0; j < R; j++) pb[j] = nxtB[(size_t)i * N2 + u + 100 * j];
        }
    // ---- stage 2: u = 10 q + r; inputs x[u + 100 j]; outputs y[r + 10 k + 100 q] in the other image ----
    if (row_ok)
        {
            const int q = u / 10, r = u - 10 * q;
            const float2 sd = g.wN2[plan.tw_off[1] + q];  // w_100^q
            const float2* x = P + i * N2 + u;
#pragma unroll
            for (int j = 0; j < R; j++)
                {
                    const float2 v = x[100 * j];
                    a[j] = FC{v.x, v.y};
                }
            pk_dft10<true>(a);
            pk_powers10(FC{sd.x, sd.y}, tw);
            float2* y = Q + i * N2 + r + 100 * q;
            y[0] = make_float2(a[0].r, a[0].i);
#pragma unroll
            for (int k = 1; k < R; k++)
                {
                    const FC o = pk_tmul<true>(a[k], tw[k]);
                    y[10 * k] = make_float2(o.r, o.i);
                }
        }
    __syncthreads();
    // ---- stage 3: r = u; inputs x[u + 100 j]; outputs n2 = u + 100 k with the inter-pass twiddle w_N^(k1 n2), back into the first image ----
    if (row_ok)
        {
            const float2 b = g.wN[(size_t)k1 * N2 + u];    // w_N^(k1 u)
            const float2 d = g.wN[(size_t)k1 * N2 + 100];  // w_N^(100 k1)
            const float2* x = Q + i * N2 + u;
#pragma unroll
            for (int j = 0; j < R; j++)
                {
                    const float2 v = x[100 * j];
                    a[j] = FC{v.x, v.y};
                }
            pk_dft10<true>(a);
            pk_powers10(FC{d.x, d.y}, tw);
            const FC bb = {b.x, b.y};
            float2* y = P + i * N2 + u;
#pragma unroll
            for (int k = 0; k < R; k++)
                {
                    const FC o = pk_tmul<true>(a[k], k == 0 ? bb : pk_mul(bb, tw[k]));
                    y[100 * k] = make_float2(o.r, o.i);
                }
        }
    __syncthreads();
    // ---- the finished rows go to their column owners: thread t takes columns 2t, 2t + 1 of the round's rows ----
    if (act)
        {
#pragma unroll
            for (int ii = 0; ii < ACQ_FUSED_RPR; ii++)
                if (ROUND * ACQ_FUSED_RPR + ii < N1)
                    {
                        const float4 v = *reinterpret_cast<const float4*>(P + ii * N2 + 2 * t);
                        col.a[ROUND * ACQ_FUSED_RPR + ii] = make_float2(v.x, v.y);
                        col.b[ROUND * ACQ_FUSED_RPR + ii] = make_float2(v.z, v.w);
                    }
        }
    // no barrier here: the next round's stage 1 writes the OTHER image; its barrier orders these reads before that image is rewritten
}

// column transform + epilogue of the column pass (acq_cols_body's, two columns per thread); ACC: grid += |.|^2
template <int N1, bool ACC>
static __device__ __forceinline__ void acq_fused_epilogue(const AcqFftPlan& plan, const AcqMagArgs& mag, FusedCols<N1>& col, int cell, int t, float* svs,
    float2 (&nxt_a)[10], float2 (&nxt_b)[10], const float2* nxa, const float2* nxb)
{
    constexpr int N2 = 1000;
    const int N = plan.N;
    float pa[N1], pb[N1];
    {
        float2 y[N1];
        RegFft<N1, 1, N1, true>::run(col.a, y, plan.w1);
        const float2* res = RegFft<N1, 1, N1, true>::result_in_first ? col.a : y;
#pragma unroll
        for (int k = 0; k < N1; k++) pa[k] = res[k].x * res[k].x + res[k].y * res[k].y;
        __builtin_amdgcn_sched_barrier(0);
        RegFft<N1, 1, N1, true>::run(col.b, y, plan.w1);
        res = RegFft<N1, 1, N1, true>::result_in_first ? col.b : y;
#pragma unroll
        for (int k = 0; k < N1; k++) pb[k] = res[k].x * res[k].x + res[k].y * res[k].y;
        __builtin_amdgcn_sched_barrier(0);
    }
    // the columns are dead: the next item's first-round inputs are requested here, behind the grid traffic of this epilogue
    if (nxa)
        {
#pragma unroll
            for (int j = 0; j < 10; j++) nxt_a[j] = nxa[100 * j];
#pragma unroll
            for (int j = 0; j < 10; j++) nxt_b[j] = nxb[100 * j];
        }
    float* gr = mag.grid + (size_t)cell * N;
    const float* gin = gr;  // read before written, by the same thread only
    const int sat = cell / mag.n_bins, bin = cell - sat * mag.n_bins;
    float* tmp = (mag.tmp && ACC && bin == mag.tmp_bin) ? mag.tmp + (size_t)sat * N : nullptr;
    MaxPair best = {-1.0f, 0xffffffffu};
    // (laundered like the round's: the 50 grid indices and range predicates below depend on the thread's position alone, and hoisted
    // out of the item loop they occupy ~60 registers for the whole kernel)
    asm volatile("" : "+v"(t));
    const bool act = t < 500;
    const int n2 = 2 * t;
    typedef float nt_f2 __attribute__((ext_vector_type(2)));
    // a thread's candidates come in increasing grid index (column 2t, 2t + 1 of row 0, then of row 1, ...), so "the first maximum
    // wins" (volk_gnsssdr_32f_index_max_32u) is a strict comparison against the best so far
    if (act && mag.offset == 0 && mag.eff == N)
        {
            // the whole row is kept (no bit-transition window): both columns of every row k are an aligned pair inside the grid
            float2 prev[ACC ? N1 : 1];
            if (ACC)
                {
#pragma unroll
                    for (int k = 0; k < N1; k++) prev[k] = *reinterpret_cast<const float2*>(gin + n2 + N2 * k);
                }
#pragma unroll
            for (int k = 0; k < N1; k++)
                {
                    const int idx = n2 + N2 * k;
                    const float v0 = ACC ? prev[ACC ? k : 0].x + pa[k] : pa[k];
                    const float v1 = ACC ? prev[ACC ? k : 0].y + pb[k] : pb[k];
                    __builtin_nontemporal_store(nt_f2{v0, v1}, reinterpret_cast<nt_f2*>(gr + idx));
                    if (tmp) *reinterpret_cast<float2*>(tmp + idx) = make_float2(pa[k], pb[k]);
                    if (v0 > best.v) best = MaxPair{v0, (unsigned)idx};
                    if (v1 > best.v) best = MaxPair{v1, (unsigned)(idx + 1)};
                }
        }
    else if (act)
        {
            // a window [offset, offset + eff) of the row is kept (bit transition): element by element
#pragma nounroll
            for (int k = 0; k < N1; k++)
                {
                    const int idx = n2 + N2 * k - mag.offset;  // grid column of the first of the two
                    float p0 = 0.f, p1 = 0.f;
#pragma unroll
                    for (int kk = 0; kk < N1; kk++)  // register arrays cannot be indexed by a loop counter: select
                        if (kk == k)
                            {
                                p0 = pa[kk];
                                p1 = pb[kk];
                            }
                    if (idx >= 0 && idx < mag.eff)
                        {
                            const float v0 = ACC ? gin[idx] + p0 : p0;
                            __builtin_nontemporal_store(v0, gr + idx);
                            if (tmp) tmp[idx] = p0;
                            if (v0 > best.v) best = MaxPair{v0, (unsigned)idx};
                        }
                    if (idx + 1 >= 0 && idx + 1 < mag.eff)
                        {
                            const float v1 = ACC ? gin[idx + 1] + p1 : p1;
                            __builtin_nontemporal_store(v1, gr + idx + 1);
                            if (tmp) tmp[idx + 1] = p1;
                            if (v1 > best.v) best = MaxPair{v1, (unsigned)(idx + 1)};
                        }
                }
        }
    // maxima of the four 256-column blocks the statistics kernel expects (acq_cols_blocks): block = two waves of column owners
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        {
            MaxPair o;
            o.v = __shfl_down(best.v, off, 64);
            o.i = __shfl_down(best.i, off, 64);
            best = max_pair(best, o);
        }
    float* sv = svs;
    unsigned* si = reinterpret_cast<unsigned*>(svs + ACQ_FUSED_THREADS / 64);
    const int lane = t & 63, wave = t >> 6;
    if (lane == 0)
        {
            sv[wave] = best.v;
            si[wave] = best.i;
        }
    __syncthreads();
    if (t < 4)
        {
            const MaxPair b = max_pair(MaxPair{sv[2 * t], si[2 * t]}, MaxPair{sv[2 * t + 1], si[2 * t + 1]});
            mag.blk_max_val[(size_t)cell * 4 + t] = b.v;
            mag.blk_max_idx[(size_t)cell * 4 + t] = b.i;
        }
    __syncthreads();  // sv / si are free for the next transform
}

template <int N1>
__global__ __launch_bounds__(ACQ_FUSED_THREADS, 2) void acq_inv_fused_kernel(AcqFftPlan plan, AcqFusedArgs g)
{
    constexpr int N2 = 1000, ROUNDS = (N1 + ACQ_FUSED_RPR - 1) / ACQ_FUSED_RPR;
    extern __shared__ float2 sm[];
    float2* W0 = sm;
    float2* W1 = sm + ACQ_FUSED_RPR * N2;
    float* svs = reinterpret_cast<float*>(sm + 2 * ACQ_FUSED_RPR * N2);
    const int t = threadIdx.x;
    const int N = plan.N;
    const bool act = t < ACQ_FUSED_RPR * 100;
    const int i = act ? t / 100 : 0, u = t - 100 * i;
    // Blocks with equal blockIdx % 8 share an XCD (an L2) and start together; a cell re-reads 400 KB of spectra, so WHICH cells run
    // side by side on an XCD decides whether those reads hit its 4 MB L2.  With >= 8 satellites XCD x owns satellites x, x + 8, ...
    // (n_own of them) and its workgroups form a (satellite, lane) grid: lane j of J walks bins j, j + J, ...; at any time the XCD
    // holds n_own code spectra and J signal spectra (4 + 8 of 200 KB each at 32 satellites), each shared by J resp. n_own
    // workgroups.  With fewer satellites the roles swap: XCD x owns bins x, x + 8, ... and every satellite.  A work item is one
    // transform; both transforms of a pair belong to one workgroup, in order.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const bool by_sat = g.n_sats >= 8;
    const int n_own = by_sat ? (g.n_sats - xcd + 7) >> 3 : g.n_sats;   // satellites this XCD works on
    const int n_walk = by_sat ? g.n_bins : (g.n_bins - xcd + 7) >> 3;  // bins every one of them is searched over here
    const int J = n_own > 0 ? per_xcd / n_own : 0;                     // lanes per satellite
    const int own = n_own > 0 ? slot % n_own : 0, lane_j = n_own > 0 ? slot / n_own : 0;
    const int sat = by_sat ? xcd + 8 * own : own;
    // items of this workgroup: (walk index w = lane_j, lane_j + J, ...) x (tr = 0 .. n_tr - 1)
    const int items = (n_own > 0 && lane_j < J && lane_j < n_walk) ? ((n_walk - lane_j + J - 1) / J) * g.n_tr : 0;
    float2 pa[10], pb[10];
    auto rows_of = [&](int item, const float2*& ra, const float2*& rb, int& cell, int& tr) {
        const int c = item / g.n_tr;
        tr = item - c * g.n_tr;
        const int w = lane_j + c * J;
        const int bin = by_sat ? w : xcd + 8 * w;
        cell = sat * g.n_bins + bin;
        ra = g.A + (size_t)(tr * g.n_bins + bin) * N;
        rb = g.B + (size_t)sat * N;
    };
    int item = 0;
    const float2 *ra = nullptr, *rb = nullptr;
    int cell = 0, tr = 0;
    if (item < items)
        {
            rows_of(item, ra, rb, cell, tr);
            if (act)
                {
#pragma unroll
                    for (int j = 0; j < 10; j++) pa[j] = ra[(size_t)i * N2 + u + 100 * j];
#pragma unroll
                    for (int j = 0; j < 10; j++) pb[j] = rb[(size_t)i * N2 + u + 100 * j];
                }
        }
    while (item < items)
        {
            // the item after this one (same cell's second transform, or the workgroup's next cell)
            const int nitem = item + 1;
            const float2 *na = ra, *nb = rb;
            int ncell = cell, ntr = tr;
            const bool have_next = nitem < items;
            if (have_next) rows_of(nitem, na, nb, ncell, ntr);
            FusedCols<N1> col;
#define FUSED_ROUND(G)                                                                                                                       \
    if (G < ROUNDS)                                                                                                                          \
        acq_fused_round<N1, G>(plan, g, (G & 1) ? W1 : W0, (G & 1) ? W0 : W1, pa, pb, col,                                                   \
            ra + (size_t)(G + 1) * ACQ_FUSED_RPR * N2, rb + (size_t)(G + 1) * ACQ_FUSED_RPR * N2, G + 1 < ROUNDS, t);                    \
    __builtin_amdgcn_sched_barrier(0)
            FUSED_ROUND(0);
            FUSED_ROUND(1);
            FUSED_ROUND(2);
            FUSED_ROUND(3);
            FUSED_ROUND(4);
#undef FUSED_ROUND
            const bool acc = (tr > 0) || (g.accumulate != 0);
            const float2* nxa = have_next ? na + (size_t)i * N2 + u : nullptr;
            const float2* nxb = have_next ? nb + (size_t)i * N2 + u : nullptr;
            if (acc)
                acq_fused_epilogue<N1, true>(plan, g.mag, col, cell, t, svs, pa, pb, act ? nxa : nullptr, nxb);
            else
                acq_fused_epilogue<N1, false>(plan, g.mag, col, cell, t, svs, pa, pb, act ? nxa : nullptr, nxb);
            item = nitem;
            ra = na;
            rb = nb;
            cell = ncell;
            tr = ntr;
        }
}

bool acq_inv_fusable(const AcqFftPlan& plan)
{
    return plan.N1 == 25 && plan.N2 == 1000 && plan.n_fac == 3 && plan.fac[0] == 10 && plan.fac[1] == 10 && plan.fac[2] == 10 && acq_cols_blocks(plan) == 4;
}

hipError_t acq_launch_inv_fused(hipStream_t st, const AcqFftPlan& plan, int n_sats, int n_bins, int n_tr, bool accumulate, const float2* A, const float2* B,
    const float2* wN2, const float2* wN, const AcqMagArgs& mag, int n_cus)
{
    if (!acq_inv_fusable(plan) || (n_tr != 1 && n_tr != 2)) return hipErrorInvalidValue;
    AcqFusedArgs g;
    std::memset(&g, 0, sizeof g);
    g.A = A;
    g.B = B;
    g.wN2 = wN2;
    g.wN = wN;
    g.mag = mag;
    g.n_sats = n_sats;
    g.n_bins = n_bins;
    g.n_tr = n_tr;
    g.accumulate = accumulate ? 1 : 0;
    g.n_cells = n_sats * n_bins;
    const size_t lds = (size_t)2 * ACQ_FUSED_RPR * 1000 * sizeof(float2) + 2 * (ACQ_FUSED_THREADS / 64) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done)
        {
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&acq_inv_fused_kernel<25>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) return ea;
            attr_done = true;
        }
    // one workgroup per CU (its registers and LDS leave room for one), a multiple of 8 so that every XCD gets the same share;
    // an XCD with fewer cells than workgroups leaves the surplus idle
    int grid = (n_cus > 0 ? n_cus : 256) & ~7;
    const int max_per_xcd = n_sats >= 8 ? ((n_sats + 7) / 8) * n_bins : n_sats * ((n_bins + 7) / 8);
    if (grid / 8 > max_per_xcd) grid = 8 * max_per_xcd;
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL(acq_inv_fused_kernel<25>, dim3(grid), dim3(ACQ_FUSED_THREADS), lds, st, plan, g);
    return hipGetLastError();
}

// Inverse row pass of one satellite batch and the two-dwell column pass of the PREVIOUS batch in one launch sized to the chip
// (4 workgroups per CU): three of every four workgroups of an XCD walk the row groups, the fourth walks the column blocks.  The
// row pass is bound by instruction issue and the LDS pipe and loses nothing with three workgroups per CU instead of four; the
// column pass is bound by bandwidth the row pass leaves unused.  No dependency inside the launch: the columns read the buffer the
// previous launch wrote, the rows write the other one.
#ifndef ACQ_ROLE_COLS
#define ACQ_ROLE_COLS 1
#endif
template <int EPI>
__global__ __launch_bounds__(ACQ_THREADS, ACQ_ROWS3_WAVES) void acq_rows3_cols_kernel(AcqFftPlan plan, AcqRows2Args g, const float2* __restrict__ Qc,
    AcqMagArgs mag, int n_cells_c, int n_xblk)
{
    extern __shared__ float2 sm[];
    const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;  // per_xcd is a multiple of 4
    constexpr int CS = ACQ_ROLE_COLS;  // column workgroups of every four
    if ((s & 3) < 4 - CS)
        acq_rows3_body<true>(plan, g, sm, xcd, (s >> 2) * (4 - CS) + (s & 3), per_xcd / 4 * (4 - CS));
    else
        {
            float* lds = reinterpret_cast<float*>(sm);
            const int cw_xcd = per_xcd / 4 * CS, n_cw = 8 * cw_xcd, items = n_cells_c * n_xblk;
#pragma nounroll
            for (int it = xcd * cw_xcd + (s >> 2) * CS + ((s & 3) - (4 - CS)); it < items; it += n_cw)
                {
                    const int cell = it / n_xblk;
                    acq_cols_body<25, true, EPI>(plan, Qc, nullptr, mag, it - cell * n_xblk, n_xblk, cell, lds, lds + 25 * ACQ_THREADS);
                    __syncthreads();
                }
        }
}

#endif  // GNSSCORR_EXPERIMENTS

// ---- permutation (gather): out[a*N2 + b] = in[a + N1*b] (* mul[...]) ----
__global__ void acq_permute_kernel(const float2* __restrict__ in, const float2* __restrict__ mul,
    float2* __restrict__ out, int N, int N1, int N2, int n_valid, size_t in_stride, size_t mul_stride, size_t out_stride)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    const int arr = blockIdx.y;
    if (o >= N) return;
    const int a = o / N2, b = o % N2;
    const int src = a + N1 * b;
    float2 v = make_float2(0.f, 0.f);
    if (src < n_valid)
        {
            v = in[(size_t)arr * in_stride + src];
            if (mul)
                {
                    // volk_32fc_x2_multiply_32fc(in, wipeoff) (pcps_acquisition.cc:717)
                    float2 w = mul[(size_t)arr * mul_stride + src];
                    v = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
                }
        }
    out[(size_t)arr * out_stride + o] = v;
}

// The same gather through LDS: the sources of 64 consecutive b and all a, src = a + N1*b, are ONE contiguous range of
// 64*N1 elements, and out[a*N2 + b0 .. b0+63] is contiguous for each a: both sides coalesced (the plain kernel reads with a
// stride of N1 elements: 14 us per dwell for 41 bins of N = 25000, 4 us this way).
#define ACQ_PERM_TB 64
__global__ __launch_bounds__(256) void acq_permute_tiled_kernel(const float2* __restrict__ in, const float2* __restrict__ mul,
    float2* __restrict__ out, int N, int N1, int N2, int n_valid, size_t in_stride, size_t mul_stride, size_t out_stride)
{
    extern __shared__ float2 tile[];  // [ACQ_PERM_TB * N1]
    const int arr = blockIdx.y;
    const int b0 = blockIdx.x * ACQ_PERM_TB;
    const int nb = min(ACQ_PERM_TB, N2 - b0);
    const int base = N1 * b0, count = N1 * nb;
    for (int i = threadIdx.x; i < count; i += 256)
        {
            const int src = base + i;
            float2 v = make_float2(0.f, 0.f);
            if (src < n_valid)
                {
                    v = in[(size_t)arr * in_stride + src];
                    if (mul)
                        {
                            // volk_32fc_x2_multiply_32fc(in, wipeoff) (pcps_acquisition.cc:717)
                            const float2 w = mul[(size_t)arr * mul_stride + src];
                            v = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
                        }
                }
            tile[i] = v;
        }
    __syncthreads();
    const int j = threadIdx.x & (ACQ_PERM_TB - 1);
    for (int a = threadIdx.x / ACQ_PERM_TB; a < N1; a += 256 / ACQ_PERM_TB)
        if (j < nb) out[(size_t)arr * out_stride + (size_t)a * N2 + b0 + j] = tile[a + N1 * j];
}

// ---- Doppler wipe-off table (pcps_acquisition::update_local_carrier, :296-310) ----
// volk_gnsssdr_s32f_sincos_32fc_generic: _phase += phase_inc in float32, out = (cosf, sinf); stored AT ITS ROW-PERMUTED POSITION
// P[bin][a][b] = wipe[bin][a + N1 b] of the table the forward row pass reads.  (Rounds 1-3 built the table in natural order from a
// sequential phase kernel, permuted it into a staging buffer and copied it back device-to-device; nothing is built in place any more,
// there is no scratch array and no copy engine takes part in a set-up call.)
// The whole table in ONE kernel: the running phase of sample n comes from the bin's arithmetic-progression segments
// (acq_phase_segments.h: the float32 running sum in closed form, bit for bit), so every sample is independent -- no sequential lane
// per bin (0.56 ms for N = 25000), no phase scratch.  seg_off[bin] .. seg_off[bin + 1]: the bin's segments, sorted by first sample.
__global__ __launch_bounds__(256) void acq_wipeoff_segments_kernel(const AcqPhaseSeg* __restrict__ segs, const int* __restrict__ seg_off,
    float2* __restrict__ out, int N, int N1, int N2)
{
    extern __shared__ float2 tile[];  // [ACQ_PERM_TB * N1]
    const int bin = blockIdx.y;
    const int b0 = blockIdx.x * ACQ_PERM_TB;
    const int nb = min(ACQ_PERM_TB, N2 - b0);
    const int base = N1 * b0, count = N1 * nb;
    const AcqPhaseSeg* sg = segs + seg_off[bin];
    const int n_segs = seg_off[bin + 1] - seg_off[bin];
    for (int i = threadIdx.x; i < count; i += 256)
        {
            const int n = base + i;
            int lo = 0, hi = n_segs - 1;
            while (lo < hi)
                {
                    const int mid = (lo + hi + 1) >> 1;
                    if (sg[mid].i0 <= n)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
            const float ph = (float)(sg[lo].p0 + (double)(n - sg[lo].i0) * sg[lo].d);  // exact in double, representable in float
            float sn, cs;
            sincosf(ph, &sn, &cs);
            tile[i] = make_float2(cs, sn);
        }
    __syncthreads();
    const int j = threadIdx.x & (ACQ_PERM_TB - 1);
    for (int a = threadIdx.x / ACQ_PERM_TB; a < N1; a += 256 / ACQ_PERM_TB)
        if (j < nb) out[(size_t)bin * N + (size_t)a * N2 + b0 + j] = tile[a + N1 * j];
}

// ---- integer input samples -> gr_complex (volk_gnsssdr_16ic_convert_32fc at the head of acquisition_core,
// pcps_acquisition.cc:676-679): plain casts ----
template <typename T>
__global__ void acq_convert_kernel(const T* __restrict__ in, float2* __restrict__ out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float2((float)in[2 * i], (float)in[2 * i + 1]);
}

hipError_t acq_launch_convert(hipStream_t st, int iq_format, const void* in, float2* out, int n)
{
    dim3 grid((n + 255) / 256);
    if (iq_format == GC_IQ_I16)
        hipLaunchKernelGGL(acq_convert_kernel<short>, grid, dim3(256), 0, st, static_cast<const short*>(in), out, n);
    else if (iq_format == GC_IQ_I8)
        hipLaunchKernelGGL(acq_convert_kernel<signed char>, grid, dim3(256), 0, st, static_cast<const signed char*>(in), out, n);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- input power (pcps_acquisition.cc:703-709) ----
__global__ __launch_bounds__(1024) void acq_input_power_kernel(const float2* __restrict__ x, int n_valid, int N,
    float* __restrict__ out_power, float* __restrict__ tmp_all, int n_sats, size_t tmp_stride)
{
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < N; i += 1024)
        {
            float p = 0.f;
            if (i < n_valid)
                {
                    float2 v = x[i];
                    p = v.x * v.x + v.y * v.y;
                }
            acc += p;
            // d_tmp_buffer holds |x|^2 after this step (:706); mirrored for every satellite
            if (tmp_all)
                for (int s = 0; s < n_sats; s++) tmp_all[(size_t)s * tmp_stride + i] = p;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        {
            float t = 0.f;
            for (int w = 0; w < 16; w++) t += red[w];
            *out_power = t / (float)N;
        }
}

// ---- final statistics ----
// The peak search over the per-block row maxima is cheap (n_bins * n_blocks entries) and done by every workgroup; the second-peak
// scan of the peak's row (N elements, ~35 instructions each) is cut into ACQ_FINAL_SPLIT pieces, one workgroup each, so that a
// satellite's statistic is not the work of a single CU (one 1024-thread workgroup per satellite took 18-20 us for N = 25000, on 32
// of the chip's 256 CUs).  The last workgroup of a satellite to finish combines the pieces (ticket counter in global memory, reset
// for the next launch; release / acquire fences at agent scope around it).
#define ACQ_FINAL_SPLIT 8
static_assert(ACQ_FINAL_SPLIT == ACQ_FINAL_PIECES, "acq_kernels.h sizes the hand-over buffer");
#define ACQ_FINAL_THREADS 256
struct MaxKey
{
    float v;
    unsigned long long k;  // row * N + index: orders equal values like the reference's scan
};
static __device__ __forceinline__ MaxKey max_key(MaxKey a, MaxKey b)
{
    return (b.v > a.v || (b.v == a.v && b.k < a.k)) ? b : a;
}

__global__ __launch_bounds__(ACQ_FINAL_THREADS) void acq_final_kernel(AcqFinalArgs a)
{
    const int sat = blockIdx.x;
    const int piece = blockIdx.y;
    const int N = a.fft_size;
    const int tid = threadIdx.x;
    constexpr int NW = ACQ_FINAL_THREADS / 64;
    __shared__ float s_peak;
    __shared__ unsigned s_row, s_time;
    __shared__ float sv[NW];
    __shared__ unsigned long long sk[NW];
    __shared__ unsigned si[NW];
    {
        // Global maximum over the per-block row maxima.  The reference scans rows in increasing Doppler
        // with a strict '>' on the row maxima (pcps_acquisition.cc:575-585 / :611-621), each row maximum
        // being the FIRST maximum of its row: the winner is the largest value, ties going to the
        // smallest (row, index).
        MaxKey b = {-1.0f, ~0ull};
        const int total = a.n_bins * a.n_blocks;
        for (int i = tid; i < total; i += ACQ_FINAL_THREADS)
            {
                const int d = i / a.n_blocks;
                const size_t e = ((size_t)sat * a.n_bins + d) * a.n_blocks + (i % a.n_blocks);
                const unsigned idx = a.blk_max_idx[e];
                const float val = a.blk_max_val[e];
                if (idx == 0xffffffffu) continue;  // block without kept samples
                MaxKey c = {val, (unsigned long long)d * (unsigned long long)N + idx};
                b = max_key(b, c);
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            {
                MaxKey o;
                o.v = __shfl_down(b.v, off, 64);
                o.k = __shfl_down(b.k, off, 64);
                b = max_key(b, o);
            }
        if ((tid & 63) == 0)
            {
                sv[tid >> 6] = b.v;
                sk[tid >> 6] = b.k;
            }
        __syncthreads();
        if (tid == 0)
            {
                for (int w = 1; w < NW; w++)
                    {
                        MaxKey c = {sv[w], sk[w]};
                        b = max_key(b, c);
                    }
                // grid_maximum starts at 0.0 with a strict '>': an all-zero grid keeps row 0 / index 0
                if (!(b.v > 0.0f))
                    {
                        b.v = 0.0f;
                        b.k = 0;
                    }
                s_peak = b.v;
                s_row = (unsigned)(b.k / (unsigned long long)N);
                s_time = (unsigned)(b.k % (unsigned long long)N);
            }
        __syncthreads();
    }
    const float peak = s_peak;
    const unsigned row = s_row, tim = s_time;
    gc_acq_result r;
    r.indext = tim;
    r.doppler_index = row;
    if (!a.step_two)
        r.doppler_hz = -a.doppler_max + a.doppler_step * (int)row;
    else
        r.doppler_hz = (int)(a.center_step_two + ((float)row - (float)floor(a.n_bins_step2 / 2.0)) * a.doppler_step2);
    r.mag = peak;
    r.input_power = a.input_power ? *a.input_power : 0.0f;
    r.second_peak = 0.0f;
    r.second_peak_full_row = 0.0f;
    r.test_statistics = 0.0f;
    // :764-768
    r.acq_delay_samples = (double)fmodf((float)tim, a.samples_per_code);
    r.acq_doppler_hz = (double)r.doppler_hz;
    if (a.use_cfar)
        {
            // max_to_input_power_statistic (:571,594-595)
            float nf = (float)N * (float)N;
            float magt = peak / (nf * nf);
            r.test_statistics = magt / r.input_power;
            if (tid == 0 && piece == 0) a.results[sat] = r;
            return;
        }
    // first_vs_second_peak_statistic (:627-664)
    int e1 = (int)tim - a.samples_per_chip;
    int e2 = (int)tim + a.samples_per_chip;
    if (e1 < 0)
        e1 = N + e1;
    else if (e2 >= N)
        e2 = e2 - N;
    int len = e2 - e1;  // the do-while clears e1, e1+1, ... (circular) up to but excluding e2
    if (len <= 0) len += N;
    const float* grow = a.grid + ((size_t)sat * a.n_bins + row) * N;
    float* tmp = a.tmp + (size_t)sat * N;
    // memcpy(d_tmp_buffer, row, d_fft_size) copies d_fft_size BYTES = N/4 floats (:647)
    const int n_copied = N / 4;
    MaxPair best_bug = {-1.0f, 0xffffffffu}, best_full = {-1.0f, 0xffffffffu};
    // this workgroup's piece of the row; U elements per thread and step, the 2 U loads issued before the first one is used
    const int per_piece = (N + ACQ_FINAL_SPLIT - 1) / ACQ_FINAL_SPLIT;
    const int p0 = piece * per_piece, p1 = min(N, p0 + per_piece);
    constexpr int U = 16;  // N = 25000: 3125 elements per piece, one step (a step is one memory round trip)
    for (int i0 = p0 + tid; i0 < p1; i0 += U * ACQ_FINAL_THREADS)
        {
            float g[U], t[U];
#pragma unroll
            for (int u = 0; u < U; u++)
                {
                    const int i = i0 + u * ACQ_FINAL_THREADS;
                    g[u] = i < p1 ? grow[i] : 0.0f;
                    t[u] = (i < p1 && i >= n_copied) ? tmp[i] : 0.0f;
                }
#pragma unroll
            for (int u = 0; u < U; u++)
                {
                    const int i = i0 + u * ACQ_FINAL_THREADS;
                    if (i >= p1) continue;
                    int d = i - e1;
                    if (d < 0) d += N;
                    const bool excluded = d < len;
                    float vb = (i < n_copied) ? g[u] : t[u];
                    float vf = g[u];
                    if (excluded)
                        {
                            vb = 0.0f;
                            vf = 0.0f;
                        }
                    tmp[i] = vb;  // the scratch keeps these contents for the next call, like d_tmp_buffer
                    MaxPair cb = {vb, (unsigned)i}, cf = {vf, (unsigned)i};
                    best_bug = max_pair(best_bug, cb);
                    best_full = max_pair(best_full, cf);
                }
        }
    float piece_val[2];
    for (int pass = 0; pass < 2; pass++)
        {
            MaxPair b = pass ? best_full : best_bug;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                {
                    MaxPair o;
                    o.v = __shfl_down(b.v, off, 64);
                    o.i = __shfl_down(b.i, off, 64);
                    b = max_pair(b, o);
                }
            __syncthreads();
            if ((tid & 63) == 0)
                {
                    sv[tid >> 6] = b.v;
                    si[tid >> 6] = b.i;
                }
            __syncthreads();
            MaxPair t = {sv[0], si[0]};
            for (int w = 1; w < NW; w++)
                {
                    MaxPair c = {sv[w], si[w]};
                    t = max_pair(t, c);
                }
            piece_val[pass] = t.v;  // only the VALUE of the second peak is used (first / second), not its position
        }
    if (tid == 0)
        {
            // hand the piece over; the workgroup that draws the last ticket combines (all in one lane: store -> release fence ->
            // ticket, ticket -> acquire fence -> loads)
            float* pv = a.part_val + ((size_t)sat * ACQ_FINAL_SPLIT + piece) * 2;
            __hip_atomic_store(pv + 0, piece_val[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pv + 1, piece_val[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned ticket = __hip_atomic_fetch_add(a.part_cnt + sat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == ACQ_FINAL_SPLIT - 1)
                {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    float sb = -1.0f, sf = -1.0f;
                    for (int q = 0; q < ACQ_FINAL_SPLIT; q++)
                        {
                            const float* qv = a.part_val + ((size_t)sat * ACQ_FINAL_SPLIT + q) * 2;
                            sb = fmaxf(sb, __hip_atomic_load(qv + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            sf = fmaxf(sf, __hip_atomic_load(qv + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        }
                    r.second_peak = sb;
                    r.second_peak_full_row = sf;
                    r.test_statistics = peak / r.second_peak;
                    a.results[sat] = r;
                    __hip_atomic_store(a.part_cnt + sat, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                }
        }
}

// -----------------------------------------------------------------------------
// host side: plan + launchers
// -----------------------------------------------------------------------------
static bool factor_rows(int N2, int* fac, int* n_fac)
{
    int n = N2, k = 0;
    const int pref[] = {16, 10, 8, 5, 4, 3, 2};
    while (n > 1)
        {
            int r = 0;
            for (int p : pref)
                if (n % p == 0)
                    {
                        r = p;
                        break;
                    }
            if (!r)
                {
                    // any other prime factor: the generic O(R^2) butterfly (slow for large R, but every length the
                    // LDS can hold is transformed -- the reference's FFTW takes any length)
                    for (int p = 7; (long)p * p <= n; p += 2)
                        if (n % p == 0)
                            {
                                r = p;
                                break;
                            }
                    if (!r) r = n;  // n itself is prime
                }
            if (!r || k >= ACQ_MAX_FACTORS) return false;
            fac[k++] = r;
            n /= r;
        }
    *n_fac = k;
    return true;
}

size_t acq_rows_lds_bytes(const AcqFftPlan& plan) { return (size_t)2 * plan.N2 * sizeof(float2); }

void acq_stage_twiddles(const AcqFftPlan& plan, float2* out)
{
    int n = plan.N2;
    for (int f = 0; f < plan.n_fac; f++)
        {
            const int R = plan.fac[f], m = n / R;
            for (int k = 1; k < R; k++)
                for (int q = 0; q < m; q++)
                    {
                        double ang = -2.0 * M_PI * (double)q * (double)k / (double)n;
                        out[plan.tw_off[f] + (k - 1) * m + q] = make_float2((float)cos(ang), (float)sin(ang));
                    }
            n = m;
        }
    out[plan.N2 - 1] = make_float2(1.f, 0.f);
    for (int i = 0; i < plan.N2; i++)
        {
            double ang = -2.0 * M_PI * (double)i / (double)plan.N2;
            out[plan.N2 + i] = make_float2((float)cos(ang), (float)sin(ang));
        }
}

bool acq_plan_make(AcqFftPlan* plan, int N, size_t lds_limit_bytes)
{
    std::memset(plan, 0, sizeof *plan);
    if (N < 1) return false;
    // N1 candidates (register DFT sizes that are instantiated); prefer rows of ~1000-2000 points:
    // long enough to occupy a 256-thread workgroup, short enough for several workgroups per CU
    const int cands[] = {1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 25, 32, 40, 50};  // 32-50: blocks of 256 k - 512 k samples
    int best = 0;
    long best_cost = -1;
    for (int n1 : cands)
        {
            if (N % n1) continue;
            int n2 = N / n1;
            if ((size_t)2 * n2 * sizeof(float2) > lds_limit_bytes) continue;
            int fac[ACQ_MAX_FACTORS], nf;
            if (!factor_rows(n2, fac, &nf)) continue;
            long cost = labs((long)n2 - 1024);
            if (best_cost < 0 || cost < best_cost)
                {
                    best_cost = cost;
                    best = n1;
                }
        }
    if (!best) return false;
    plan->N = N;
    plan->N1 = best;
    plan->N2 = N / best;
    factor_rows(plan->N2, plan->fac, &plan->n_fac);
    {
        // offsets of the per-stage twiddle tables [k-1][q] (sizes sum to N2 - 1)
        int n = plan->N2, off = 0;
        for (int f = 0; f < plan->n_fac; f++)
            {
                const int R = plan->fac[f], m = n / R;
                plan->tw_off[f] = off;
                off += (R - 1) * m;
                n = m;
            }
    }
    for (int k = 0; k < best; k++)
        {
            double a = -2.0 * M_PI * (double)k / (double)best;
            plan->w1[k] = make_float2((float)cos(a), (float)sin(a));
        }
    return true;
}

hipError_t acq_launch_permute(hipStream_t st, const float2* in, const float2* mul, float2* out,
    const AcqFftPlan& plan, int n_valid, int n_arrays, size_t in_stride, size_t mul_stride, size_t out_stride)
{
    static const bool plain = [] {
        const char* e = gc_exp_env("GNSSCORR_ACQ_PERMUTE");
        return e && std::strcmp(e, "plain") == 0;
    }();
    const size_t lds = sizeof(float2) * ACQ_PERM_TB * (size_t)plan.N1;
    if (!plain && lds <= 48 * 1024)
        {
            dim3 grid((plan.N2 + ACQ_PERM_TB - 1) / ACQ_PERM_TB, n_arrays);
            hipLaunchKernelGGL(acq_permute_tiled_kernel, grid, dim3(256), lds, st, in, mul, out, plan.N, plan.N1, plan.N2, n_valid, in_stride, mul_stride,
                out_stride);
            return hipGetLastError();
        }
    dim3 grid((plan.N + 255) / 256, n_arrays);
    hipLaunchKernelGGL(acq_permute_kernel, grid, dim3(256), 0, st, in, mul, out, plan.N, plan.N1, plan.N2, n_valid,
        in_stride, mul_stride, out_stride);
    return hipGetLastError();
}

static int acq_rows_order()
{
    static const int row_order = [] {
        const char* e = gc_exp_env("GNSSCORR_ACQ_ROW_ORDER");
        return e ? std::atoi(e) : 2;
    }();
    return row_order;
}

#ifdef GNSSCORR_EXPERIMENTS
bool acq_rows_cols_fusable(const AcqFftPlan& plan)
{
    int rpw, iters[ACQ_MAX_FACTORS];
    return plan.N1 == 25 && plan.N2 == 1000 && plan.n_fac == 3 && plan.fac[0] == 10 && plan.fac[1] == 10 && plan.fac[2] == 10 &&
           acq_rows2_config(plan, &rpw, iters) && (size_t)rpw * plan.N2 * sizeof(float2) >= (size_t)(25 * ACQ_THREADS + 2 * (ACQ_THREADS / 64)) * sizeof(float);
}

hipError_t acq_launch_rows_cols(hipStream_t st, const AcqFftPlan& plan, int n_cells, const float2* A, AcqCellMap mapA, const float2* B,
    AcqCellMap mapB, float2* Q, const float2* wN2, const float2* wN, int epilogue, int n_cells_cols, const float2* Qc, const AcqMagArgs& mag)
{
    if (!acq_rows_cols_fusable(plan) || (epilogue != ACQ_EPI_MAG2 && epilogue != ACQ_EPI_MAG2_ACC)) return hipErrorInvalidValue;
    AcqRows2Args g;
    std::memset(&g, 0, sizeof g);
    int iters[ACQ_MAX_FACTORS];
    acq_rows2_config(plan, &g.rpw, iters);
    g.A = A;
    g.mapA = mapA;
    g.B = B;
    g.mapB = mapB;
    g.Q = Q;
    g.wN2 = wN2;
    g.wN = wN;
    g.n_rows = plan.N1 * n_cells;
    g.n_bins = mapA.mod;
    g.n_sats = n_cells / mapA.mod;
    g.n_groups = (g.n_rows + g.rpw - 1) / g.rpw;
    g.sat_fastest = g.n_sats > 1 ? acq_rows_order() : 0;
    const size_t lds2 = (size_t)g.rpw * plan.N2 * sizeof(float2);
    const int n_xblk = acq_cols_blocks(plan);
    dim3 grid(256 * 4);  // 4 workgroups per CU (what the row pass's LDS admits), a multiple of 32
    if (epilogue == ACQ_EPI_MAG2)
        hipLaunchKernelGGL(acq_rows3_cols_kernel<ACQ_EPI_MAG2>, grid, dim3(ACQ_THREADS), lds2, st, plan, g, Qc, mag, n_cells_cols, n_xblk);
    else
        hipLaunchKernelGGL(acq_rows3_cols_kernel<ACQ_EPI_MAG2_ACC>, grid, dim3(ACQ_THREADS), lds2, st, plan, g, Qc, mag, n_cells_cols, n_xblk);
    return hipGetLastError();
}

#endif  // GNSSCORR_EXPERIMENTS

hipError_t acq_launch_rows(hipStream_t st, bool inverse, const AcqFftPlan& plan, int n_cells,
    const float2* A, AcqCellMap mapA, const float2* B, AcqCellMap mapB,
    float2* Q, const float2* wN2, const float2* wN)
{
    static const bool force_wg = [] {
        const char* e = gc_exp_env("GNSSCORR_ACQ_ROWS");
        return e && std::strcmp(e, "wg") == 0;
    }();
    AcqRows2Args g;
    std::memset(&g, 0, sizeof g);
    int iters[ACQ_MAX_FACTORS];
    const AcqRows2Entry* entry = nullptr;
    // the packed kernel understands the engine's cell numbering only: cell = sat * n_bins + bin
    const bool cell_layout_ok = mapA.div == 1 && n_cells % mapA.mod == 0 && (B == nullptr ? n_cells == mapA.mod : (mapB.div == mapA.mod && mapB.mod >= (1 << 30)));
    if (!force_wg && cell_layout_ok && plan.n_fac <= 4 && acq_rows2_config(plan, &g.rpw, iters))
        {
            for (const AcqRows2Entry& e : acq_rows2_registry)
                {
                    bool same = (e.n_stages == plan.n_fac);
                    for (int f = 0; f < plan.n_fac && same; f++) same = (e.ri[f] == R2(plan.fac[f], iters[f]));
                    if (same) entry = &e;
                }
        }
    if (entry)
        {
            g.A = A;
            g.mapA = mapA;
            g.B = B;
            g.mapB = mapB;
            g.Q = Q;
            g.wN2 = wN2;
            g.wN = wN;
            g.n_rows = plan.N1 * n_cells;
            g.n_bins = mapA.mod;
            g.n_sats = n_cells / mapA.mod;
            g.n_groups = (g.n_rows + g.rpw - 1) / g.rpw;
            dim3 grid2((unsigned)((g.n_groups + 7) / 8 * 8));
            const size_t lds2 = (size_t)g.rpw * plan.N2 * sizeof(float2);
            AcqFftPlan plan_arg = plan;
            void* args[] = {&plan_arg, &g};
            static const bool no_pairs = [] {
                const char* e = gc_exp_env("GNSSCORR_ACQ_ROWS");
                return e && std::strcmp(e, "nopair") == 0;  // A/B knob: the plain packed kernel
            }();
            AcqRows2Fn fn = inverse ? entry->inv : entry->fwd;
            if (!no_pairs && entry->pair_fwd) fn = inverse ? entry->pair_inv : entry->pair_fwd;
            static const bool interleaved_pairs = [] {
                const char* e = gc_exp_env("GNSSCORR_ACQ_ROWS");
                return e && std::strcmp(e, "pair2p") == 0;  // A/B knob: the pair kernel with (re, im) packing
            }();
#ifdef GNSSCORR_EXPERIMENTS
            if (interleaved_pairs && entry->pair_fwd)
                fn = inverse ? reinterpret_cast<AcqRows2Fn>(&acq_rows2p_kernel<true, 10, 10, 10>) : reinterpret_cast<AcqRows2Fn>(&acq_rows2p_kernel<false, 10, 10, 10>);
#else
            (void)interleaved_pairs;
#endif
            // row order of the pair kernel: 0 = (bin, sat, k1), 1 = (bin, k1, sat), 2 = (k1, bin, sat) with the last digit running fastest.
            // An XCD walks a contiguous eighth of the rows; with 2 it needs row k1 of every spectrum and of every code at a time (a few
            // hundred KB, read once per launch), with 0 / 1 it sweeps all the codes once per bin and its L2 (4 MB) has dropped them by then.
            g.sat_fastest = (B != nullptr && g.n_sats > 1) ? acq_rows_order() : 0;
            static const int dbg = [] {
                const char* e = gc_exp_env("GNSSCORR_ACQ_DBG");  // phase-elimination timing (ACQ_ROWS3_DBG builds only)
                return e ? std::atoi(e) : 0;
            }();
            g.dbg = dbg;
#if ACQ_ROWS3_DBG
            static unsigned long long* d_ts = nullptr;
            static size_t ts_cap = 0;
            if ((dbg & 64) && inverse)
                {
                    if (ts_cap < (size_t)g.n_groups * 8)
                        {
                            (void)hipFree(d_ts);
                            ts_cap = (size_t)g.n_groups * 8;
                            (void)hipMalloc(&d_ts, ts_cap * sizeof(unsigned long long));
                        }
                    g.ts = d_ts;
                }
#endif
            static const int persist = [] {
                const char* e = gc_exp_env("GNSSCORR_ACQ_PERSIST");  // workgroups per CU of the persistent row launch (0: one block per group)
                return e ? std::atoi(e) : 0;  // measured: with the (bin, k1, sat) order 4 per CU (what the LDS admits) was 2 % faster than a block per group,
                                              // with the (k1, bin, sat) order a block per group is 2-3 % faster (0.311-0.321 vs 0.320-0.327 ms per search)
            }();
            if (persist > 0 && entry->pair_fwd && fn == (inverse ? entry->pair_inv : entry->pair_fwd))
                {
                    const unsigned cap = (unsigned)(persist * 256);
                    if (grid2.x > cap) grid2.x = cap;
                }
#if ACQ_ROWS3_DBG
            if ((dbg & 64) && inverse && g.ts)
                {
                    // TIMING EXPERIMENT: run the launch, fetch the stamps, print where a workgroup's lifetime goes (synchronises the device)
                    hipError_t el = hipLaunchKernel(reinterpret_cast<const void*>(fn), grid2, dim3(ACQ_THREADS), args, lds2, st);
                    (void)hipDeviceSynchronize();
                    static int printed = 0;
                    if (printed++ % 40 == 39)
                        {
                            std::vector<unsigned long long> h((size_t)g.n_groups * 8);
                            (void)hipMemcpy(h.data(), g.ts, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                            double sum[6] = {0, 0, 0, 0, 0, 0};
                            unsigned long long tmin = ~0ull, tmax = 0;
                            for (int i = 0; i < g.n_groups; i++)
                                {
                                    const unsigned long long* t = &h[(size_t)i * 8];
                                    for (int k = 0; k < 6; k++) sum[k] += (double)(t[k + 1] - t[k]);
                                    tmin = std::min(tmin, t[0]);
                                    tmax = std::max(tmax, t[6]);
                                }
                            if (const char* tf = std::getenv("GNSSCORR_ACQ_TS_FILE"))
                                {
                                    if (FILE* f = std::fopen(tf, "wb"))
                                        {
                                            std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
                                            std::fclose(f);
                                        }
                                }
                            std::fprintf(stderr, "rows3 stamps (10 ns ticks, mean over %d workgroups): inputs+stage1 %.0f | barrier1 %.0f | stage2 %.0f | stage3 to stores %.0f | stores issue %.0f | last barrier %.0f ; kernel span %.0f\n",
                                g.n_groups, sum[0] / g.n_groups, sum[1] / g.n_groups, sum[2] / g.n_groups, sum[3] / g.n_groups, sum[4] / g.n_groups, sum[5] / g.n_groups, (double)(tmax - tmin));
                        }
                    return el;
                }
#endif
#if ACQ_ROWS3_THREADS != 256
            if (fn == (inverse ? reinterpret_cast<AcqRows2Fn>(&acq_rows3_kernel<true>) : reinterpret_cast<AcqRows2Fn>(&acq_rows3_kernel<false>)))
                {
                    g.rpw *= ACQ_ROWS3_THREADS / 256;
                    g.n_groups = (g.n_rows + g.rpw - 1) / g.rpw;
                    const dim3 grid3((unsigned)((g.n_groups + 7) / 8 * 8));
                    const size_t lds3 = (size_t)g.rpw * plan.N2 * sizeof(float2);
                    static bool attr_set[2] = {false, false};
                    if (!attr_set[inverse ? 1 : 0])
                        {
                            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
                            if (ea != hipSuccess) return ea;
                            attr_set[inverse ? 1 : 0] = true;
                        }
                    return hipLaunchKernel(reinterpret_cast<const void*>(fn), grid3, dim3(ACQ_ROWS3_THREADS), args, lds3, st);
                }
#endif
            return hipLaunchKernel(reinterpret_cast<const void*>(fn), grid2, dim3(ACQ_THREADS), args, lds2, st);
        }
    dim3 grid(plan.N1, n_cells);
    size_t lds = acq_rows_lds_bytes(plan);
    if (lds > 64 * 1024)
        {
            // rows longer than 2730 points need more than the default 64 KB of dynamic LDS
            hipError_t ea = inverse ? hipFuncSetAttribute(reinterpret_cast<const void*>(&acq_rows_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                    : hipFuncSetAttribute(reinterpret_cast<const void*>(&acq_rows_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) return ea;
        }
    static const int row_threads = [] {
        const char* e = gc_exp_env("GNSSCORR_ACQ_ROW_THREADS");
        int v = e ? std::atoi(e) : 0;
        return (v == 64 || v == 128 || v == 256) ? v : 0;
    }();
    const int nthr = row_threads ? row_threads : ACQ_THREADS;
    if (inverse)
        hipLaunchKernelGGL(acq_rows_kernel<true>, grid, dim3(nthr), lds, st, plan, A, mapA, B, mapB, Q, wN2, wN);
    else
        hipLaunchKernelGGL(acq_rows_kernel<false>, grid, dim3(nthr), lds, st, plan, A, mapA, B, mapB, Q, wN2, wN);
    return hipGetLastError();
}

int acq_cols_blocks(const AcqFftPlan& plan) { return (plan.N2 + ACQ_THREADS - 1) / ACQ_THREADS; }

template <int N1>
static hipError_t launch_cols_n1(hipStream_t st, bool inverse, int epilogue, const AcqFftPlan& plan, dim3 grid,
    const float2* Q, float2* out, const AcqMagArgs& mag)
{
#define LAUNCH(INV, EPI) \
    hipLaunchKernelGGL((acq_cols_kernel<N1, INV, EPI>), grid, dim3(ACQ_THREADS), 0, st, plan, Q, out, mag)
    if (inverse)
        {
            switch (epilogue)
                {
                case ACQ_EPI_COMPLEX: LAUNCH(true, ACQ_EPI_COMPLEX); break;
                case ACQ_EPI_MAG: LAUNCH(true, ACQ_EPI_MAG); break;
                case ACQ_EPI_MAG_ACC: LAUNCH(true, ACQ_EPI_MAG_ACC); break;
                case ACQ_EPI_MAG2: LAUNCH(true, ACQ_EPI_MAG2); break;
                case ACQ_EPI_MAG2_ACC: LAUNCH(true, ACQ_EPI_MAG2_ACC); break;
                default: return hipErrorInvalidValue;
                }
        }
    else
        {
            switch (epilogue)
                {
                case ACQ_EPI_COMPLEX: LAUNCH(false, ACQ_EPI_COMPLEX); break;
                case ACQ_EPI_PERM: LAUNCH(false, ACQ_EPI_PERM); break;
                case ACQ_EPI_COMPLEX_CONJ_PERM: LAUNCH(false, ACQ_EPI_COMPLEX_CONJ_PERM); break;
                default: return hipErrorInvalidValue;
                }
        }
#undef LAUNCH
    return hipGetLastError();
}

hipError_t acq_launch_cols(hipStream_t st, bool inverse, int epilogue, const AcqFftPlan& plan, int n_cells,
    const float2* Q, float2* out, const AcqMagArgs* mag)
{
    dim3 grid(acq_cols_blocks(plan), n_cells);
    AcqMagArgs m;
    std::memset(&m, 0, sizeof m);
    if (mag) m = *mag;
#if ACQ_ROWS3_DBG
    {
        static const int dbg = [] {
            const char* e = gc_exp_env("GNSSCORR_ACQ_DBG");
            return e ? std::atoi(e) : 0;
        }();
        if ((dbg & 32) && inverse && mag && plan.N1 == 25 && (epilogue == ACQ_EPI_MAG2 || epilogue == ACQ_EPI_MAG2_ACC))
            {
                // as many transforms as the real pass (2 per cell), but writing 8N each: the buffer must hold operands + results; the
                // mock keeps to the first 400 transforms' worth of space by wrapping the result slot
                const int n_sats = n_cells / m.n_bins, n_spectra = 2 * m.n_bins;
                const int n_tr = 2 * n_cells;
                // results wrap inside the buffer: slot index modulo what fits behind the operands
                (void)n_tr;
                hipLaunchKernelGGL(acq_cols_first_mock_kernel, dim3(acq_cols_blocks(plan), (unsigned)(2 * n_cells - n_spectra - n_sats - 1)), dim3(ACQ_THREADS), 0, st, plan,
                    const_cast<float2*>(Q), n_sats, n_spectra);
                return hipGetLastError();
            }
    }
#endif
    switch (plan.N1)
        {
#define CASE(K) \
    case K: return launch_cols_n1<K>(st, inverse, epilogue, plan, grid, Q, out, m);
            CASE(1)
            CASE(2)
            CASE(3)
            CASE(4)
            CASE(5)
            CASE(6)
            CASE(8)
            CASE(9)
            CASE(10)
            CASE(12)
            CASE(15)
            CASE(16)
            CASE(20)
            CASE(25)
            CASE(32)
            CASE(40)
            CASE(50)
#undef CASE
        default:
            return hipErrorInvalidValue;
        }
}

hipError_t acq_launch_wipeoff_segments(hipStream_t st, const AcqPhaseSeg* segs, const int* seg_off, float2* out, int n_bins, const AcqFftPlan& plan)
{
    const size_t lds = sizeof(float2) * ACQ_PERM_TB * (size_t)plan.N1;
    if (lds > 48 * 1024) return hipErrorInvalidValue;  // N1 <= 50 in every plan acq_plan_make produces: 25.6 KB
    dim3 grid((plan.N2 + ACQ_PERM_TB - 1) / ACQ_PERM_TB, n_bins);
    hipLaunchKernelGGL(acq_wipeoff_segments_kernel, grid, dim3(256), lds, st, segs, seg_off, out, plan.N, plan.N1, plan.N2);
    return hipGetLastError();
}

hipError_t acq_launch_input_power(hipStream_t st, const float2* x, int n_valid, int N, float* out_power, float* tmp_all,
    int n_sats, size_t tmp_stride)
{
    hipLaunchKernelGGL(acq_input_power_kernel, dim3(1), dim3(1024), 0, st, x, n_valid, N, out_power, tmp_all, n_sats, tmp_stride);
    return hipGetLastError();
}

hipError_t acq_launch_final(hipStream_t st, const AcqFinalArgs& a, int n_sats)
{
    hipLaunchKernelGGL(acq_final_kernel, dim3(n_sats, a.use_cfar ? 1 : ACQ_FINAL_SPLIT), dim3(ACQ_FINAL_THREADS), 0, st, a);
    return hipGetLastError();
}
