// trk_chips.hpp -- the plain multicorrelator loop summed per CHIP instead of per sample.
//
// The reference resamples the code to one replica value per sample and tap
// (volk_gnsssdr_32f_xn_resampler_32f_xn.h:77-94) and multiplies every rotated sample with it
// (volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn.h:81-113).  The chip index
//     idx_t(n) = floor((step * (float)n + shift_t) - rem)        (three float32 roundings)
// is non-decreasing in n (every float operation is monotone), so a tap's replica is piecewise constant:
//     sum_n y[n] * code[idx_t(n)]  =  sum_i code[i] * (P[b_t(i+1)] - P[b_t(i)]),
// P[v] = sum of the rotated samples before v, b_t(i) = the first sample whose index reaches i.  Locating b_t(i) EXACTLY
// (an estimate from the real-valued phase, then evaluations of the float32 expression itself on both sides of it until
// they bracket the edge) keeps every sample on the chip the reference would give it, while the per-sample work drops to
// the carrier rotation and a running sum: ~19 wave-instructions per sample instead of 29 for three taps.
//
// One wave owns a contiguous range of 512-sample segments; a lane holds EIGHT CONSECUTIVE samples of a segment:
//   * full segments are loaded coalesced (1 KiB per wave instruction, as the per-sample loop does) and transposed through
//     a swizzled LDS image (conflict-free ds_write_b128 / ds_read_b128); the at most two ragged segments use masked
//     per-sample loads;
//   * sample j of a lane is rotated by the wave-uniform exp(j*k*dtheta), k < 8 (scalar registers) and summed into the
//     lane's local prefix; only the lane total is rotated by the lane's own carrier z0, then scanned across the wave
//     (DPP row shifts / broadcasts): P[8l + k] = E[l] + z0[l] * q[l][k];
//   * q, E and z0 go to LDS; then one lane per (tap, edge of the segment) finds its edge, reads P there, takes the
//     difference to its neighbour's and multiplies with the chip that ended.  The chip still open at the end of a segment is
//     carried to the next one as a (negative) prefix, so a segment needs exactly as many lanes as it has edges: 21 per tap
//     for GPS L1 C/A at 25 Msps, which is what 64 lanes hold for three taps.
#ifndef TRK_CHIPS_HPP
#define TRK_CHIPS_HPP

#define TRK_SEG 512                 // samples per wave iteration
#define TRK_CHIPS_WAVE_FLOATS 1280  // LDS scratch per wave: 8 x 64 float2 (prefix rows; also the transpose staging) + 64 float4
#ifndef TRK_CHIPS_RESYNC
#define TRK_CHIPS_RESYNC 16         // segments between exact re-evaluations of the lanes' carriers
#endif
#ifndef TRK_CHIPS_DIRECT
#define TRK_CHIPS_DIRECT 0          // 1: every lane loads its 64 contiguous bytes itself (no LDS transpose)
#endif

#define GC_DPPF(v, ctrl, row_mask, bank_mask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, bank_mask, false))

// inclusive prefix sum over the 64 lanes of a wave
static __device__ __forceinline__ float wave_scan(float v)
{
    v += GC_DPPF(v, 0x111, 0xf, 0xf);  // row_shr:1
    v += GC_DPPF(v, 0x112, 0xf, 0xf);  // row_shr:2
    v += GC_DPPF(v, 0x114, 0xf, 0xf);  // row_shr:4
    v += GC_DPPF(v, 0x118, 0xf, 0xf);  // row_shr:8: prefix inside every row of 16
    v += GC_DPPF(v, 0x142, 0xa, 0xf);  // row_bcast:15 into rows 1 and 3
    v += GC_DPPF(v, 0x143, 0xc, 0xf);  // row_bcast:31 into rows 2 and 3
    return v;
}
// the value of the previous lane (0 in lane 0)
static __device__ __forceinline__ float wave_prev(float v) { return GC_DPPF(v, 0x138, 0xf, 0xf); }  // wave_shr:1
static __device__ __forceinline__ float lane_bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
static __device__ __forceinline__ float lane_fetch(float v, int src_lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}
// LDS slot of 16-byte piece g of a segment (256 pieces): lane l reads pieces 4l..4l+3, lane l writes pieces j*64 + l
static __device__ __forceinline__ int seg_slot(int g) { return g ^ ((g >> 4) & 3); }

// Segments [g0, g1) of one (channel, epoch, slice); sample n of the epoch is base[n + a], valid for 0 <= n < N (V = N + a).
// scratch: THREADS / 64 * TRK_CHIPS_WAVE_FLOATS floats of LDS, 16-byte aligned.  Adds the lanes' partial sums to accr / acci.
template <int NTAPS, bool WINDOWED, int THREADS, bool DATA>
static __device__ __forceinline__ void trk_loop_chips(const GC_GLOBAL f32x2* __restrict__ base, const float* __restrict__ table,
    const float* __restrict__ table2, int a, int N, int V, int g0, int g1, int lo, int L, float step, float rem,
    const float (&shifts)[NTAPS], double theta0, double dtheta, float* __restrict__ scratch,
    float (&accr)[NTAPS + (DATA ? 1 : 0)], float (&acci)[NTAPS + (DATA ? 1 : 0)])
{
    constexpr int SEG = TRK_SEG, W = THREADS / 64, NT = NTAPS + (DATA ? 1 : 0), LPT = 64 / NT, PT = NTAPS / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: the segment loop runs on scalar registers
    const int spw = (g1 - g0 + W - 1) / W;
    const int gs = g0 + wave * spw, ge = min(g1, gs + spw);
    if (gs >= ge || N <= 0) return;

    if (!(step >= 0.0f && step <= 64.0f))
        {
            // a code that runs backwards is not piecewise constant in the sense used below (and an absurd or non-finite rate
            // would mean millions of edges per segment): per-sample evaluation
            for (int v = max(gs * SEG, a) + lane; v < min(ge * SEG, V); v += 64)
                {
                    const int n = v - a;
                    float zr, zi;
                    carrier_at<false>(n, theta0, dtheta, 0.0, zr, zi);
                    const f32x2 x = base[v];
                    const float yr = fmaf(x.x, zr, -(x.y * zi)), yi = fmaf(x.x, zi, x.y * zr);
                    const float s = step * (float)n;
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        {
                            const int i = floor_to_int((s + shifts[t < NTAPS ? t : PT]) - rem);
                            const float* tb = t < NTAPS ? table : table2;
                            const float cv = WINDOWED ? tb[i - lo] : tb[posmod(i, L)];
                            accr[t] = fmaf(yr, cv, accr[t]);
                            acci[t] = fmaf(yi, cv, acci[t]);
                        }
                }
            return;
        }

    float* wl = scratch + wave * TRK_CHIPS_WAVE_FLOATS;
    f32x2* qrow = reinterpret_cast<f32x2*>(wl);        // [8][64]: row k, lane l = sum of the lane's first k samples (row 0 unused)
    f32x4* stage = reinterpret_cast<f32x4*>(wl);       // the segment's 256 16-byte pieces (aliases qrow: used before it)
    f32x4* rec = reinterpret_cast<f32x4*>(wl + 1024);  // [64]: (E.re, E.im, z0.re, z0.im)

    // this lane's tap: lanes [t*LPT, (t+1)*LPT) serve tap t (the data component's prompt correlator is tap NTAPS)
    int my_t = lane / LPT;
    const bool grp_ok = my_t < NT;
    my_t = min(my_t, NT - 1);
    const int m = lane - my_t * LPT;
    float my_shift = shifts[0];
#pragma unroll
    for (int t = 1; t < NTAPS; t++) my_shift = (my_t == t) ? shifts[t] : my_shift;
    if (DATA) my_shift = (my_t == NTAPS) ? shifts[PT] : my_shift;
    const float* my_tab = (DATA && my_t == NTAPS) ? table2 : table;
    const float* my_tl = my_tab - lo;
    auto code_at = [&](int i) -> float { return WINDOWED ? my_tl[i] : my_tab[posmod(i, L)]; };
    auto idx = [&](int n) -> int {
        const float s = step * (float)n;
        return floor_to_int((s + my_shift) - rem);
    };
    const float inv_step = 1.0f / step;
    const float est_off = rem - my_shift;

    // wave-uniform rotators exp(j*k*dtheta), k < 8, and exp(j*SEG*dtheta): one sincos, spread over the lanes
    float Wr[8], Wi[8], wsr, wsi;
    {
        const int k = lane & 15;
        double t = (double)(k < 8 ? k : SEG) * dtheta * 0.15915494309189533577;
        t -= rint(t);
        float s, c;
        sincosf((float)(t * 6.283185307179586477), &s, &c);
#pragma unroll
        for (int j = 0; j < 8; j++)
            {
                Wr[j] = lane_bcast(c, j);
                Wi[j] = lane_bcast(s, j);
            }
        wsr = lane_bcast(c, 8);
        wsi = lane_bcast(s, 8);
    }
    float z0r, z0i;  // carrier of the lane's first sample of the segment

    auto seg_full = [&](int g) { return g * SEG >= a && (g + 1) * SEG <= V; };
    f32x4 xn[4];
    auto issue = [&](int g) {
        const GC_GLOBAL char* p = reinterpret_cast<const GC_GLOBAL char*>(base) + (size_t)g * (SEG * sizeof(f32x2));
#pragma unroll
        for (int j = 0; j < 4; j++)
            {
#if TRK_CHIPS_DIRECT
                xn[j] = *reinterpret_cast<const GC_GLOBAL f32x4*>(p + (lane * 4 + j) * 16);
#else
                xn[j] = *reinterpret_cast<const GC_GLOBAL f32x4*>(p + (j * 64 + lane) * 16);
#endif
            }
    };
    // LDS addresses of the transpose (constant per lane)
    int wslot[4], rslot[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        {
            wslot[j] = seg_slot(j * 64 + lane);
            rslot[j] = seg_slot(4 * lane + j);
        }

    float pl_r = 0.0f, pl_i = 0.0f;  // prefix at the tap's latest edge, relative to the current segment's start (uniform in a tap's lanes)
    float ar = 0.0f, ai = 0.0f;
    int i_open = 0;

    if (seg_full(gs)) issue(gs);
    for (int g = gs; g < ge; ++g)
        {
            const int sb = g * SEG;
            const int nseg0 = sb - a;  // sample number of the segment's first slot
            const bool full = seg_full(g);
            float xr[8], xi[8];
            if (full)
                {
#if TRK_CHIPS_DIRECT
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        {
                            xr[2 * k] = xn[k].x, xi[2 * k] = xn[k].y;
                            xr[2 * k + 1] = xn[k].z, xi[2 * k + 1] = xn[k].w;
                        }
#else
#pragma unroll
                    for (int j = 0; j < 4; j++) stage[wslot[j]] = xn[j];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        {
                            const f32x4 v = stage[rslot[k]];
                            xr[2 * k] = v.x, xi[2 * k] = v.y;
                            xr[2 * k + 1] = v.z, xi[2 * k + 1] = v.w;
                        }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#endif
                }
            else
                {
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        {
                            const int v = sb + 8 * lane + j;
                            f32x2 s = {0.f, 0.f};
                            if (v >= a && v < V) s = base[v];
                            xr[j] = s.x;
                            xi[j] = s.y;
                        }
                }
            if (g + 1 < ge && seg_full(g + 1)) issue(g + 1);
            if ((g - gs) % TRK_CHIPS_RESYNC == 0) carrier_at<false>(nseg0 + 8 * lane, theta0, dtheta, 0.0, z0r, z0i);

            // ---- local prefix of the lane's samples, each rotated by the wave-uniform part of its carrier ----
            float qr[8], qi[8];
            qr[0] = xr[0];
            qi[0] = xi[0];
#pragma unroll
            for (int j = 1; j < 8; j++)
                {
                    const float ur = fmaf(xr[j], Wr[j], -(xi[j] * Wi[j]));
                    const float ui = fmaf(xr[j], Wi[j], xi[j] * Wr[j]);
                    qr[j] = qr[j - 1] + ur;
                    qi[j] = qi[j - 1] + ui;
                }
            // ---- lane totals in the common frame, prefix over the lanes ----
            const float tr = fmaf(qr[7], z0r, -(qi[7] * z0i));
            const float ti = fmaf(qr[7], z0i, qi[7] * z0r);
            const float sr = wave_scan(tr), si = wave_scan(ti);
            const float er = wave_prev(sr), ei = wave_prev(si);
            const float tot_r = lane_bcast(sr, 63), tot_i = lane_bcast(si, 63);
#pragma unroll
            for (int k = 1; k < 8; k++) qrow[k * 64 + lane] = f32x2{qr[k - 1], qi[k - 1]};
            rec[lane] = f32x4{er, ei, z0r, z0i};
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ---- one lane per (tap, edge) ----
            const int r0 = max(a - sb, 0), r1 = min(V - sb, SEG);  // valid slots of the segment (r1 > r0)
            if (g == gs) i_open = idx(nseg0 + r0);
            const int i_end = idx(nseg0 + r1 - 1);
            const int n_edges = grp_ok ? i_end - i_open : 0;
            for (int p0 = 0; __builtin_amdgcn_ballot_w64(p0 < n_edges) != 0; p0 += LPT)
                {
                    const int mm = p0 + m;
                    const bool act = mm < n_edges;
                    const int i = i_open + 1 + (act ? mm : 0);  // the chip that starts at this lane's edge
                    // first slot whose index reaches i: estimate, then bracket with the exact expression
                    float est = ceilf(((float)i + est_off) * inv_step) - (float)nseg0;
                    est = fminf(fmaxf(est, (float)r0), (float)(r1 - 1));
                    int rel = (int)est;
                    rel = min(max(rel, r0), r1 - 1);
                    for (;;)
                        {
                            const int n = nseg0 + rel;
                            const bool lo_ok = (rel <= r0) || (idx(n - 1) < i);
                            const bool hi_ok = (rel >= r1 - 1) || (idx(n) >= i);
                            const bool done = !act || (lo_ok && hi_ok);
                            if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
                            rel += done ? 0 : (lo_ok ? 1 : -1);
                        }
                    // P[rel] = E[l] + z0[l] * q[l][k]
                    const int lb = rel >> 3, kb = rel & 7;
                    const f32x4 rc = rec[lb];
                    f32x2 q = qrow[kb * 64 + lb];
                    q.x = kb ? q.x : 0.0f;
                    q.y = kb ? q.y : 0.0f;
                    const float pr = rc.x + fmaf(q.x, rc.z, -(q.y * rc.w));
                    const float pi = rc.y + fmaf(q.x, rc.w, q.y * rc.z);
                    // the chip that ENDS here started at the previous lane's edge (first lane: at the tap's latest edge so far)
                    float ppr = wave_prev(pr), ppi = wave_prev(pi);
                    ppr = (m == 0) ? pl_r : ppr;
                    ppi = (m == 0) ? pl_i : ppi;
                    const float cv = code_at(i - 1);
                    const float dr = act ? pr - ppr : 0.0f, di = act ? pi - ppi : 0.0f;
                    ar = fmaf(dr, cv, ar);
                    ai = fmaf(di, cv, ai);
                    // the tap's latest edge: its last active lane of this pass
                    const int cnt = min(n_edges - p0, LPT);
                    const int src = cnt > 0 ? my_t * LPT + cnt - 1 : lane;
                    const float lr = lane_fetch(pr, src), li = lane_fetch(pi, src);
                    pl_r = cnt > 0 ? lr : pl_r;
                    pl_i = cnt > 0 ? li : pl_i;
                }
            i_open = grp_ok ? i_end : i_open;
            // the next segment counts its prefix from its own start
            pl_r -= tot_r;
            pl_i -= tot_i;
            // carrier of the next segment
            const float nzr = fmaf(z0r, wsr, -(z0i * wsi));
            z0i = fmaf(z0r, wsi, z0i * wsr);
            z0r = nzr;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    // the chip still open at the end of the range: everything after its edge
    if (grp_ok && m == 0)
        {
            const float cv = code_at(i_open);
            ar = fmaf(-pl_r, cv, ar);
            ai = fmaf(-pl_i, cv, ai);
        }
#pragma unroll
    for (int t = 0; t < NT; t++)
        {
            accr[t] += (grp_ok && my_t == t) ? ar : 0.0f;
            acci[t] += (grp_ok && my_t == t) ? ai : 0.0f;
        }
}

#endif  // TRK_CHIPS_HPP
