// trk_closed_loop.hip -- closed-loop DLL/PLL tracking entirely on the GPU.
//
// One workgroup per channel runs K code periods back to back: the multicorrelator of
// trk_device.hpp for the epoch, then -- on one lane -- the per-epoch scalar maths the
// reference block does on the host between two correlations:
//   cn0_and_tracking_lock_status   dll_pll_veml_tracking.cc:839-878  (lock_detectors.cc:71-111)
//   run_dll_pll                    dll_pll_veml_tracking.cc:914-973  (tracking_discriminators.cc:41-128,
//                                  tracking_loop_filter.cc:74-245, tracking_FLL_PLL_filter.cc:55-133)
//   update_tracking_vars           dll_pll_veml_tracking.cc:998-1070
//   pull-in alignment              dll_pll_veml_tracking.cc:1568-1600
// so that a launch needs no host round trip per millisecond.  The per-epoch
// record carries what the block puts in Gnss_Synchro and in its binary dump.  The whole state machine of
// general_work runs here: pull-in (1), wide tracking with secondary-code / preamble synchronisation (2,
// :1601-1773), extended coherent integration (3, :1774-1826) and narrow tracking (4, :1827-1896), with the
// data-component prompt correlator of pilot tracking (:899-910) and the high-dynamics rate smoothers (:1016-1064).
#include "gc_internal.h"
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include "gc_stream.h"
#include "trk_device.hpp"
#include <cstring>
#include <vector>

#define LOOP_MAX_CN0 64
#ifndef LOOP_NT
#define LOOP_NT 0  // nontemporal IQ loads: off -- the channels of a launch re-read one RF stream from the caches, period after period
#endif
#ifndef LOOP_WHOLE
#define LOOP_WHOLE 0  // ragged first / last chunks fetched whole where they lie inside the buffer (trk_device.hpp): off -- a lone
                      // workgroup per channel on cache-resident samples gains nothing from it and pays for the masks (0.713 vs 0.689 ms for
                      // 256 channels x 64 periods)
#endif
#ifndef LOOP_ALIGN_PAIRS
#define LOOP_ALIGN_PAIRS 1  // chunk grid of a window: the 16-byte one (the batched kernel's 128-byte grid makes nearly every period start with
                            // a ragged chunk: 0.710 vs 0.689 ms)
#endif
#ifndef LOOP_PF
#define LOOP_PF 2  // 16-byte loads in flight per lane; 4 was measured slower (0.72 vs 0.68 ms for 256 channels x 64 periods): a lone
                   // workgroup per CU is bound by instruction issue at two waves per SIMD, not by loads in flight
#endif
#define LOOP_MAX_SMOOTHER 16
#define LOOP_PI_2 6.283185307179586

// code loop filter without the last integrator (the block constructs it with include_last_integrator = false)
struct DevLoopFilter
{
    float b[4], a[3];
    int nb, na;
    float in[4], out[4];
    int idx;
};

struct DevPll
{
    int order;
    float w, x, a2, a3, b3, w0p, w0p2, w0p3, w0f, w0f2;
};

// gc_loop_sync_conf in device form: the bit patterns are stored newest-symbol-first (bit k = k-th newest symbol), the
// order of the sign shift register they are compared with
struct LoopSync
{
    int extend_symbols, track_pilot, symbols_per_bit, secondary_len, preamble_len;
    float bit_sync_min_time_s, pll_bw_narrow_hz, dll_bw_narrow_hz, el_narrow_chips, vel_narrow_chips;
    unsigned sec_ones[4];   // bit k set: secondary_code[len-1-k] == '1'
    unsigned pre_plus[6];   // bit k set: preamble_symbols[len-1-k] == +1
};

// per-channel persistent state (device memory between launches, LDS during one)
struct LoopChan
{
    gc_loop_conf conf;
    LoopSync sync;
    unsigned hist[6];       // signs of the last prompts, newest in bit 0 (1 = negative real part): d_Prompt_circular_buffer /
                            // d_symbol_history, of which only the signs are ever read
    int hist_count;
    float2 accu[5];         // d_VE_accu, d_E_accu, d_P_accu, d_L_accu, d_VL_accu
    float2 prompt_data;     // d_Prompt_Data
    int current_symbol, extend_count;
    // high dynamics (:1016-1033, :1047-1064): histories of (NCO step, block length), newest last, 2 * smoother_length deep
    double carr_hist[2 * LOOP_MAX_SMOOTHER][2], code_hist[2 * LOOP_MAX_SMOOTHER][2];
    int carr_hist_n, code_hist_n;
    double carrier_phase_rate_step_rad, code_phase_rate_step_chips;
    TrkChan chan;       // iq, code table, taps
    int n_taps;
    int state;          // 0 standby, 1 pull-in, 2 tracking
    int cloop, pull_in_transitory;
    unsigned long long pos;             // stream index of the next unread sample
    unsigned long long sample_counter;  // d_sample_counter
    unsigned long long acq_sample_stamp;
    double acq_code_phase_samples, acq_carrier_doppler_hz;
    double carrier_doppler_hz, code_freq_chips;
    double carrier_phase_step_rad, code_phase_step_chips;
    double rem_code_phase_samples, rem_code_phase_chips, acc_carrier_phase_rad;
    float rem_carr_phase_rad;
    int current_prn_length_samples;
    double current_correlation_time_s;
    double carr_phase_error_hz, carr_freq_error_hz, carr_error_filt_hz, code_error_chips, code_error_filt_chips;
    float2 P_accu_old;
    DevLoopFilter dll;
    DevPll pll;
    float2 prompt_buffer[LOOP_MAX_CN0];
    int cn0_estimation_counter, carrier_lock_fail_counter;
    double carrier_lock_test, cn0_db_hz;
    int lost_lock_events;
};

// ---- loop filter design: tracking_loop_filter.cc:104-245 (include_last_integrator == false) ----
static __device__ __forceinline__ void dll_design(DevLoopFilter& f, int order, float bw, float T)
{
    const float zeta = 1.0 / sqrt(2.0);
    float g1, g2, g3, wn;
    f.nb = f.na = 0;
    switch (order)
        {
        case 1:
            wn = bw * 4.0;
            g1 = wn;
            f.b[0] = g1;
            f.nb = 1;
            break;
        case 3:
            {
                wn = bw / 0.7845;
                const float a3 = 1.1, b3 = 2.4;
                g1 = wn * wn * wn;
                g2 = a3 * wn * wn;
                g3 = b3 * wn;
                f.b[0] = g3 + T / 2.0 * (g2 + T / 2.0 * g1);
                f.b[1] = g1 * T * T / 2.0 - 2.0 * g3;
                f.b[2] = g3 + T / 2.0 * (-g2 + T / 2.0 * g1);
                f.nb = 3;
                f.a[0] = 2.0f;
                f.a[1] = -1.0f;
                f.na = 2;
                break;
            }
        default:
            wn = bw * (8.0 * zeta) / (4.0 * zeta * zeta + 1.0);
            g1 = wn * wn;
            g2 = wn * 2.0 * zeta;
            f.b[0] = (g1 * T / 2.0 + g2);
            f.b[1] = g1 * T / 2.0 - g2;
            f.nb = 2;
            f.a[0] = 1.0f;
            f.na = 1;
            break;
        }
}

static __device__ __forceinline__ void dll_initialize(DevLoopFilter& f)
{
    for (int i = 0; i < 4; i++) f.in[i] = f.out[i] = 0.0f;
    f.idx = 3;
}

static __device__ __forceinline__ float dll_apply(DevLoopFilter& f, float v)
{
    float result = 0.0f;
    for (int i = 0; i < f.na; ++i) result += f.a[i] * f.out[(f.idx + i) % 4];
    f.idx--;
    if (f.idx < 0) f.idx += 4;
    f.in[f.idx] = v;
    for (int i = 0; i < f.nb; ++i) result += f.b[i] * f.in[(f.idx + i) % 4];
    f.out[f.idx] = result;
    return result;
}

// ---- carrier loop filter: tracking_FLL_PLL_filter.cc:55-133 ----
static __device__ __forceinline__ void pll_set_params(DevPll& p, float fll_bw_hz, float pll_bw_hz, int order)
{
    p.order = order;
    p.w = p.x = p.a3 = p.b3 = p.w0p3 = p.w0f2 = 0.0f;
    if (order == 3)
        {
            p.b3 = 2.400f;
            p.a3 = 1.100f;
            p.a2 = 1.414f;
            p.w0p = pll_bw_hz / 0.7845;
            p.w0p2 = p.w0p * p.w0p;
            p.w0p3 = p.w0p2 * p.w0p;
            p.w0f = fll_bw_hz / 0.53;
            p.w0f2 = p.w0f * p.w0f;
        }
    else
        {
            p.a2 = 1.414f;
            p.w0p = pll_bw_hz / 0.53;
            p.w0p2 = p.w0p * p.w0p;
            p.w0f = fll_bw_hz / 0.25;
        }
}

// set_params on a running filter (:1754): new coefficients, integrators untouched
static __device__ __forceinline__ void pll_retune(DevPll& p, float fll_bw_hz, float pll_bw_hz, int order)
{
    const float w = p.w, x = p.x;
    pll_set_params(p, fll_bw_hz, pll_bw_hz, order);
    p.w = w;
    p.x = x;
}

static __device__ __forceinline__ float pll_get_carrier_error(DevPll& p, float fll, float pll, float T)
{
    float carrier_error_hz;
    if (p.order == 3)
        {
            p.w = p.w + T * (p.w0p3 * pll + p.w0f2 * fll);
            p.x = p.x + T * (0.5 * p.w + p.a2 * p.w0f * fll + p.a3 * p.w0p2 * pll);
            carrier_error_hz = 0.5 * p.x + p.b3 * p.w0p * pll;
        }
    else
        {
            const float w_new = p.w + pll * p.w0p2 * T + fll * p.w0f * T;
            carrier_error_hz = 0.5 * (w_new + p.w) + p.a2 * p.w0p * pll;
            p.w = w_new;
        }
    return carrier_error_hz;
}

static __device__ double cabs_d(float2 v) { return (double)hypotf(v.x, v.y); }  // std::abs(gr_complex) is float hypot

// start_tracking (dll_pll_veml_tracking.cc:549-747), loop part
static __device__ void loop_start(LoopChan& s)
{
    const gc_loop_conf& c = s.conf;
    s.acq_code_phase_samples = c.acq_delay_samples;
    s.acq_carrier_doppler_hz = c.acq_doppler_hz;
    s.acq_sample_stamp = c.acq_samplestamp_samples;
    s.sample_counter = c.sample_counter;
    s.carrier_doppler_hz = s.acq_carrier_doppler_hz;
    s.carrier_phase_step_rad = LOOP_PI_2 * s.carrier_doppler_hz / c.fs_in;
    pll_set_params(s.pll, c.fll_bw_hz, c.pll_bw_hz, c.pll_filter_order);
    if (s.pll.order == 3)
        {
            s.pll.x = 2.0 * (float)s.acq_carrier_doppler_hz;
            s.pll.w = 0;
        }
    else
        {
            s.pll.w = (float)s.acq_carrier_doppler_hz;
            s.pll.x = 0;
        }
    dll_design(s.dll, c.dll_filter_order, c.dll_bw_hz, (float)c.code_period_s);
    dll_initialize(s.dll);
    s.carrier_lock_fail_counter = 0;
    s.rem_code_phase_samples = 0.0;
    s.rem_carr_phase_rad = 0.0f;
    s.rem_code_phase_chips = 0.0;
    s.acc_carrier_phase_rad = 0.0;
    s.cn0_estimation_counter = 0;
    s.carrier_lock_test = 1.0;
    s.cn0_db_hz = 0.0;
    s.current_correlation_time_s = c.code_period_s;
    s.code_freq_chips = c.code_chip_rate_hz;
    s.code_phase_step_chips = s.code_freq_chips / c.fs_in;
    s.current_prn_length_samples = (int)c.vector_length;
    s.P_accu_old = make_float2(0.f, 0.f);
    s.carr_phase_error_hz = s.carr_freq_error_hz = s.carr_error_filt_hz = s.code_error_chips = s.code_error_filt_chips = 0.0;
    s.state = 1;
    s.cloop = 1;
    s.pull_in_transitory = 1;
    s.lost_lock_events = 0;
    for (int t = 0; t < 5; t++) s.accu[t] = make_float2(0.f, 0.f);
    s.prompt_data = make_float2(0.f, 0.f);
    s.current_symbol = 0;
    s.extend_count = 0;
    s.hist_count = 0;
    for (int i = 0; i < 6; i++) s.hist[i] = 0u;
    s.carrier_phase_rate_step_rad = s.code_phase_rate_step_chips = 0.0;
    s.carr_hist_n = s.code_hist_n = 0;
}

// cn0_and_tracking_lock_status (:839-878); false = loss of lock
static __device__ __forceinline__ bool loop_lock_status(LoopChan& s, double coh_integration_time_s)
{
    const gc_loop_conf& c = s.conf;
    if (s.cn0_estimation_counter < c.cn0_samples)
        {
            s.prompt_buffer[s.cn0_estimation_counter] = s.accu[2];
            s.cn0_estimation_counter++;
            return true;
        }
    s.cn0_estimation_counter = 0;
    double Psig = 0.0, Ptot = 0.0;
    float sum_I = 0.f, sum_Q = 0.f;
    for (int i = 0; i < c.cn0_samples; i++)
        {
            const float2 v = s.prompt_buffer[i];
            Psig += fabs((double)v.x);
            Ptot += (double)v.y * (double)v.y + (double)v.x * (double)v.x;
            sum_I += v.x;
            sum_Q += v.y;
        }
    Psig /= (double)c.cn0_samples;
    Psig = Psig * Psig;
    Ptot /= (double)c.cn0_samples;
    const double SNR = Psig / (Ptot - Psig);
    s.cn0_db_hz = (double)(float)(10.0 * log10(SNR) - 10.0 * log10(coh_integration_time_s));
    const float NBP = sum_I * sum_I + sum_Q * sum_Q, NBD = sum_I * sum_I - sum_Q * sum_Q;
    s.carrier_lock_test = (double)(NBD / NBP);
    if (!s.pull_in_transitory)
        {
            if (s.carrier_lock_test < c.carrier_lock_th || s.cn0_db_hz < c.cn0_min)
                s.carrier_lock_fail_counter++;
            else if (s.carrier_lock_fail_counter > 0)
                s.carrier_lock_fail_counter--;
        }
    if (s.carrier_lock_fail_counter > c.max_lock_fail)
        {
            s.lost_lock_events++;  // message 3 on the "events" port
            s.carrier_lock_fail_counter = 0;
            return false;
        }
    return true;
}

// clear_tracking_vars (:976-995)
static __device__ __forceinline__ void loop_clear_tracking_vars(LoopChan& s)
{
    s.prompt_data = make_float2(0.f, 0.f);
    s.P_accu_old = make_float2(0.f, 0.f);
    s.carr_phase_error_hz = s.carr_freq_error_hz = s.carr_error_filt_hz = 0.0;
    s.code_error_chips = s.code_error_filt_chips = 0.0;
    s.current_symbol = 0;
    s.hist_count = 0;
    for (int i = 0; i < 6; i++) s.hist[i] = 0u;
    s.carrier_phase_rate_step_rad = s.code_phase_rate_step_chips = 0.0;
    s.carr_hist_n = s.code_hist_n = 0;
}

// run_dll_pll (:914-973) on the accumulators
static __device__ __forceinline__ void loop_run_dll_pll(LoopChan& s, bool veml)
{
    const gc_loop_conf& c = s.conf;
    const float2 VE = s.accu[0], E = s.accu[1], P = s.accu[2], L = s.accu[3], VL = s.accu[4];
    if (s.cloop)
        s.carr_phase_error_hz = ((P.x != 0.0f) ? (double)atanf(P.y / P.x) : 0.0) / LOOP_PI_2;
    else
        s.carr_phase_error_hz = (double)atan2f(P.y, P.x) / LOOP_PI_2;
    if ((s.pull_in_transitory && c.enable_fll_pull_in) || c.enable_fll_steady_state)
        {
            const double dot = s.P_accu_old.x * P.x + s.P_accu_old.y * P.y;
            const double cross = s.P_accu_old.x * P.y - P.x * s.P_accu_old.y;
            s.carr_freq_error_hz = atan2(cross, dot) / (s.current_correlation_time_s - 0.0) / LOOP_PI_2;
            s.P_accu_old = P;
            if (s.pull_in_transitory && c.enable_fll_pull_in)
                s.carr_error_filt_hz = pll_get_carrier_error(s.pll, (float)s.carr_freq_error_hz, 0.0f, (float)s.current_correlation_time_s);
            else
                s.carr_error_filt_hz = pll_get_carrier_error(s.pll, (float)s.carr_freq_error_hz, (float)s.carr_phase_error_hz, (float)s.current_correlation_time_s);
        }
    else
        s.carr_error_filt_hz = pll_get_carrier_error(s.pll, 0.0f, (float)s.carr_phase_error_hz, (float)s.current_correlation_time_s);
    s.carrier_doppler_hz = s.carr_error_filt_hz;
    if (veml)
        {
            const double pe = sqrt((double)(VE.x * VE.x + VE.y * VE.y) + (double)(E.x * E.x + E.y * E.y));
            const double pl = sqrt((double)(VL.x * VL.x + VL.y * VL.y) + (double)(L.x * L.x + L.y * L.y));
            s.code_error_chips = (pe + pl == 0.0) ? 0.0 : (pe - pl) / (pe + pl);
        }
    else
        {
            const double pe = cabs_d(E), pl = cabs_d(L);
            s.code_error_chips = (pe + pl == 0.0) ? 0.0 : 0.5 * (pe - pl) / (pe + pl);
        }
    s.code_error_filt_chips = dll_apply(s.dll, (float)s.code_error_chips);
    s.code_freq_chips = (1.0 + (s.carrier_doppler_hz / c.signal_carrier_freq_hz)) * c.code_chip_rate_hz - s.code_error_filt_chips;
}

// the rate smoother of both NCOs (:1016-1033, :1047-1064): once 2 * smoother_length (value, samples) pairs are held, the
// rate is (mean of the newer half - mean of the older half) / samples of the newer half
static __device__ __forceinline__ double loop_smoothed_rate(double (*hist)[2], int& count, int sl, double value, double samples, double current)
{
    const int cap = 2 * sl;
    if (count == cap)
        {
            for (int k = 1; k < cap; k++)
                {
                    hist[k - 1][0] = hist[k][0];
                    hist[k - 1][1] = hist[k][1];
                }
            count--;
        }
    hist[count][0] = value;
    hist[count][1] = samples;
    count++;
    if (count < cap) return current;
    double cp1 = 0.0, cp2 = 0.0, ns = 0.0;
    for (int k = 0; k < sl; k++)
        {
            cp1 += hist[k][0];
            cp2 += hist[cap - k - 1][0];
            ns += hist[cap - k - 1][1];
        }
    cp1 /= (double)sl;
    cp2 /= (double)sl;
    return (cp2 - cp1) / ns;
}

// update_tracking_vars (:998-1070)
static __device__ __forceinline__ void loop_update_tracking_vars(LoopChan& s)
{
    const gc_loop_conf& c = s.conf;
    const int sl = (int)c.high_dyn_smoother_length;
    const double T_prn_samples = (1.0 / s.code_freq_chips) * (double)c.code_length_chips * c.fs_in;
    const double K_blk_samples = T_prn_samples + s.rem_code_phase_samples;
    s.current_prn_length_samples = (int)floor(K_blk_samples);
    s.carrier_phase_step_rad = LOOP_PI_2 * s.carrier_doppler_hz / c.fs_in;
    const double n = (double)s.current_prn_length_samples;
    if (sl > 0) s.carrier_phase_rate_step_rad = loop_smoothed_rate(s.carr_hist, s.carr_hist_n, sl, s.carrier_phase_step_rad, n, s.carrier_phase_rate_step_rad);
    s.rem_carr_phase_rad += (float)(s.carrier_phase_step_rad * n + 0.5 * s.carrier_phase_rate_step_rad * n * n);
    s.rem_carr_phase_rad = fmodf(s.rem_carr_phase_rad, (float)LOOP_PI_2);
    s.acc_carrier_phase_rad -= (s.carrier_phase_step_rad * n + 0.5 * s.carrier_phase_rate_step_rad * n * n);
    s.code_phase_step_chips = s.code_freq_chips / c.fs_in;
    if (sl > 0) s.code_phase_rate_step_chips = loop_smoothed_rate(s.code_hist, s.code_hist_n, sl, s.code_phase_step_chips, n, s.code_phase_rate_step_chips);
    s.rem_code_phase_samples = K_blk_samples - n;
    s.rem_code_phase_chips = s.code_freq_chips * s.rem_code_phase_samples / c.fs_in;
}

// save_correlation_results (:1072-1125): accumulate with the secondary-code sign
static __device__ __forceinline__ void loop_save_correlation_results(LoopChan& s, const float2* taps, bool veml)
{
    const LoopSync& y = s.sync;
    float sign = 1.0f;
    if (y.secondary_len > 0)
        {
            // sec_ones is stored newest-first: character i of the string is bit (len - 1 - i)
            const int bit = y.secondary_len - 1 - s.current_symbol;
            if ((y.sec_ones[bit >> 5] >> (bit & 31)) & 1u) sign = -1.0f;
            s.current_symbol = (s.current_symbol + 1) % y.secondary_len;
        }
    else
        {
            s.current_symbol++;
            s.current_symbol = y.symbols_per_bit > 0 ? s.current_symbol % y.symbols_per_bit : 0;
        }
    for (int t = 0; t < 5; t++)
        {
            if (!veml && (t == 0 || t == 4)) continue;
            const float2 v = taps[veml ? t : t - 1];
            if (sign > 0.0f)
                {
                    s.accu[t].x += v.x;
                    s.accu[t].y += v.y;
                }
            else
                {
                    s.accu[t].x -= v.x;
                    s.accu[t].y -= v.y;
                }
        }
    s.cloop = y.track_pilot ? 0 : 1;
}

// pushes the sign of the prompt into the history; true when the last `len` signs match `pattern`
// (exactly == all bits equal; or, for the secondary code, all bits opposite as well)
static __device__ __forceinline__ bool loop_push_and_match(LoopChan& s, float prompt_re, int len, const unsigned* pattern, bool either_polarity)
{
    const unsigned neg = prompt_re < 0.0f ? 1u : 0u;
    for (int w = 5; w > 0; w--) s.hist[w] = (s.hist[w] << 1) | (s.hist[w - 1] >> 31);
    s.hist[0] = (s.hist[0] << 1) | neg;
    if (s.hist_count < len) s.hist_count++;
    if (s.hist_count < len) return false;
    int diff = 0;
    for (int w = 0; w * 32 < len; w++)
        {
            const int nb = min(32, len - w * 32);
            const unsigned mask = nb == 32 ? 0xffffffffu : ((1u << nb) - 1u);
            diff += __popc((s.hist[w] ^ pattern[w]) & mask);
        }
    return either_polarity ? (diff == 0 || diff == len) : diff == len;
}

static __device__ __forceinline__ void loop_write_record(const LoopChan& s, gc_loop_record* rec, int valid, int integrating, int extend_count)
{
    for (int t = 0; t < 5; t++)
        {
            rec->accu[2 * t] = s.accu[t].x;
            rec->accu[2 * t + 1] = s.accu[t].y;
        }
    rec->extend_count = extend_count;
    rec->integrating = integrating;
    rec->valid = valid;
}

// everything general_work does with one code period's correlator outputs (states 2, 3, 4; :1601-1896)
template <int NTAPS, bool DATA>
static __device__ __forceinline__ void loop_after_correlation(LoopChan& s, const float2* taps, gc_loop_record* rec)
{
    const gc_loop_conf& c = s.conf;
    const LoopSync& y = s.sync;
    const bool veml = NTAPS == 5;
    const float2 P = taps[NTAPS / 2];
    {
        unsigned* w = reinterpret_cast<unsigned*>(rec);
        for (unsigned i = 0; i < sizeof(gc_loop_record) / 4; i++) w[i] = 0u;
    }
    for (int t = 0; t < NTAPS; t++)
        {
            rec->corr[2 * t] = taps[t].x;
            rec->corr[2 * t + 1] = taps[t].y;
        }
    s.prompt_data = DATA ? taps[NTAPS] : P;
    const int ext = y.extend_symbols > 1 ? y.extend_symbols : 1;
    if (s.state == 2)
        {
            // single correlation step variables (:1604-1612)
            for (int t = 0; t < 5; t++) s.accu[t] = make_float2(0.f, 0.f);
            for (int t = 0; t < NTAPS; t++) s.accu[veml ? t : t + 1] = taps[t];
            if (!loop_lock_status(s, c.code_period_s))
                {
                    loop_clear_tracking_vars(s);
                    s.state = 0;
                    loop_write_record(s, rec, 0, 0, 0);
                }
            else
                {
                    loop_run_dll_pll(s, veml);
                    loop_update_tracking_vars(s);
                    loop_write_record(s, rec, 1, 0, s.extend_count);  // log_data(false), :1627
                    bool next_state;
                    if (y.secondary_len > 0)
                        {
                            // acquire_secondary (:800-836) over the last secondary_len prompts: '0' <-> positive prompt or the
                            // exact opposite.  A negative prompt on a '0' counts +1: sign bit == 1 and code bit == 0 differ.
                            next_state = loop_push_and_match(s, P.x, y.secondary_len, y.sec_ones, true);
                        }
                    else if (y.symbols_per_bit > 1)
                        {
                            // preamble search after bit_sync_min_time_s of tracking (:1645-1685)
                            next_state = false;
                            const float t_trk = (float)((double)(float)(s.sample_counter - s.acq_sample_stamp) / c.fs_in);
                            if (t_trk > y.bit_sync_min_time_s && y.preamble_len > 0)
                                {
                                    // corr == length  <=>  every symbol's clipped sign equals the preamble's: negative
                                    // (bit 1) where the preamble is -1 (pre_plus bit 0): all bits differ
                                    next_state = loop_push_and_match(s, P.x, y.preamble_len, y.pre_plus, false);
                                }
                        }
                    else
                        next_state = true;
                    if (next_state)
                        {
                            for (int t = 0; t < 5; t++) s.accu[t] = make_float2(0.f, 0.f);
                            s.hist_count = 0;
                            for (int i = 0; i < 6; i++) s.hist[i] = 0u;
                            s.current_symbol = 0;
                            if (ext > 1)
                                {
                                    s.extend_count = 0;
                                    s.current_correlation_time_s = (double)((float)ext * (float)c.code_period_s);
                                    s.state = 3;
                                    // narrow loop filters and taps (:1751-1766)
                                    dll_design(s.dll, c.dll_filter_order, y.dll_bw_narrow_hz, (float)s.current_correlation_time_s);
                                    pll_retune(s.pll, c.fll_bw_hz, y.pll_bw_narrow_hz, c.pll_filter_order);
                                    const float spc = (float)c.code_samples_per_chip;
                                    if (veml)
                                        {
                                            s.chan.shifts[0] = -y.vel_narrow_chips * spc;
                                            s.chan.shifts[1] = -y.el_narrow_chips * spc;
                                            s.chan.shifts[3] = y.el_narrow_chips * spc;
                                            s.chan.shifts[4] = y.vel_narrow_chips * spc;
                                        }
                                    else
                                        {
                                            s.chan.shifts[0] = -y.el_narrow_chips * spc;
                                            s.chan.shifts[2] = y.el_narrow_chips * spc;
                                        }
                                }
                            else
                                s.state = 4;
                        }
                }
        }
    else if (s.state == 3)
        {
            loop_update_tracking_vars(s);
            loop_save_correlation_results(s, taps, veml);
            s.extend_count++;
            if (s.extend_count == ext - 1)
                {
                    s.extend_count = 0;
                    s.state = 4;
                }
            loop_write_record(s, rec, 1, 1, s.extend_count);  // log_data(true), :1824
        }
    else  // state 4
        {
            loop_save_correlation_results(s, taps, veml);
            if (!loop_lock_status(s, c.code_period_s * (double)ext))
                {
                    loop_clear_tracking_vars(s);
                    s.state = 0;
                    loop_write_record(s, rec, 0, 0, 0);
                }
            else
                {
                    loop_run_dll_pll(s, veml);
                    loop_update_tracking_vars(s);
                    loop_write_record(s, rec, 1, 0, s.extend_count);  // log_data(false), :1880
                    for (int t = 0; t < 5; t++) s.accu[t] = make_float2(0.f, 0.f);
                    if (ext > 1) s.state = 3;
                }
        }
    s.sample_counter += (unsigned long long)s.current_prn_length_samples;
    s.pos += (unsigned long long)s.current_prn_length_samples;
    rec->prompt_data[0] = s.prompt_data.x;
    rec->prompt_data[1] = s.prompt_data.y;
    rec->sample_counter = s.sample_counter;
    rec->acc_carrier_phase_rad = s.acc_carrier_phase_rad;
    rec->rem_code_phase_samples = s.rem_code_phase_samples;
    rec->carrier_doppler_hz = (float)s.carrier_doppler_hz;
    rec->code_freq_chips = (float)s.code_freq_chips;
    rec->carr_phase_error_hz = (float)s.carr_phase_error_hz;
    rec->carr_error_filt_hz = (float)s.carr_error_filt_hz;
    rec->code_error_chips = (float)s.code_error_chips;
    rec->code_error_filt_chips = (float)s.code_error_filt_chips;
    rec->cn0_db_hz = (float)s.cn0_db_hz;
    rec->carrier_lock_test = (float)s.carrier_lock_test;
    rec->state = s.state;
    rec->current_prn_length_samples = s.current_prn_length_samples;
}

// What general_work does BEFORE the correlation of a code period (:1552-1600, :886-897), by one lane: end of the pull-in
// transitory, the pull-in sample skip (state 1 -> 2), the decision whether a whole block is available below `limit`, and the
// correlator's scalars narrowed to float exactly where do_correlation_step narrows them.  Returns 0 (and writes the period's
// invalid record) when there is nothing to correlate: standby, or the input is exhausted.
template <bool HD>
static __device__ __forceinline__ int loop_prepare(LoopChan& s, unsigned long long limit, gc_epoch_params& s_p, gc_loop_record* rec)
{
    const gc_loop_conf& c = s.conf;
    int go = 1;
    if (s.pull_in_transitory)
        {
            if (c.pull_in_time_s < (s.sample_counter - s.acq_sample_stamp) / (unsigned long long)(int)c.fs_in) s.pull_in_transitory = 0;
        }
    if (s.state == 1)
        {
            // pull-in (:1568-1600): skip samples until the incoming code is aligned with the replica
            const long long acq_trk_diff_samples = (long long)s.sample_counter - (long long)s.acq_sample_stamp;
            const double delta = (double)acq_trk_diff_samples - s.acq_code_phase_samples;
            s.code_freq_chips = c.code_chip_rate_hz;
            s.code_phase_step_chips = s.code_freq_chips / c.fs_in;
            const double T_prn_mod_samples = (1.0 / s.code_freq_chips) * (double)c.code_length_chips * c.fs_in;
            s.acq_code_phase_samples = T_prn_mod_samples - fmod(delta, T_prn_mod_samples);
            s.current_prn_length_samples = (int)round(T_prn_mod_samples);
            const int samples_offset = (int)round(s.acq_code_phase_samples);
            s.acc_carrier_phase_rad -= s.carrier_phase_step_rad * (double)samples_offset;
            s.state = 2;
            s.sample_counter += samples_offset;
            s.pos += samples_offset;
        }
    if (s.state < 2 || s.pos + c.vector_length > limit) go = 0;  // standby, or the input block is exhausted
    if (go)
        {
            // do_correlation_step (:886-897): the scalars are narrowed to float exactly there
            const float spc = (float)c.code_samples_per_chip;
            const float rem_carr = s.rem_carr_phase_rad;
            const float pstep = (float)s.carrier_phase_step_rad;
            s_p.sample_offset = s.pos;
            s_p.phase0_re = cosf(rem_carr);
            s_p.phase0_im = -sinf(rem_carr);
            s_p.phase_inc_re = cosf(pstep);
            s_p.phase_inc_im = -sinf(pstep);
            const float prate = HD ? (float)s.carrier_phase_rate_step_rad : 0.0f;
            s_p.phase_rate_re = cosf(prate);
            s_p.phase_rate_im = -sinf(prate);
            s_p.rem_code_phase_chips = (float)s.rem_code_phase_chips * spc;
            s_p.code_phase_step_chips = (float)s.code_phase_step_chips * spc;
            s_p.code_phase_rate_step_chips = HD ? (float)s.code_phase_rate_step_chips * spc : 0.0f;
            s_p.n_samples = (int)c.vector_length;
        }
    else
        {
            // nothing to correlate: an invalid record marks the epoch (records are written in place, field by field: no stack copies)
            unsigned* w = reinterpret_cast<unsigned*>(rec);
            for (unsigned i = 0; i < sizeof(gc_loop_record) / 4; i++) w[i] = 0u;
            rec->state = s.state;
            rec->sample_counter = s.sample_counter;
        }
    return go;
}

// THREADS per channel: 1024 when there are few channels (one workgroup per CU), 256 when there are many
// DATA: pilot tracking, every channel carries the data component's replica (chan.code2)
// HD:   Dll_Pll_Conf::high_dyn -- the high-dynamics resampler and rotator (carrier and code rate terms)
template <int NTAPS, int THREADS, int FMT, bool DATA, bool HD = false>
__global__ __launch_bounds__(THREADS) void trk_closed_loop_kernel(LoopChan* __restrict__ chans,
    gc_loop_record* __restrict__ recs, int n_epochs, int lds_table_floats, const unsigned long long* __restrict__ limits, int resident)
{
    extern __shared__ float lds[];
    __shared__ LoopChan s;
    __shared__ gc_epoch_params s_p;
    __shared__ float2 s_corr[GC_MAX_TAPS];
    __shared__ int s_go;
    const int ch = blockIdx.x;
    const int tid = threadIdx.x;
    {
        // cooperative copy of the channel state into LDS
        const unsigned* src = reinterpret_cast<const unsigned*>(&chans[ch]);
        unsigned* dst = reinterpret_cast<unsigned*>(&s);
        for (unsigned i = tid; i < sizeof(LoopChan) / 4; i += THREADS) dst[i] = src[i];
    }
    __syncthreads();
    if (s.n_taps != NTAPS)
        {
            // a channel that has not been started (or was stopped): standby records, nothing else
            for (int e = 0; e < n_epochs; e++)
                {
                    unsigned* w = reinterpret_cast<unsigned*>(&recs[(size_t)ch * n_epochs + e]);
                    for (unsigned i = tid; i < sizeof(gc_loop_record) / 4; i += THREADS) w[i] = 0u;
                }
            return;
        }
    // samples available to this launch: the channel's buffer length, or (ring input) the stream's head
    const unsigned long long limit = limits ? limits[ch] : s.chan.n_iq;
    // The replica does not change during a launch and this workgroup has the CU's LDS to itself: the doubled image
    // R[i] = code[i mod L] is loaded ONCE and every period addresses its window inside it, instead of re-reading the window from
    // global memory period after period (one round trip and a barrier per period)
    if (resident)
        {
            trk_fill_resident<THREADS>(lds, s.chan.code, DATA ? s.chan.code2 : nullptr, s.chan.code_len);
            __syncthreads();
        }

    for (int e = 0; e < n_epochs; e++)
        {
            gc_loop_record* rec = &recs[(size_t)ch * n_epochs + e];
            if (tid == 0)
                {
                    const int go = loop_prepare<HD>(s, limit, s_p, rec);
                    s_go = go;
                }
            __syncthreads();
            if (!s_go) continue;  // uniform: every later epoch of this launch is skipped the same way

            const float2 r = trk_epoch<NTAPS, HD, HD, FMT, false, false, THREADS, DATA, LOOP_PF, false, LOOP_NT != 0, LOOP_WHOLE != 0>(s.chan, s_p, 0, 1, lds_table_floats, lds, LOOP_ALIGN_PAIRS, resident != 0);
            if (tid < NTAPS + (DATA ? 1 : 0)) s_corr[tid] = r;
            __syncthreads();

            if (tid == 0) loop_after_correlation<NTAPS, DATA>(s, s_corr, rec);
            // the next iteration's barrier orders these writes before any other thread reads s / s_p again
        }
    __syncthreads();
    {
        unsigned* dst = reinterpret_cast<unsigned*>(&chans[ch]);
        const unsigned* src = reinterpret_cast<const unsigned*>(&s);
        for (unsigned i = tid; i < sizeof(LoopChan) / 4; i += THREADS) dst[i] = src[i];
    }
}

#ifdef GNSSCORR_EXPERIMENTS
// -----------------------------------------------------------------------------
// Few channels on a big chip (the sharded receiver: 32 channels per GPU, BASELINE configs[4]): one workgroup per channel leaves
// 7 of 8 CUs idle and a code period costs its full 10 us whatever the load.  Here a channel-period is cut into S slices, one
// workgroup each (the batched kernel's slicing: trk_epoch's `slice` of `n_slices`), ONE launch per code period: a slice leaves
// its partial sums in global memory and draws a ticket; the workgroup that draws the last one adds the partials IN SLICE ORDER
// (deterministic: no float atomics), runs the period's scalar loop maths on one lane exactly as the persistent kernel does
// (loop_after_correlation), prepares the NEXT period's correlator scalars (loop_prepare) and writes the state back.  The kernel
// boundary is the only grid-wide synchronisation: no spinning, no co-residency assumption.  Records agree with the persistent
// kernel's to float rounding (the sums are associated differently); for a given slice count they are reproducible bit for bit.
// MEASURED SLOWER than the persistent kernel and therefore an experiments-build option only (gc_trk_loop_set_geometry refuses
// slices > 1 in the product library): 32 channels x 25 Msps, 13.4 us per code period with 4 or 8 slices (15.3 with 2, 16.4 with
// 16) against 11.4 us for one 1024-thread workgroup per channel -- of the launch's 12 us only ~2 are the slice's correlation; the
// rest is a chain of dependent round trips the persistent kernel does not have (descriptor and scalars from the previous launch,
// code window, first samples, write-through of the partials, ticket, state in, state out) around the same 2.6 us of one-lane maths.
// -----------------------------------------------------------------------------
// Hand-off of the partial sums: every byte is stored with sc1 (agent-scope relaxed atomic stores: write-through), the storing lane
// drains them (s_waitcnt vmcnt(0)) and then adds to the channel's ticket counter; the lane whose add returns S - 1 reads them with
// sc1 loads, which bypass its CU's L1.  That is the fence-free form MI355X_MICROARCH.md lists as measured valid on gfx950 (one
// storing lane per workgroup, the last adder told by the value its add returned); LOOP_SLICE_FENCES=1 adds the agent-scope
// release / acquire pair of acq_final_kernel around it (~3.4 us per period on the critical path).
#ifndef LOOP_SLICE_FENCES
#define LOOP_SLICE_FENCES 0
#endif
struct LoopPrep
{
    gc_epoch_params p;
    int go;
    int pad;
};

// the correlator scalars of the first period of a launch sequence (and its invalid record when there is nothing to correlate)
template <bool HD>
__global__ __launch_bounds__(64) void trk_loop_prepare_kernel(LoopChan* __restrict__ chans, LoopPrep* __restrict__ prep, gc_loop_record* __restrict__ recs, int n_epochs,
    const unsigned long long* __restrict__ limits)
{
    __shared__ LoopChan s;
    __shared__ gc_epoch_params s_p;
    const int ch = blockIdx.x, tid = threadIdx.x;
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(&chans[ch]);
        unsigned* dst = reinterpret_cast<unsigned*>(&s);
        for (unsigned i = tid; i < sizeof(LoopChan) / 4; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    if (s.n_taps == 0) return;  // standby slot: the slice kernel writes its all-zero records
    if (tid == 0)
        {
            const unsigned long long limit = limits ? limits[ch] : s.chan.n_iq;
            const int go = loop_prepare<HD>(s, limit, s_p, &recs[(size_t)ch * n_epochs]);
            prep[ch].p = s_p;
            prep[ch].go = go;
        }
    __syncthreads();
    {
        unsigned* dst = reinterpret_cast<unsigned*>(&chans[ch]);
        const unsigned* src = reinterpret_cast<const unsigned*>(&s);
        for (unsigned i = tid; i < sizeof(LoopChan) / 4; i += 64) dst[i] = src[i];
    }
}

// code period `e` of every channel: blockIdx -> (channel, slice) with all slices of a channel on one XCD (they share the window)
template <int NTAPS, int THREADS, int FMT, bool DATA, bool HD>
__global__ __launch_bounds__(THREADS) void trk_closed_loop_slice_kernel(LoopChan* __restrict__ chans, gc_loop_record* __restrict__ recs, int n_epochs, int e,
    int n_channels, int n_slices, int lds_table_floats, const unsigned long long* __restrict__ limits, LoopPrep* __restrict__ prep, float2* __restrict__ partial,
    unsigned* __restrict__ tickets)
{
    extern __shared__ float lds[];
    __shared__ LoopChan s;  // the finishing workgroup's copy of the channel state (the others use .chan and .n_taps only)
    __shared__ gc_epoch_params s_p;
    __shared__ float2 s_corr[GC_MAX_TAPS];
    __shared__ int s_go, s_last;
    const int tid = threadIdx.x;
    // b = (group * n_slices + slice) * 8 + x, channel = group * 8 + x: equal b % 8 (one XCD under round-robin placement) for a channel's slices
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int slice = q % n_slices, ch = (q / n_slices) * 8 + x;
    if (ch >= n_channels) return;
    constexpr int NOUT = NTAPS + (DATA ? 1 : 0);
    gc_loop_record* rec = &recs[(size_t)ch * n_epochs + e];
    if (tid == 0)
        {
            s.chan = chans[ch].chan;
            s.n_taps = chans[ch].n_taps;
            s_p = prep[ch].p;
            s_go = prep[ch].go;
        }
    __syncthreads();
    if (s.n_taps != NTAPS)
        {
            // a channel that has not been started (or was stopped): standby record, written by its first slice
            if (slice == 0)
                {
                    unsigned* w = reinterpret_cast<unsigned*>(rec);
                    for (unsigned i = tid; i < sizeof(gc_loop_record) / 4; i += THREADS) w[i] = 0u;
                }
            return;
        }
    if (s_go)
        {
            const float2 r = trk_epoch<NTAPS, HD, HD, FMT, false, false, THREADS, DATA, LOOP_PF, false, LOOP_NT != 0, LOOP_WHOLE != 0>(s.chan, s_p, slice, n_slices, lds_table_floats, lds, LOOP_ALIGN_PAIRS);
            if (tid < NOUT) s_corr[tid] = r;
        }
    __syncthreads();
    if (tid == 0)
        {
            // hand the partial sums over; the workgroup that draws the last ticket finishes the period (all in one lane:
            // stores -> release fence -> ticket, ticket -> acquire fence -> loads; the same protocol as acq_final_kernel)
            float* pv = reinterpret_cast<float*>(partial + ((size_t)ch * n_slices + slice) * GC_MAX_TAPS);
            if (s_go)
                for (int t = 0; t < NOUT; t++)
                    {
                        __hip_atomic_store(pv + 2 * t, s_corr[t].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(pv + 2 * t + 1, s_corr[t].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#if LOOP_SLICE_FENCES
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sc1 (write-through) stores above have left for L2 / memory
            const unsigned ticket = __hip_atomic_fetch_add(tickets + ch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (ticket == (unsigned)n_slices - 1u) ? 1 : 0;
#if LOOP_SLICE_FENCES
            if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        }
    __syncthreads();
    if (!s_last) return;
    // ---- the last slice to finish: the rest of general_work for this period ----
    // (the channel state and prep[] were written by the PREVIOUS launch: visible across the kernel boundary to plain loads)
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(&chans[ch]);
        unsigned* dst = reinterpret_cast<unsigned*>(&s);
        for (unsigned i = tid; i < sizeof(LoopChan) / 4; i += THREADS) dst[i] = src[i];
    }
    __syncthreads();
    if (tid == 0)
        {
            const unsigned long long limit = limits ? limits[ch] : s.chan.n_iq;
            if (s_go)
                {
                    float2 taps[GC_MAX_TAPS];
                    for (int t = 0; t < NOUT; t++) taps[t] = make_float2(0.f, 0.f);
                    for (int sl = 0; sl < n_slices; sl++)  // slice order: the same sums whichever workgroup finishes last
                        {
                            const float* qv = reinterpret_cast<const float*>(partial + ((size_t)ch * n_slices + sl) * GC_MAX_TAPS);
                            for (int t = 0; t < NOUT; t++)
                                {
                                    taps[t].x += __hip_atomic_load(qv + 2 * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    taps[t].y += __hip_atomic_load(qv + 2 * t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                        }
                    for (int t = 0; t < NOUT; t++) s_corr[t] = taps[t];
                    loop_after_correlation<NTAPS, DATA>(s, s_corr, rec);
                }
            // the next period's scalars (a period that found nothing to correlate leaves the state where it was: the next one
            // is decided again, as the persistent kernel decides every period)
            if (e + 1 < n_epochs)
                {
                    const int go = loop_prepare<HD>(s, limit, s_p, rec + 1);
                    prep[ch].p = s_p;
                    prep[ch].go = go;
                }
            __hip_atomic_store(tickets + ch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        }
    __syncthreads();
    {
        unsigned* dst = reinterpret_cast<unsigned*>(&chans[ch]);
        const unsigned* src = reinterpret_cast<const unsigned*>(&s);
        for (unsigned i = tid; i < sizeof(LoopChan) / 4; i += THREADS) dst[i] = src[i];
    }
}

#else
struct LoopPrep;
#endif  // GNSSCORR_EXPERIMENTS

__global__ void trk_loop_start_kernel(LoopChan* chans, int ch)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) loop_start(chans[ch]);
}

// -----------------------------------------------------------------------------
// host API
// -----------------------------------------------------------------------------
struct gc_trk_loop
{
    gc_ctx* ctx = nullptr;
    gc_ctx_ref ctx_ref;
    int n_channels = 0, max_code_len = 0, n_taps = 0;
    LoopChan* d_chans = nullptr;
    float* d_codes = nullptr;
    float* d_data_codes = nullptr;            // pilot tracking: data-component replicas, allocated on first use
    std::vector<LoopSync> sync;               // per channel (gc_trk_loop_set_sync); extend_symbols == 0: none installed
    std::vector<int> data_code_len;           // samples of the data replica uploaded for the channel (0: none)
    int pilot = -1;                           // pilot mode of the started channels (-1: none started yet)
    int high_dyn = 0;                         // high-dynamics mode of the started channels
    gc_loop_record* d_recs = nullptr;
    size_t recs_cap = 0;
    int forced_threads = 0;  // gc_trk_loop_set_geometry: threads per workgroup of the persistent kernel (0: by channel count)
    int forced_slices = 0;   // gc_trk_loop_set_geometry: workgroups per channel-period (0: by channel count; 1: persistent kernel)
    LoopPrep* d_prep = nullptr;      // sliced launches: next period's correlator scalars per channel
    float2* d_slice_partial = nullptr;
    unsigned* d_tickets = nullptr;
    int slice_cap = 0;               // slices the partial buffer was allocated for
    int iq_format = GC_IQ_F32;  // sample format of every channel's input (gc_trk_loop_set_input_format)
    std::vector<char> started;
    std::vector<const void*> iq;
    std::vector<unsigned long long> n_iq;
    // ring input (gc_trk_loop_set_input_stream)
    std::vector<gc_stream*> streams;          // per channel, or NULL
    std::vector<unsigned long long> pos_host; // last known stream position of the channel (from the records)
    std::vector<char> pos_known;
    std::vector<char> idle;                   // the channel's last record said state 0 (standby after loss of lock)
    // Per-launch ring limits travel through a small ring of slots (pinned host copy + device copy + an event recorded behind the
    // launch that reads the device copy): an asynchronous launch keeps its slot until it has finished, so back-to-back
    // push -> run_dev -> push -> run_dev never rewrites limits an earlier, still queued launch will read.
    static constexpr int LIMIT_SLOTS = 8;
    unsigned long long* d_limits[LIMIT_SLOTS] = {};
    unsigned long long* h_limits[LIMIT_SLOTS] = {};   // pinned
    hipEvent_t limit_done[LIMIT_SLOTS] = {};
    bool limit_used[LIMIT_SLOTS] = {};
    int limit_next = 0;
    // the last launch of the engine, on whatever stream the caller chose: entry points that rewrite device state wait for it
    hipEvent_t last_launch = nullptr;
    bool launched = false;
};

// waits until no launch of the engine is in flight on any stream (the caller holds the context mutex)
static hipError_t loop_quiesce(gc_trk_loop* l)
{
    hipError_t e = hipStreamSynchronize(l->ctx->stream);
    if (e == hipSuccess && l->launched) e = hipEventSynchronize(l->last_launch);
    return e;
}

static void loop_free_limits(gc_trk_loop* l)
{
    for (int k = 0; k < gc_trk_loop::LIMIT_SLOTS; k++)
        {
            (void)hipFree(l->d_limits[k]);
            if (l->h_limits[k]) (void)hipHostFree(l->h_limits[k]);
            if (l->limit_done[k]) (void)hipEventDestroy(l->limit_done[k]);
        }
    if (l->last_launch) (void)hipEventDestroy(l->last_launch);
}

extern "C" {

gc_status gc_trk_loop_create(gc_ctx* ctx, int n_channels, int max_code_length, gc_trk_loop** out)
{
    GC_REQUIRE(ctx && out, "gc_trk_loop_create: NULL argument");
    *out = nullptr;
    GC_REQUIRE(n_channels > 0, "gc_trk_loop_create: n_channels must be > 0");
    GC_REQUIRE(max_code_length > 0 && max_code_length + 64 <= 16000, "gc_trk_loop_create: max_code_length must be in 1..%d", 16000 - 64);
    gc_device_guard g(ctx->device);
    gc_trk_loop* l = new gc_trk_loop();
    l->ctx = ctx;
    l->ctx_ref.bind(ctx);
    l->n_channels = n_channels;
    l->max_code_len = max_code_length;
    hipError_t e1 = hipMalloc(&l->d_chans, sizeof(LoopChan) * n_channels);
    hipError_t e2 = hipMalloc(&l->d_codes, sizeof(float) * (size_t)n_channels * max_code_length);
    if (e1 != hipSuccess || e2 != hipSuccess)
        {
            (void)hipFree(l->d_chans);
            (void)hipFree(l->d_codes);
            delete l;
            return gc_fail(GC_ERR_HIP, "gc_trk_loop_create: hipMalloc failed");
        }
    (void)hipMemset(l->d_chans, 0, sizeof(LoopChan) * n_channels);
    l->started.assign(n_channels, 0);
    l->sync.assign(n_channels, LoopSync());
    l->data_code_len.assign(n_channels, 0);
    l->iq.assign(n_channels, nullptr);
    l->n_iq.assign(n_channels, 0);
    l->streams.assign(n_channels, nullptr);
    l->pos_host.assign(n_channels, 0);
    l->pos_known.assign(n_channels, 0);
    l->idle.assign(n_channels, 0);
    bool ok = hipEventCreateWithFlags(&l->last_launch, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < gc_trk_loop::LIMIT_SLOTS && ok; k++)
        ok = hipMalloc(&l->d_limits[k], sizeof(unsigned long long) * n_channels) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void**>(&l->h_limits[k]), sizeof(unsigned long long) * n_channels, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&l->limit_done[k], hipEventDisableTiming) == hipSuccess;
    if (!ok)
        {
            loop_free_limits(l);
            (void)hipFree(l->d_chans);
            (void)hipFree(l->d_codes);
            delete l;
            return gc_fail(GC_ERR_HIP, "gc_trk_loop_create: allocation failed");
        }
    *out = l;
    return GC_OK;
}

gc_status gc_trk_loop_destroy(gc_trk_loop* l)
{
    if (!l) return GC_OK;
    gc_device_guard g(l->ctx->device);
    (void)loop_quiesce(l);
    (void)hipFree(l->d_chans);
    (void)hipFree(l->d_codes);
    (void)hipFree(l->d_data_codes);
    (void)hipFree(l->d_recs);
    (void)hipFree(l->d_prep);
    (void)hipFree(l->d_slice_partial);
    (void)hipFree(l->d_tickets);
    loop_free_limits(l);
    for (gc_stream* r : l->streams)
        if (r) gc_stream_drop(r);
    delete l;
    return GC_OK;
}

gc_status gc_trk_loop_set_geometry(gc_trk_loop* l, int threads_per_workgroup, int slices_per_channel)
{
    GC_REQUIRE(l, "gc_trk_loop_set_geometry: NULL handle");
    GC_REQUIRE(threads_per_workgroup == 0 || threads_per_workgroup == 256 || threads_per_workgroup == 512 || threads_per_workgroup == 1024,
        "gc_trk_loop_set_geometry: threads_per_workgroup must be 0 (automatic), 256, 512 or 1024");
    GC_REQUIRE(slices_per_channel >= 0 && slices_per_channel <= 16, "gc_trk_loop_set_geometry: slices_per_channel must be in 0..16 (0: automatic)");
#ifndef GNSSCORR_EXPERIMENTS
    GC_REQUIRE(slices_per_channel <= 1, "gc_trk_loop_set_geometry: sliced code periods measured slower than one workgroup per channel and exist in "
                                        "experiments builds only (make exp)");
#endif
    std::lock_guard<std::mutex> lk(l->ctx->mtx);
    l->forced_threads = threads_per_workgroup;
    l->forced_slices = slices_per_channel;
    return GC_OK;
}

gc_status gc_trk_loop_set_input_dev(gc_trk_loop* l, int ch, const void* dev_iq, uint64_t n_samples)
{
    GC_REQUIRE(l && dev_iq, "gc_trk_loop_set_input_dev: NULL argument");
    GC_REQUIRE(ch >= 0 && ch < l->n_channels, "gc_trk_loop_set_input_dev: channel %d out of range", ch);
    const uintptr_t es = l->iq_format == GC_IQ_F32 ? 8 : l->iq_format == GC_IQ_I16 ? 4 : 2;
    GC_REQUIRE((reinterpret_cast<uintptr_t>(dev_iq) % es) == 0, "gc_trk_loop_set_input_dev: IQ pointer must be aligned to one sample (%d bytes)", (int)es);
    gc_device_guard g(l->ctx->device);
    std::lock_guard<std::mutex> lk(l->ctx->mtx);
    if (l->streams[ch]) gc_stream_drop(l->streams[ch]);
    l->streams[ch] = nullptr;
    l->iq[ch] = dev_iq;
    l->n_iq[ch] = n_samples;
    if (l->started[ch])
        {
            // a running channel keeps its state; only the input block changes (stream position restarts at 0)
            GC_HIP(loop_quiesce(l));
            LoopChan h;
            GC_HIP(hipMemcpy(&h, l->d_chans + ch, sizeof h, hipMemcpyDeviceToHost));
            h.chan.iq = dev_iq;
            h.chan.n_iq = n_samples;
            h.chan.ring_len = 0;
            h.pos = 0;
            GC_HIP(hipMemcpy(l->d_chans + ch, &h, sizeof h, hipMemcpyHostToDevice));
        }
    return GC_OK;
}

gc_status gc_trk_loop_set_input_format(gc_trk_loop* l, int iq_format)
{
    GC_REQUIRE(l, "gc_trk_loop_set_input_format: NULL handle");
    GC_REQUIRE(iq_format == GC_IQ_F32 || iq_format == GC_IQ_I16 || iq_format == GC_IQ_I8, "gc_trk_loop_set_input_format: unknown format %d", iq_format);
    if (iq_format == l->iq_format) return GC_OK;
    for (int i = 0; i < l->n_channels; i++)
        if (l->started[i] || l->iq[i] != nullptr)
            return gc_fail(GC_ERR_STATE, "gc_trk_loop_set_input_format: set the format before binding inputs or starting channels");
    l->iq_format = iq_format;
    return GC_OK;
}

gc_status gc_trk_loop_set_input_stream(gc_trk_loop* l, int ch, gc_stream* s)
{
    GC_REQUIRE(l && s, "gc_trk_loop_set_input_stream: NULL argument");
    GC_REQUIRE(ch >= 0 && ch < l->n_channels, "gc_trk_loop_set_input_stream: channel %d out of range", ch);
    GC_REQUIRE(s->ctx->device == l->ctx->device, "gc_trk_loop_set_input_stream: the stream lives on another GPU");
    GC_REQUIRE(s->iq_format == l->iq_format, "gc_trk_loop_set_input_stream: stream format %d, engine format %d (gc_trk_loop_set_input_format)",
        s->iq_format, l->iq_format);
    if (l->started[ch]) return gc_fail(GC_ERR_STATE, "gc_trk_loop_set_input_stream: channel %d is running; bind the stream before gc_trk_loop_start", ch);
    std::lock_guard<std::mutex> lk(l->ctx->mtx);
    gc_stream_keep(s);
    if (l->streams[ch]) gc_stream_drop(l->streams[ch]);
    l->streams[ch] = s;
    l->iq[ch] = s->d_ring;
    l->n_iq[ch] = ~0ull;
    return GC_OK;
}

gc_status gc_trk_loop_set_sync(gc_trk_loop* l, int ch, const gc_loop_sync_conf* sync, const float* data_code, int data_code_length)
{
    GC_REQUIRE(l, "gc_trk_loop_set_sync: NULL handle");
    GC_REQUIRE(ch >= 0 && ch < l->n_channels, "gc_trk_loop_set_sync: channel %d out of range", ch);
    if (!sync)
        {
            std::lock_guard<std::mutex> lk(l->ctx->mtx);
            l->sync[ch] = LoopSync();
            l->data_code_len[ch] = 0;
            return GC_OK;
        }
    GC_REQUIRE(sync->extend_correlation_symbols >= 1, "gc_trk_loop_set_sync: extend_correlation_symbols must be >= 1");
    GC_REQUIRE(sync->secondary_code_length >= 0 && sync->secondary_code_length <= 128, "gc_trk_loop_set_sync: secondary_code_length must be in 0..128");
    GC_REQUIRE(sync->preamble_length_symbols >= 0 && sync->preamble_length_symbols <= 192, "gc_trk_loop_set_sync: preamble_length_symbols must be in 0..192");
    GC_REQUIRE(sync->symbols_per_bit >= 0, "gc_trk_loop_set_sync: symbols_per_bit must be >= 0");
    LoopSync y;
    std::memset(&y, 0, sizeof y);
    y.extend_symbols = sync->extend_correlation_symbols;
    y.track_pilot = sync->track_pilot ? 1 : 0;
    y.symbols_per_bit = sync->symbols_per_bit;
    y.secondary_len = sync->secondary_code_length;
    y.preamble_len = sync->preamble_length_symbols;
    y.bit_sync_min_time_s = sync->bit_sync_min_time_s;
    y.pll_bw_narrow_hz = sync->pll_bw_narrow_hz;
    y.dll_bw_narrow_hz = sync->dll_bw_narrow_hz;
    y.el_narrow_chips = sync->early_late_space_narrow_chips;
    y.vel_narrow_chips = sync->very_early_late_space_narrow_chips;
    for (int i = 0; i < y.secondary_len; i++)
        {
            const char ch_i = sync->secondary_code[i];
            GC_REQUIRE(ch_i == '0' || ch_i == '1', "gc_trk_loop_set_sync: secondary_code[%d] is not '0' or '1'", i);
            const int bit = y.secondary_len - 1 - i;  // newest symbol in bit 0
            if (ch_i == '1') y.sec_ones[bit >> 5] |= 1u << (bit & 31);
        }
    for (int i = 0; i < y.preamble_len; i++)
        {
            const int v = sync->preamble_symbols[i];
            GC_REQUIRE(v == 1 || v == -1, "gc_trk_loop_set_sync: preamble_symbols[%d] is not +1 / -1", i);
            const int bit = y.preamble_len - 1 - i;
            if (v == 1) y.pre_plus[bit >> 5] |= 1u << (bit & 31);
        }
    gc_device_guard g(l->ctx->device);
    std::lock_guard<std::mutex> lk(l->ctx->mtx);
    if (y.track_pilot)
        {
            GC_REQUIRE(data_code, "gc_trk_loop_set_sync: track_pilot needs the data component's replica");
            GC_REQUIRE(data_code_length > 0 && data_code_length <= l->max_code_len, "gc_trk_loop_set_sync: data_code_length %d not in 1..%d", data_code_length,
                l->max_code_len);
            if (!l->d_data_codes) GC_HIP(hipMalloc(&l->d_data_codes, sizeof(float) * (size_t)l->n_channels * l->max_code_len));
            GC_HIP(loop_quiesce(l));
            GC_HIP(hipMemcpy(l->d_data_codes + (size_t)ch * l->max_code_len, data_code, sizeof(float) * data_code_length, hipMemcpyHostToDevice));
            l->data_code_len[ch] = data_code_length;  // must equal the tracking replica's length: checked at start
        }
    l->sync[ch] = y;
    return GC_OK;
}

gc_status gc_trk_loop_start(gc_trk_loop* l, int ch, const gc_loop_conf* conf, const float* code, int code_length)
{
    GC_REQUIRE(l && conf && code, "gc_trk_loop_start: NULL argument");
    GC_REQUIRE(ch >= 0 && ch < l->n_channels, "gc_trk_loop_start: channel %d out of range", ch);
    GC_REQUIRE(code_length > 0 && code_length <= l->max_code_len, "gc_trk_loop_start: code_length %d not in 1..%d", code_length, l->max_code_len);
    GC_REQUIRE(l->iq[ch] != nullptr, "gc_trk_loop_start: channel %d has no input (gc_trk_loop_set_input_dev)", ch);
    GC_REQUIRE(conf->cn0_samples >= 1 && conf->cn0_samples <= LOOP_MAX_CN0, "gc_trk_loop_start: cn0_samples must be in 1..%d", LOOP_MAX_CN0);
    GC_REQUIRE(conf->vector_length > 0 && conf->fs_in > 0 && conf->code_chip_rate_hz > 0, "gc_trk_loop_start: bad signal description");
    GC_REQUIRE((uint32_t)code_length == conf->code_length_chips * conf->code_samples_per_chip,
        "gc_trk_loop_start: code_length %d != code_length_chips * code_samples_per_chip", code_length);
    GC_REQUIRE(conf->high_dyn_smoother_length <= LOOP_MAX_SMOOTHER, "gc_trk_loop_start: high_dyn_smoother_length must be <= %d", LOOP_MAX_SMOOTHER);
    const int n_taps = conf->veml ? 5 : 3;
    gc_device_guard g(l->ctx->device);
    std::lock_guard<std::mutex> lk(l->ctx->mtx);
    {
        // the engine's tap count is that of its running channels: it is free again once every channel has been stopped
        bool any_started = false;
        for (int i = 0; i < l->n_channels; i++) any_started |= (i != ch && l->started[i]);
        GC_REQUIRE(!any_started || l->n_taps == n_taps, "gc_trk_loop_start: all channels of one loop engine use the same tap count (%d)", l->n_taps);
    }
    LoopSync y = l->sync[ch];
    if (y.extend_symbols == 0)
        {
            // nothing installed: a signal whose bit synchronisation never happens (stays in state 2)
            y.extend_symbols = 1;
            y.symbols_per_bit = 2;
        }
    if (y.track_pilot)
        GC_REQUIRE(l->data_code_len[ch] == code_length, "gc_trk_loop_start: the data replica has %d samples, the tracking replica %d", l->data_code_len[ch],
            code_length);
    {
        bool others_started = false;
        for (int i = 0; i < l->n_channels; i++) others_started |= (i != ch && l->started[i]);
        if (!others_started) l->pilot = -1;
    }
    const int hd = conf->high_dyn_smoother_length > 0 ? 1 : 0;
    if (l->pilot < 0)
        {
            l->pilot = y.track_pilot;
            l->high_dyn = hd;
        }
    GC_REQUIRE(l->pilot == y.track_pilot, "gc_trk_loop_start: all channels of one loop engine share the pilot mode (track_pilot = %d)", l->pilot);
    GC_REQUIRE(l->high_dyn == hd, "gc_trk_loop_start: all channels of one loop engine share the high_dyn mode (%d)", l->high_dyn);

    hipStream_t st = l->ctx->stream;
    GC_HIP(loop_quiesce(l));
    l->n_taps = n_taps;  // every check has passed
    GC_HIP(hipMemcpy(l->d_codes + (size_t)ch * l->max_code_len, code, sizeof(float) * code_length, hipMemcpyHostToDevice));
    LoopChan h;
    std::memset(&h, 0, sizeof h);
    h.conf = *conf;
    h.sync = y;
    h.chan.code2 = y.track_pilot ? l->d_data_codes + (size_t)ch * l->max_code_len : nullptr;
    h.chan.iq = l->iq[ch];
    h.chan.n_iq = l->n_iq[ch];
    h.chan.code = l->d_codes + (size_t)ch * l->max_code_len;
    h.chan.code_len = code_length;
    if (gc_stream* r = l->streams[ch])
        {
            GC_REQUIRE(conf->vector_length <= r->mirror, "gc_trk_loop_start: vector_length %u exceeds the stream's max_window %llu", conf->vector_length,
                (unsigned long long)r->mirror);
            h.chan.ring_len = (unsigned)r->capacity;
        }
    h.n_taps = n_taps;
    // tap shifts in code samples (dll_pll_veml_tracking.cc:372-390, :720-732)
    const float spc = (float)conf->code_samples_per_chip;
    if (conf->veml)
        {
            h.chan.shifts[0] = -conf->very_early_late_space_chips * spc;
            h.chan.shifts[1] = -conf->early_late_space_chips * spc;
            h.chan.shifts[2] = 0.0f;
            h.chan.shifts[3] = conf->early_late_space_chips * spc;
            h.chan.shifts[4] = conf->very_early_late_space_chips * spc;
        }
    else
        {
            h.chan.shifts[0] = -conf->early_late_space_chips * spc;
            h.chan.shifts[1] = 0.0f;
            h.chan.shifts[2] = conf->early_late_space_chips * spc;
        }
    // ring input is addressed with absolute stream sample numbers: the channel starts where its counter says
    h.pos = l->streams[ch] ? conf->sample_counter : 0;
    l->pos_host[ch] = h.pos;
    l->pos_known[ch] = 1;
    l->idle[ch] = 0;
    GC_HIP(hipMemcpy(l->d_chans + ch, &h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(trk_loop_start_kernel, dim3(1), dim3(64), 0, st, l->d_chans, ch);
    GC_HIP(hipGetLastError());
    GC_HIP(hipStreamSynchronize(st));
    l->started[ch] = 1;
    return GC_OK;
}

}  // extern "C"

extern "C" gc_status gc_trk_loop_stop(gc_trk_loop* l, int ch)
{
    GC_REQUIRE(l, "gc_trk_loop_stop: NULL handle");
    GC_REQUIRE(ch >= 0 && ch < l->n_channels, "gc_trk_loop_stop: channel %d out of range", ch);
    gc_device_guard g(l->ctx->device);
    std::lock_guard<std::mutex> lk(l->ctx->mtx);
    if (!l->started[ch]) return GC_OK;
    GC_HIP(loop_quiesce(l));
    // n_taps = 0 marks the slot as standby for the kernel; the rest of the state is rewritten by the next start
    int zero = 0;
    GC_HIP(hipMemcpy(reinterpret_cast<char*>(l->d_chans + ch) + offsetof(LoopChan, n_taps), &zero, sizeof zero, hipMemcpyHostToDevice));
    l->started[ch] = 0;
    return GC_OK;
}

#ifdef GNSSCORR_EXPERIMENTS
// one launch per code period: `n_epochs` launches of n_channels x n_slices workgroups behind one prepare launch
template <int NT, int FM, bool DA, bool HD>
static hipError_t loop_launch_slices_t(gc_trk_loop* l, int n_epochs, gc_loop_record* dev_records, hipStream_t st, int n_slices, int lds_table_floats,
    const unsigned long long* limits)
{
    constexpr int TH = 256;
    const size_t lds_bytes = (size_t)(trk_hdr_floats(TH) + lds_table_floats) * sizeof(float);
    if (lds_bytes > 48 * 1024)
        {
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&trk_closed_loop_slice_kernel<NT, TH, FM, DA, HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (ea != hipSuccess) return ea;
        }
    hipLaunchKernelGGL((trk_loop_prepare_kernel<HD>), dim3(l->n_channels), dim3(64), 0, st, l->d_chans, l->d_prep, dev_records, n_epochs, limits);
    const unsigned grid = (unsigned)(((l->n_channels + 7) / 8) * n_slices * 8);
    for (int e = 0; e < n_epochs; e++)
        hipLaunchKernelGGL((trk_closed_loop_slice_kernel<NT, TH, FM, DA, HD>), dim3(grid), dim3(TH), lds_bytes, st, l->d_chans, dev_records, n_epochs, e, l->n_channels, n_slices,
            lds_table_floats, limits, l->d_prep, l->d_slice_partial, l->d_tickets);
    return hipGetLastError();
}
template <int NT, int FM>
static hipError_t loop_launch_slices_f(gc_trk_loop* l, bool pilot, bool hd, int n_epochs, gc_loop_record* dev_records, hipStream_t st, int n_slices, int lds_table_floats,
    const unsigned long long* limits)
{
    if (pilot) return hd ? loop_launch_slices_t<NT, FM, true, true>(l, n_epochs, dev_records, st, n_slices, lds_table_floats, limits)
                         : loop_launch_slices_t<NT, FM, true, false>(l, n_epochs, dev_records, st, n_slices, lds_table_floats, limits);
    return hd ? loop_launch_slices_t<NT, FM, false, true>(l, n_epochs, dev_records, st, n_slices, lds_table_floats, limits)
              : loop_launch_slices_t<NT, FM, false, false>(l, n_epochs, dev_records, st, n_slices, lds_table_floats, limits);
}
static hipError_t loop_launch_slices(gc_trk_loop* l, bool pilot, int n_epochs, gc_loop_record* dev_records, hipStream_t st, int n_slices, int lds_table_floats,
    const unsigned long long* limits)
{
    const bool hd = l->high_dyn != 0;
    if (l->n_taps == 5)
        {
            if (l->iq_format == GC_IQ_I16) return loop_launch_slices_f<5, GC_IQ_I16>(l, pilot, hd, n_epochs, dev_records, st, n_slices, lds_table_floats, limits);
            if (l->iq_format == GC_IQ_I8) return loop_launch_slices_f<5, GC_IQ_I8>(l, pilot, hd, n_epochs, dev_records, st, n_slices, lds_table_floats, limits);
            return loop_launch_slices_f<5, GC_IQ_F32>(l, pilot, hd, n_epochs, dev_records, st, n_slices, lds_table_floats, limits);
        }
    if (l->iq_format == GC_IQ_I16) return loop_launch_slices_f<3, GC_IQ_I16>(l, pilot, hd, n_epochs, dev_records, st, n_slices, lds_table_floats, limits);
    if (l->iq_format == GC_IQ_I8) return loop_launch_slices_f<3, GC_IQ_I8>(l, pilot, hd, n_epochs, dev_records, st, n_slices, lds_table_floats, limits);
    return loop_launch_slices_f<3, GC_IQ_F32>(l, pilot, hd, n_epochs, dev_records, st, n_slices, lds_table_floats, limits);
}

#endif  // GNSSCORR_EXPERIMENTS

extern "C" {

static gc_status loop_launch(gc_trk_loop* l, int n_epochs, gc_loop_record* dev_records, hipStream_t st, bool positions_known)
{
    {
        // channels that were never started (or were stopped) sit in standby: all-zero records, state 0
        bool any = false;
        for (int i = 0; i < l->n_channels; i++) any |= l->started[i] != 0;
        if (!any) return gc_fail(GC_ERR_STATE, "gc_trk_loop_run: no channel has been started (gc_trk_loop_start)");
    }
    // ring inputs: this launch may use what has been pushed so far, and must not be overtaken by later pushes.  Reader slots
    // are reserved (floor = the oldest sample a channel with a known position still needs, else the oldest resident one) before
    // the residency check and the enqueue; the limits are the heads seen by those reservations.
    std::vector<gc_stream*> rings;
    std::vector<uint64_t> floors;
    std::vector<char> floor_unknown;
    std::vector<gc_stream_ticket> tickets;
    bool any_ring = false;
    // this launch's slot of the limits ring: free once the launch that used it last has finished
    const int slot = l->limit_next;
    if (l->limit_used[slot]) GC_HIP(hipEventSynchronize(l->limit_done[slot]));
    unsigned long long* h_limits = l->h_limits[slot];
    for (int i = 0; i < l->n_channels; i++)
        {
            gc_stream* r = l->streams[i];
            h_limits[i] = l->n_iq[i];
            if (!r) continue;
            any_ring = true;
            // standby (never started, stopped, or lost lock): reads nothing, holds nothing back in the ring
            if (!l->started[i] || l->idle[i]) continue;
            const bool known = positions_known && l->pos_known[i];
            size_t k = std::find(rings.begin(), rings.end(), r) - rings.begin();
            if (k == rings.size())
                {
                    rings.push_back(r);
                    floors.push_back(~0ull);
                    floor_unknown.push_back(0);
                }
            // one channel with an unknown position pins the floor at the oldest resident sample
            if (!known) floor_unknown[k] = 1;
            floors[k] = std::min<uint64_t>(floors[k], known ? l->pos_host[i] : ~0ull);
        }
    auto cancel_all = [&]() {
        for (size_t k = 0; k < tickets.size(); k++) gc_stream_cancel_read(rings[k], tickets[k]);
    };
    // Between the reservations below and their commit behind the launch NOTHING may return without releasing the reader slots: a slot
    // left `pending` blocks every later push that would evict below its floor, and gc_stream_drop's drain, for ever.
#define LOOP_HIP_OR_CANCEL(call)                                                                                              \
    do                                                                                                                        \
        {                                                                                                                     \
            hipError_t e_ = (call);                                                                                           \
            if (e_ != hipSuccess)                                                                                             \
                {                                                                                                             \
                    cancel_all();                                                                                             \
                    return gc_fail(GC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
                }                                                                                                             \
        }                                                                                                                     \
    while (0)
    tickets.resize(rings.size());
    for (size_t k = 0; k < rings.size(); k++)
        {
            const bool pinned_oldest = floor_unknown[k] || floors[k] == ~0ull;
            gc_status rs = gc_stream_begin_read(rings[k], st, pinned_oldest ? GC_STREAM_FLOOR_OLDEST : floors[k], &tickets[k]);
            if (rs != GC_OK)
                {
                    cancel_all();
                    if (!pinned_oldest)
                        return gc_fail(GC_ERR_STATE, "gc_trk_loop_run: a channel (at sample %llu) fell behind the ring", (unsigned long long)floors[k]);
                    return rs;
                }
        }
    for (int i = 0; i < l->n_channels; i++)
        {
            gc_stream* r = l->streams[i];
            if (!r) continue;
            if (!l->started[i] || l->idle[i])
                {
                    h_limits[i] = 0;
                    continue;
                }
            const size_t k = std::find(rings.begin(), rings.end(), r) - rings.begin();
            h_limits[i] = tickets[k].head;
        }
    if (any_ring)
        {
            hipError_t ce = hipMemcpyAsync(l->d_limits[slot], h_limits, sizeof(unsigned long long) * l->n_channels, hipMemcpyHostToDevice, st);
            if (ce != hipSuccess)
                {
                    cancel_all();
                    return gc_fail(GC_ERR_HIP, "gc_trk_loop_run: %s", hipGetErrorString(ce));
                }
        }
    const unsigned long long* limits = any_ring ? l->d_limits[slot] : nullptr;
    const bool pilot = l->pilot > 0;
    // LDS code image: the doubled resident one (2 L + 64 floats per replica, loaded once per launch) when it fits beside the
    // header, else the per-period window (L + 64)
    // header, else the per-period window (L + 64).  With more channels than CUs several workgroups share a CU's 160 KB: the resident
    // image is then taken only while two workgroups still fit (<= 64 KiB each; Galileo E1 with the pilot's data component is 131 KB:
    // one workgroup per CU, which only costs nothing while every channel has a CU to itself)
    const int n_cus = l->ctx->n_cus > 0 ? l->ctx->n_cus : 256;
    int lds_table_floats = (2 * l->max_code_len + 64) * (pilot ? 2 : 1);
    int resident = 1;
    const size_t resident_bytes = (size_t)(trk_hdr_floats(1024) + lds_table_floats) * sizeof(float);
    if (resident_bytes > 150 * 1024 || (l->n_channels > n_cus && resident_bytes > 64 * 1024))
        {
            lds_table_floats = (l->max_code_len + 64) * (pilot ? 2 : 1);
            resident = 0;
        }
    // few channels: more threads each, so that a channel's epoch is spread over a whole CU (measured, 256 channels x 64
    // epochs on 256 CUs: 0.89 / 0.65 / 0.69 ms with 256 / 512 / 1024 threads)
    const int threads = l->forced_threads ? l->forced_threads : (2 * l->n_channels <= n_cus ? 1024 : l->n_channels <= 2 * n_cus ? 512 : 256);
    const size_t lds_bytes = (size_t)(trk_hdr_floats(threads) + lds_table_floats) * sizeof(float);
    // workgroups per channel-period: 1 = the persistent one-workgroup-per-channel kernel (always, in the product library); an
    // experiments build cuts the period into slices on request (gc_trk_loop_set_geometry), one launch per period: measured slower
    int n_slices = 1;
#ifdef GNSSCORR_EXPERIMENTS
    n_slices = std::max(1, l->forced_slices);
    if (n_slices > 1)
        {
            if (!l->d_prep || n_slices > l->slice_cap)
                {
                    (void)hipFree(l->d_slice_partial);
                    l->d_slice_partial = nullptr;
                    if (!l->d_prep) LOOP_HIP_OR_CANCEL(hipMalloc(&l->d_prep, sizeof(LoopPrep) * l->n_channels));
                    if (!l->d_tickets)
                        {
                            LOOP_HIP_OR_CANCEL(hipMalloc(&l->d_tickets, sizeof(unsigned) * l->n_channels));
                            LOOP_HIP_OR_CANCEL(hipMemsetAsync(l->d_tickets, 0, sizeof(unsigned) * l->n_channels, st));
                        }
                    LOOP_HIP_OR_CANCEL(hipMalloc(&l->d_slice_partial, sizeof(float2) * GC_MAX_TAPS * (size_t)l->n_channels * n_slices));
                    l->slice_cap = n_slices;
                }
            const hipError_t se = loop_launch_slices(l, pilot, n_epochs, dev_records, st, n_slices, (l->max_code_len + 64) * (pilot ? 2 : 1), limits);
            if (se != hipSuccess)
                {
                    cancel_all();
                    return gc_fail(GC_ERR_HIP, "gc_trk_loop_run: kernel launch failed: %s", hipGetErrorString(se));
                }
        }
#endif
    if (n_slices <= 1)
    {
#define LAUNCH_LOOP_D(NT, TH, FM, DA)                                                                                                             \
    do                                                                                                                                        \
        {                                                                                                                                     \
            if (lds_bytes > 48 * 1024)                                                                                                        \
                LOOP_HIP_OR_CANCEL(hipFuncSetAttribute(reinterpret_cast<const void*>(&trk_closed_loop_kernel<NT, TH, FM, DA>),                            \
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                                                             \
            hipLaunchKernelGGL((trk_closed_loop_kernel<NT, TH, FM, DA>), dim3(l->n_channels), dim3(TH), lds_bytes, st, l->d_chans, dev_records, \
                n_epochs, lds_table_floats, limits, resident);                                                                                          \
        }                                                                                                                                     \
    while (0)
#define LAUNCH_LOOP(NT, TH, FM)                      \
    do                                               \
        {                                            \
            if (pilot) LAUNCH_LOOP_D(NT, TH, FM, true); \
            else LAUNCH_LOOP_D(NT, TH, FM, false);   \
        }                                            \
    while (0)
#define LAUNCH_LOOP_FMT(NT, TH)                                           \
    do                                                                    \
        {                                                                 \
            if (l->iq_format == GC_IQ_I16) LAUNCH_LOOP(NT, TH, GC_IQ_I16); \
            else if (l->iq_format == GC_IQ_I8) LAUNCH_LOOP(NT, TH, GC_IQ_I8); \
            else LAUNCH_LOOP(NT, TH, GC_IQ_F32);                          \
        }                                                                 \
    while (0)
    if (l->high_dyn)
        {
            // high-dynamics kernels: 256 threads per channel (the per-sample exact rotator keeps a workgroup busy)
#define LAUNCH_LOOP_HD_D(NT, FM, DA)                                                                                                              \
    do                                                                                                                                        \
        {                                                                                                                                     \
            if (lds_bytes_hd > 48 * 1024)                                                                                                     \
                LOOP_HIP_OR_CANCEL(hipFuncSetAttribute(reinterpret_cast<const void*>(&trk_closed_loop_kernel<NT, 256, FM, DA, true>),                     \
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_hd));                                                          \
            hipLaunchKernelGGL((trk_closed_loop_kernel<NT, 256, FM, DA, true>), dim3(l->n_channels), dim3(256), lds_bytes_hd, st, l->d_chans,  \
                dev_records, n_epochs, lds_table_floats, limits, resident);                                                                             \
        }                                                                                                                                     \
    while (0)
#define LAUNCH_LOOP_HD(NT, FM)                     \
    do                                             \
        {                                          \
            if (pilot) LAUNCH_LOOP_HD_D(NT, FM, true); \
            else LAUNCH_LOOP_HD_D(NT, FM, false);  \
        }                                          \
    while (0)
            const size_t lds_bytes_hd = (size_t)(trk_hdr_floats(256) + lds_table_floats) * sizeof(float);
            if (l->n_taps == 5)
                {
                    if (l->iq_format == GC_IQ_I16) LAUNCH_LOOP_HD(5, GC_IQ_I16);
                    else if (l->iq_format == GC_IQ_I8) LAUNCH_LOOP_HD(5, GC_IQ_I8);
                    else LAUNCH_LOOP_HD(5, GC_IQ_F32);
                }
            else
                {
                    if (l->iq_format == GC_IQ_I16) LAUNCH_LOOP_HD(3, GC_IQ_I16);
                    else if (l->iq_format == GC_IQ_I8) LAUNCH_LOOP_HD(3, GC_IQ_I8);
                    else LAUNCH_LOOP_HD(3, GC_IQ_F32);
                }
#undef LAUNCH_LOOP_HD
#undef LAUNCH_LOOP_HD_D
        }
    else if (l->n_taps == 5)
        {
            if (threads == 1024) LAUNCH_LOOP_FMT(5, 1024);
            else if (threads == 512) LAUNCH_LOOP_FMT(5, 512);
            else LAUNCH_LOOP_FMT(5, 256);
        }
    else
        {
            if (threads == 1024) LAUNCH_LOOP_FMT(3, 1024);
            else if (threads == 512) LAUNCH_LOOP_FMT(3, 512);
            else LAUNCH_LOOP_FMT(3, 256);
        }
#undef LAUNCH_LOOP_FMT
#undef LAUNCH_LOOP
#undef LAUNCH_LOOP_D
    }
    {
        hipError_t le = hipGetLastError();
        if (le != hipSuccess)
            {
                cancel_all();
                return gc_fail(GC_ERR_HIP, "gc_trk_loop_run: kernel launch failed: %s", hipGetErrorString(le));
            }
    }
#undef LOOP_HIP_OR_CANCEL
    // the kernel is enqueued: the tickets are committed whatever the event records below return (the error is reported afterwards)
    gc_status out = GC_OK;
    hipError_t ee = hipEventRecord(l->last_launch, st);
    l->launched = true;
    if (ee == hipSuccess && any_ring)
        {
            ee = hipEventRecord(l->limit_done[slot], st);
            if (ee == hipSuccess)
                {
                    l->limit_used[slot] = true;
                    l->limit_next = (slot + 1) % gc_trk_loop::LIMIT_SLOTS;
                }
        }
    if (ee != hipSuccess)
        {
            // without its completion event the launch cannot be tracked: wait for it here, then commit
            (void)hipStreamSynchronize(st);
            out = gc_fail(GC_ERR_HIP, "gc_trk_loop_run: hipEventRecord failed: %s", hipGetErrorString(ee));
        }
    for (size_t k = 0; k < rings.size(); k++)
        {
            gc_status rs = gc_stream_end_read(rings[k], st, tickets[k]);
            if (rs != GC_OK && out == GC_OK) out = rs;
        }
    return out;
}

gc_status gc_trk_loop_run_dev(gc_trk_loop* l, int n_epochs, gc_loop_record* dev_records, void* stream)
{
    GC_REQUIRE(l && dev_records, "gc_trk_loop_run_dev: NULL argument");
    GC_REQUIRE(n_epochs > 0, "gc_trk_loop_run_dev: n_epochs must be > 0");
    gc_device_guard g(l->ctx->device);
    std::lock_guard<std::mutex> lk(l->ctx->mtx);
    // the records stay in HBM: the channels' positions are unknown to the host from here on
    for (auto& k : l->pos_known) k = 0;
    return loop_launch(l, n_epochs, dev_records, gc_pick_stream(l->ctx, stream), false);
}

gc_status gc_trk_loop_run(gc_trk_loop* l, int n_epochs, gc_loop_record* host_records)
{
    GC_REQUIRE(l && host_records, "gc_trk_loop_run: NULL argument");
    GC_REQUIRE(n_epochs > 0, "gc_trk_loop_run: n_epochs must be > 0");
    gc_device_guard g(l->ctx->device);
    std::lock_guard<std::mutex> lk(l->ctx->mtx);
    hipStream_t st = l->ctx->stream;
    const size_t n = (size_t)l->n_channels * n_epochs;
    if (n > l->recs_cap)
        {
            (void)hipFree(l->d_recs);
            l->d_recs = nullptr;
            l->recs_cap = 0;
            GC_HIP(hipMalloc(&l->d_recs, n * sizeof(gc_loop_record)));
            l->recs_cap = n;
        }
    gc_status s = loop_launch(l, n_epochs, l->d_recs, st, true);
    if (s != GC_OK) return s;
    GC_HIP(hipMemcpyAsync(host_records, l->d_recs, n * sizeof(gc_loop_record), hipMemcpyDeviceToHost, st));
    GC_HIP(hipStreamSynchronize(st));
    // ring channels: Tracking_sample_counter of the last record IS the absolute position (it started at sample_counter)
    for (int i = 0; i < l->n_channels; i++)
        if (l->streams[i] && l->started[i])
            {
                const gc_loop_record& last = host_records[(size_t)i * n_epochs + (n_epochs - 1)];
                l->pos_host[i] = last.sample_counter;
                l->pos_known[i] = 1;
                l->idle[i] = last.state == 0;  // loss of lock: the channel waits for the next gc_trk_loop_start
            }
    return GC_OK;
}

}  // extern "C"
