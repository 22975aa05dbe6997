"""Python restatement of the dll_pll_veml_tracking loop (states 1-2) on top of the CPU oracle correlator:
the checker for the device closed-loop engine (test infrastructure).

Follows dll_pll_veml_tracking.cc: pull-in :1568-1600, do_correlation_step :886-897, cn0_and_tracking_lock_status
:839-878, run_dll_pll :914-973, update_tracking_vars :998-1070, with the loop filters of
tracking_loop_filter.cc:104-245 (no last integrator) and tracking_FLL_PLL_filter.cc:55-133, float32 where the
reference uses float."""
import numpy as np

PI_2 = 6.283185307179586
f32 = np.float32


class LoopFilter:
    def __init__(self, T, bw, order=2):
        T = f32(T)
        zeta = f32(1.0 / np.sqrt(2.0))
        if order == 1:
            self.b, self.a = [f32(f32(bw * 4.0))], []
        elif order == 3:
            wn = f32(bw / 0.7845)
            a3, b3 = f32(1.1), f32(2.4)
            g1, g2, g3 = f32(wn * wn * wn), f32(a3 * wn * wn), f32(b3 * wn)
            self.b = [f32(float(g3) + float(T) / 2.0 * (float(g2) + float(T) / 2.0 * float(g1))), f32(float(f32(f32(g1 * T) * T)) / 2.0 - 2.0 * float(g3)),
                f32(float(g3) + float(T) / 2.0 * (-float(g2) + float(T) / 2.0 * float(g1)))]
            self.a = [f32(2.0), f32(-1.0)]
        else:
            wn = f32(float(f32(bw)) * (8.0 * float(zeta)) / (4.0 * float(zeta) * float(zeta) + 1.0))
            g1 = f32(wn * wn)
            g2 = f32(float(wn) * 2.0 * float(zeta))
            self.b = [f32(float(f32(g1 * T)) / 2.0 + float(g2)), f32(float(f32(g1 * T)) / 2.0 - float(g2))]
            self.a = [f32(1.0)]
        self.x = [f32(0)] * 4
        self.y = [f32(0)] * 4
        self.i = 3

    def apply(self, v):
        r = f32(0)
        for k, a in enumerate(self.a):
            r = f32(r + f32(a * self.y[(self.i + k) % 4]))
        self.i = (self.i - 1) % 4
        self.x[self.i] = f32(v)
        for k, b in enumerate(self.b):
            r = f32(r + f32(b * self.x[(self.i + k) % 4]))
        self.y[self.i] = r
        return r


class Pll:
    def __init__(self, fll_bw, pll_bw, order=3):
        self.order = order
        self.a2 = f32(1.414)
        if order == 3:
            self.b3, self.a3 = f32(2.4), f32(1.1)
            self.w0p = f32(pll_bw / 0.7845)
            self.w0p2 = f32(self.w0p * self.w0p)
            self.w0p3 = f32(self.w0p2 * self.w0p)
            self.w0f = f32(fll_bw / 0.53)
            self.w0f2 = f32(self.w0f * self.w0f)
        else:
            self.w0p = f32(pll_bw / 0.53)
            self.w0p2 = f32(self.w0p * self.w0p)
            self.w0f = f32(fll_bw / 0.25)
        self.w = f32(0)
        self.x = f32(0)

    def initialize(self, doppler):
        if self.order == 3:
            self.x, self.w = f32(2.0 * float(f32(doppler))), f32(0)
        else:
            self.w, self.x = f32(doppler), f32(0)

    def get_carrier_error(self, fll, pll, T):
        fll, pll, T = f32(fll), f32(pll), f32(T)
        if self.order == 3:
            self.w = f32(self.w + f32(T * f32(f32(self.w0p3 * pll) + f32(self.w0f2 * fll))))
            self.x = f32(float(self.x) + float(T) * (0.5 * float(self.w) + float(f32(f32(self.a2 * self.w0f) * fll)) + float(f32(f32(self.a3 * self.w0p2) * pll))))
            return f32(0.5 * float(self.x) + float(f32(f32(self.b3 * self.w0p) * pll)))
        w_new = f32(f32(self.w + f32(f32(pll * self.w0p2) * T)) + f32(f32(fll * self.w0f) * T))
        err = f32(0.5 * float(f32(w_new + self.w)) + float(f32(f32(self.a2 * self.w0p) * pll)))
        self.w = w_new
        return err


def run(oracle, x, code, conf, n_epochs):
    """conf: dict with the gc_loop_conf fields.  Returns a list of per-epoch dicts (valid epochs only)."""
    fs = conf["fs_in"]
    spc = conf["code_samples_per_chip"]
    veml = bool(conf["veml"])
    if veml:
        shifts = np.array([-conf["very_early_late_space_chips"] * spc, -conf["early_late_space_chips"] * spc, 0.0,
            conf["early_late_space_chips"] * spc, conf["very_early_late_space_chips"] * spc], np.float32)
    else:
        shifts = np.array([-conf["early_late_space_chips"] * spc, 0.0, conf["early_late_space_chips"] * spc], np.float32)
    N = conf["vector_length"]
    dll = LoopFilter(conf["code_period_s"], conf["dll_bw_hz"], conf["dll_filter_order"])
    pll = Pll(conf["fll_bw_hz"], conf["pll_bw_hz"], conf["pll_filter_order"])
    pll.initialize(conf["acq_doppler_hz"])
    doppler = conf["acq_doppler_hz"]
    step = PI_2 * doppler / fs
    rem_carr = f32(0.0)
    rem_code_samples = 0.0
    rem_code_chips = 0.0
    acc_phase = 0.0
    sample_counter = conf["sample_counter"]
    # pull-in
    diff = sample_counter - conf["acq_samplestamp_samples"]
    delta = float(diff) - conf["acq_delay_samples"]
    code_freq = conf["code_chip_rate_hz"]
    code_step = code_freq / fs
    T_prn = (1.0 / code_freq) * conf["code_length_chips"] * fs
    acq_code_phase = T_prn - np.fmod(delta, T_prn)
    offset = int(np.round(acq_code_phase))
    acc_phase -= step * offset
    sample_counter += offset
    pos = offset
    prompt_buffer = []
    cn0, lock_test = 0.0, 1.0
    fail = 0
    pi = 1 if veml else 0
    out = []
    for _ in range(n_epochs):
        if pos + N > len(x):
            break
        corr = oracle.multicorrelator(x[pos:], code, shifts, f32(rem_carr), f32(step), f32(f32(rem_code_chips) * f32(spc)), f32(f32(code_step) * f32(spc)), N)
        P = corr[1 + pi]
        E, L = corr[pi], corr[2 + pi]
        if len(prompt_buffer) < conf["cn0_samples"]:
            prompt_buffer.append(P)
        else:
            pb = np.array(prompt_buffer, np.complex64)
            prompt_buffer = []
            psig = np.mean(np.abs(pb.real.astype(np.float64))) ** 2
            ptot = np.mean(pb.real.astype(np.float64) ** 2 + pb.imag.astype(np.float64) ** 2)
            cn0 = float(f32(10 * np.log10(psig / (ptot - psig)) - 10 * np.log10(conf["code_period_s"])))
            si, sq = f32(0), f32(0)
            for v in pb:
                si = f32(si + v.real)
                sq = f32(sq + v.imag)
            lock_test = float(f32(f32(f32(si * si) - f32(sq * sq)) / f32(f32(si * si) + f32(sq * sq))))
        perr = (float(np.arctan(f32(P.imag / P.real))) if P.real != 0 else 0.0) / PI_2
        doppler = float(pll.get_carrier_error(0.0, perr, conf["code_period_s"]))
        if veml:
            pe = np.sqrt(float(abs(corr[0]) ** 2) + float(abs(E) ** 2))
            pl = np.sqrt(float(abs(corr[4]) ** 2) + float(abs(L) ** 2))
            cerr = 0.0 if pe + pl == 0 else (pe - pl) / (pe + pl)
        else:
            pe, pl = float(f32(abs(E))), float(f32(abs(L)))
            cerr = 0.0 if pe + pl == 0 else 0.5 * (pe - pl) / (pe + pl)
        cfilt = float(dll.apply(cerr))
        code_freq = (1.0 + doppler / conf["signal_carrier_freq_hz"]) * conf["code_chip_rate_hz"] - cfilt
        T_prn = (1.0 / code_freq) * conf["code_length_chips"] * fs
        K = T_prn + rem_code_samples
        cur = int(np.floor(K))
        step = PI_2 * doppler / fs
        rem_carr = f32(rem_carr + f32(step * cur))
        rem_carr = f32(np.fmod(rem_carr, f32(PI_2)))
        acc_phase -= step * cur
        code_step = code_freq / fs
        rem_code_samples = K - cur
        rem_code_chips = code_freq * rem_code_samples / fs
        sample_counter += cur
        pos += cur
        out.append(dict(corr=corr, doppler=doppler, code_freq=code_freq, cur=cur, sample_counter=sample_counter, cn0=cn0, lock_test=lock_test,
            perr=perr, cerr=cerr, rem_code_samples=rem_code_samples, acc_phase=acc_phase))
    return out
