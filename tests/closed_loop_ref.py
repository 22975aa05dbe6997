"""Python restatement of the dll_pll_veml_tracking loop (states 1-4) on top of the CPU oracle correlator:
the checker for the device closed-loop engine (test infrastructure).

Follows dll_pll_veml_tracking.cc: pull-in :1568-1600, do_correlation_step :886-911, cn0_and_tracking_lock_status
:839-878, run_dll_pll :914-973, update_tracking_vars :998-1070, acquire_secondary :800-836, save_correlation_results
:1072-1125 and the states 2 / 3 / 4 of general_work :1601-1896, with the loop filters of
tracking_loop_filter.cc:104-245 (no last integrator) and tracking_FLL_PLL_filter.cc:55-133, float32 where the
reference uses float."""
import numpy as np

PI_2 = 6.283185307179586
f32 = np.float32


class LoopFilter:
    def __init__(self, T, bw, order=2):
        self.order = order
        self.design(T, bw)
        self.x = [f32(0)] * 4
        self.y = [f32(0)] * 4
        self.i = 3

    def design(self, T, bw):
        """update_coefficients: new coefficients, history untouched (set_noise_bandwidth / set_update_interval)."""
        order = self.order
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
        self.set_params(fll_bw, pll_bw, order)
        self.w = f32(0)
        self.x = f32(0)

    def set_params(self, fll_bw, pll_bw, order):
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


def run(oracle, x, code, conf, n_epochs, sync=None, data_code=None):
    """conf: dict with the gc_loop_conf fields; sync: dict with the gc_loop_sync_conf fields (secondary_code a '0'/'1'
    string, preamble_symbols a list of +1/-1) or None (never leaves state 2).  Returns a list of per-epoch dicts."""
    fs = conf["fs_in"]
    spc = conf["code_samples_per_chip"]
    veml = bool(conf["veml"])
    y = dict(extend_correlation_symbols=1, track_pilot=False, symbols_per_bit=2, secondary_code="", preamble_symbols=[],
        bit_sync_min_time_s=10.0, pll_bw_narrow_hz=0.0, dll_bw_narrow_hz=0.0, early_late_space_narrow_chips=0.0, very_early_late_space_narrow_chips=0.0)
    if sync:
        y.update(sync)
    ext = max(1, y["extend_correlation_symbols"])
    sec, pre = y["secondary_code"], list(y["preamble_symbols"])
    pilot = bool(y["track_pilot"])

    def taps_for(el, vel):
        if veml:
            return np.array([-vel * spc, -el * spc, 0.0, el * spc, vel * spc], np.float32)
        return np.array([-el * spc, 0.0, el * spc], np.float32)

    shifts = taps_for(f32(conf["early_late_space_chips"]), f32(conf["very_early_late_space_chips"]))
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
    acq_stamp = conf["acq_samplestamp_samples"]
    # pull-in
    diff = sample_counter - acq_stamp
    delta = float(diff) - conf["acq_delay_samples"]
    code_freq = conf["code_chip_rate_hz"]
    code_step = code_freq / fs
    T_prn = (1.0 / code_freq) * conf["code_length_chips"] * fs
    acq_code_phase = T_prn - np.fmod(delta, T_prn)
    offset = int(np.round(acq_code_phase))
    acc_phase -= step * offset
    sample_counter += offset
    pos = offset
    st = dict(state=2, cloop=True, corr_time=conf["code_period_s"], cur=N, prompt_buffer=[], cn0=0.0, lock_test=1.0, fail=0,
        perr=0.0, cerr=0.0, hist=[], current_symbol=0, ext_count=0, pull_in=True, P_old=np.complex64(0))
    accu = np.zeros(5, np.complex64)  # VE E P L VL
    out = []
    # high dynamics (:1016-1033, :1047-1064)
    sl = int(conf.get("high_dyn_smoother_length", 0))
    hd = dict(carr=[], code=[], carr_rate=0.0, code_rate=0.0)

    def smoothed(hist, value, samples, current):
        hist.append((value, samples))
        del hist[:-2 * sl]
        if len(hist) < 2 * sl:
            return current
        cp1 = sum(h[0] for h in hist[:sl]) / sl
        newer = [hist[2 * sl - k - 1] for k in range(sl)]
        cp2 = sum(h[0] for h in newer) / sl
        return (cp2 - cp1) / sum(h[1] for h in newer)

    def lock_status(coh_time):
        if len(st["prompt_buffer"]) < conf["cn0_samples"]:
            st["prompt_buffer"].append(np.complex64(accu[2]))
            return True
        pb = np.array(st["prompt_buffer"], np.complex64)
        st["prompt_buffer"] = []
        psig = np.mean(np.abs(pb.real.astype(np.float64))) ** 2
        ptot = np.mean(pb.real.astype(np.float64) ** 2 + pb.imag.astype(np.float64) ** 2)
        with np.errstate(all="ignore"):
            st["cn0"] = float(f32(10 * np.log10(psig / (ptot - psig)) - 10 * np.log10(coh_time)))
        si, sq = f32(0), f32(0)
        for v in pb:
            si = f32(si + v.real)
            sq = f32(sq + v.imag)
        st["lock_test"] = float(f32(f32(f32(si * si) - f32(sq * sq)) / f32(f32(si * si) + f32(sq * sq))))
        if not st["pull_in"]:
            if st["lock_test"] < conf["carrier_lock_th"] or st["cn0"] < conf["cn0_min"]:
                st["fail"] += 1
            elif st["fail"] > 0:
                st["fail"] -= 1
        if st["fail"] > conf["max_lock_fail"]:
            st["fail"] = 0
            return False
        return True

    def run_dll_pll():
        nonlocal doppler, code_freq
        P = accu[2]
        if st["cloop"]:
            st["perr"] = (float(np.arctan(f32(P.imag / P.real))) if P.real != 0 else 0.0) / PI_2
        else:
            st["perr"] = float(np.arctan2(f32(P.imag), f32(P.real))) / PI_2
        if (st["pull_in"] and conf["enable_fll_pull_in"]) or conf["enable_fll_steady_state"]:
            po = st["P_old"]
            dot = float(f32(f32(po.real * P.real) + f32(po.imag * P.imag)))
            cross = float(f32(f32(po.real * P.imag) - f32(P.real * po.imag)))
            ferr = np.arctan2(cross, dot) / st["corr_time"] / PI_2
            st["P_old"] = np.complex64(P)
            if st["pull_in"] and conf["enable_fll_pull_in"]:
                doppler = float(pll.get_carrier_error(ferr, 0.0, st["corr_time"]))
            else:
                doppler = float(pll.get_carrier_error(ferr, st["perr"], st["corr_time"]))
        else:
            doppler = float(pll.get_carrier_error(0.0, st["perr"], st["corr_time"]))
        if veml:
            pe = np.sqrt(float(f32(f32(accu[0].real * accu[0].real) + f32(accu[0].imag * accu[0].imag))) + float(f32(f32(accu[1].real * accu[1].real) + f32(accu[1].imag * accu[1].imag))))
            pl = np.sqrt(float(f32(f32(accu[4].real * accu[4].real) + f32(accu[4].imag * accu[4].imag))) + float(f32(f32(accu[3].real * accu[3].real) + f32(accu[3].imag * accu[3].imag))))
            st["cerr"] = 0.0 if pe + pl == 0 else (pe - pl) / (pe + pl)
        else:
            pe, pl = float(f32(abs(accu[1]))), float(f32(abs(accu[3])))
            st["cerr"] = 0.0 if pe + pl == 0 else 0.5 * (pe - pl) / (pe + pl)
        cfilt = float(dll.apply(st["cerr"]))
        code_freq = (1.0 + doppler / conf["signal_carrier_freq_hz"]) * conf["code_chip_rate_hz"] - cfilt

    def update_tracking_vars():
        nonlocal step, rem_carr, acc_phase, code_step, rem_code_samples, rem_code_chips
        T_prn = (1.0 / code_freq) * conf["code_length_chips"] * fs
        K = T_prn + rem_code_samples
        cur = int(np.floor(K))
        st["cur"] = cur
        step = PI_2 * doppler / fs
        if sl:
            hd["carr_rate"] = smoothed(hd["carr"], step, float(cur), hd["carr_rate"])
        adv = step * cur + 0.5 * hd["carr_rate"] * float(cur) * float(cur)
        rem_carr = f32(rem_carr + f32(adv))
        rem_carr = f32(np.fmod(rem_carr, f32(PI_2)))
        acc_phase -= adv
        code_step = code_freq / fs
        if sl:
            hd["code_rate"] = smoothed(hd["code"], code_step, float(cur), hd["code_rate"])
        rem_code_samples = K - cur
        rem_code_chips = code_freq * rem_code_samples / fs

    def save_correlation_results(taps):
        sign = 1.0
        if sec:
            if sec[st["current_symbol"]] != "0":
                sign = -1.0
            st["current_symbol"] = (st["current_symbol"] + 1) % len(sec)
        else:
            st["current_symbol"] = (st["current_symbol"] + 1) % y["symbols_per_bit"] if y["symbols_per_bit"] > 0 else 0
        for t in range(5):
            if not veml and t in (0, 4):
                continue
            v = taps[t if veml else t - 1]
            accu[t] = np.complex64(accu[t] + v) if sign > 0 else np.complex64(accu[t] - v)
        st["cloop"] = not pilot

    for _ in range(n_epochs):
        if st["state"] == 0 or pos + N > len(x):
            break
        if st["pull_in"] and conf["pull_in_time_s"] < (sample_counter - acq_stamp) // int(fs):
            st["pull_in"] = False
        args = (f32(rem_carr), f32(step), f32(f32(rem_code_chips) * f32(spc)), f32(f32(code_step) * f32(spc)), N)
        if sl:
            args = args + (f32(hd["carr_rate"]), f32(f32(hd["code_rate"]) * f32(spc)), True)
        pos_in, shifts_in = pos, shifts.copy()
        corr = oracle.multicorrelator(x[pos:], code, shifts, *args)
        pdata = oracle.multicorrelator(x[pos:], data_code, shifts[len(shifts) // 2:len(shifts) // 2 + 1], *args)[0] if pilot else corr[len(shifts) // 2]
        P = corr[len(shifts) // 2]
        valid, integrating, log_accu, log_count = 0, 0, None, 0
        state_in = st["state"]
        if st["state"] == 2:
            accu[:] = 0
            for t in range(len(corr)):
                accu[t if veml else t + 1] = corr[t]
            if not lock_status(conf["code_period_s"]):
                st["state"] = 0
                log_accu = accu.copy()
                hd.update(carr=[], code=[], carr_rate=0.0, code_rate=0.0)
            else:
                run_dll_pll()
                update_tracking_vars()
                valid, log_accu, log_count = 1, accu.copy(), st["ext_count"]
                nxt = False
                if sec:
                    st["hist"].append(P.real < 0)
                    st["hist"] = st["hist"][-len(sec):]
                    if len(st["hist"]) == len(sec):
                        cv = sum((1 if (neg == (c == "0")) else -1) for neg, c in zip(st["hist"], sec))
                        nxt = abs(cv) == len(sec)
                elif y["symbols_per_bit"] > 1:
                    t_trk = f32(float(f32(sample_counter - acq_stamp)) / fs)
                    if t_trk > y["bit_sync_min_time_s"] and pre:
                        st["hist"].append(P.real < 0)
                        st["hist"] = st["hist"][-len(pre):]
                        if len(st["hist"]) == len(pre):
                            cv = sum((-p if neg else p) for neg, p in zip(st["hist"], pre))
                            nxt = cv == len(pre)
                else:
                    nxt = True
                if nxt:
                    accu[:] = 0
                    st["hist"] = []
                    st["current_symbol"] = 0
                    if ext > 1:
                        st["ext_count"] = 0
                        st["corr_time"] = float(f32(f32(ext) * f32(conf["code_period_s"])))
                        st["state"] = 3
                        dll.design(st["corr_time"], y["dll_bw_narrow_hz"])
                        pll.set_params(conf["fll_bw_hz"], y["pll_bw_narrow_hz"], conf["pll_filter_order"])
                        shifts = taps_for(f32(y["early_late_space_narrow_chips"]), f32(y["very_early_late_space_narrow_chips"]))
                    else:
                        st["state"] = 4
        elif st["state"] == 3:
            update_tracking_vars()
            save_correlation_results(corr)
            st["ext_count"] += 1
            if st["ext_count"] == ext - 1:
                st["ext_count"] = 0
                st["state"] = 4
            valid, integrating, log_accu, log_count = 1, 1, accu.copy(), st["ext_count"]
        else:
            save_correlation_results(corr)
            if not lock_status(conf["code_period_s"] * ext):
                st["state"] = 0
                log_accu = accu.copy()
            else:
                run_dll_pll()
                update_tracking_vars()
                valid, log_accu, log_count = 1, accu.copy(), st["ext_count"]
                accu[:] = 0
                if ext > 1:
                    st["state"] = 3
        cur = st["cur"]
        sample_counter += cur
        pos += cur
        out.append(dict(corr=corr, doppler=doppler, code_freq=code_freq, cur=cur, sample_counter=sample_counter, cn0=st["cn0"], lock_test=st["lock_test"],
            perr=st["perr"], cerr=st["cerr"], rem_code_samples=rem_code_samples, acc_phase=acc_phase, state=st["state"], state_in=state_in, valid=valid,
            integrating=integrating, accu=log_accu, ext_count=log_count, prompt_data=pdata, args=args, pos=pos_in, shifts=shifts_in))
    return out
