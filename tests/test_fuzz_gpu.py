"""Randomised differential test of the tracking multicorrelator against the oracle: many (window, NCO, tap, code, format,
mode) combinations per launch.  The default run is a few thousand channel-epochs; GNSSCORR_FUZZ_BATCHES=N widens it."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _one_batch(gctx, oracle, rng, fmt_name, high_dyn):
    import gnsscorr
    import torch
    n_taps = int(rng.integers(1, 9))
    L = int(rng.choice([1, 2, 31, 511, 1023, 2046, 4092, 8184, int(rng.integers(3, 12000))]))
    code = np.sign(rng.standard_normal(L)).astype(np.float32)
    code[code == 0] = 1
    shifts = np.sort(rng.uniform(-3.0, 3.0, n_taps)).astype(np.float32)
    if high_dyn:
        # the reference's high-dynamics resampler needs ordered taps whose sample delays stay inside the window
        shifts = np.sort(rng.uniform(-1.0, 1.0, n_taps)).astype(np.float32)
    n_sig = 16384
    raw = rng.standard_normal((n_sig, 2))
    fmt = getattr(gnsscorr, fmt_name)
    if fmt_name == "GC_IQ_F32":
        q = raw.astype(np.float32)
        sig = q.reshape(-1).view(np.complex64)
    elif fmt_name == "GC_IQ_I16":
        q = np.round(raw * 500).astype(np.int16)
        sig = q.astype(np.float32).reshape(-1).view(np.complex64)
    else:
        q = np.clip(np.round(raw * 30), -128, 127).astype(np.int8)
        sig = q.astype(np.float32).reshape(-1).view(np.complex64)
    d = torch.from_numpy(q).cuda()
    n_epochs = 48
    recs, refs = [], []
    for _ in range(n_epochs):
        n = int(rng.choice([0, 1, 2, 17, 255, 256, 257, 511, 512, 513, 1000, 2047, 4000, int(rng.integers(16 if high_dyn else 0, 6000))]))
        if high_dyn and n < 64:
            n = 64 + n
        off = int(rng.integers(0, n_sig - n + 1))
        rem_carr = float(np.float32(rng.uniform(-7, 7)))
        pstep = float(np.float32(rng.uniform(-3.1, 3.1) if rng.random() < 0.3 else rng.uniform(-0.02, 0.02)))
        rem_code = float(np.float32(rng.uniform(-3 * L, 3 * L) if rng.random() < 0.3 else rng.uniform(-2, 2)))
        cstep = float(np.float32(rng.uniform(0.0, 3.0) if rng.random() < 0.3 else rng.uniform(0.01, 0.6)))
        prate = float(np.float32(rng.uniform(-1e-7, 1e-7))) if high_dyn else 0.0
        crate = float(np.float32(rng.uniform(0.0, 1e-9))) if high_dyn else 0.0
        if high_dyn:
            # tap delays in samples must be < n (the reference memcpy()s n - delay floats)
            while n > 0 and (shifts[-1] - shifts[0]) / max(cstep, 1e-6) + n_taps >= n:
                cstep = float(np.float32(cstep * 2 + 0.05))
        recs.append(gnsscorr.epoch_params(off, rem_carr, pstep, rem_code, cstep, n, carr_phase_rate_step_rad=prate, code_phase_rate_step_chips=crate))
        refs.append(oracle.multicorrelator(sig[off:], code, shifts, np.float32(rem_carr), np.float32(pstep), np.float32(rem_code), np.float32(cstep), n,
            phase_rate_step=np.float32(prate), code_rate_step=np.float32(crate), high_dyn=high_dyn))
    b = gnsscorr.TrackingBatch(gctx, 1, n_taps, L, high_dyn=high_dyn)
    if fmt != gnsscorr.GC_IQ_F32:
        b.set_input_format(fmt)
    b.set_code(0, code, shifts)
    b.set_input_dev(0, d.data_ptr(), n_sig)
    if rng.random() < 0.3:
        b.set_slices(int(rng.integers(2, 9)))
    out = b.run(n_epochs, gnsscorr.epoch_params_array(recs))[0]
    b.close()
    scale = {"GC_IQ_F32": 1.0, "GC_IQ_I16": 500.0, "GC_IQ_I8": 30.0}[fmt_name]
    worst = 0.0
    for k in range(n_epochs):
        n = recs[k].n_samples
        tol = 3e-5 * scale * np.sqrt(max(n, 1)) + 1e-4 * float(np.max(np.abs(refs[k]))) if n else 0.0
        err = float(np.max(np.abs(out[k] - refs[k])))
        assert err <= tol, (fmt_name, high_dyn, n_taps, L, k, n, recs[k].sample_offset, shifts, out[k], refs[k])
        worst = max(worst, err / tol if tol else 0.0)
    return worst


def test_randomised_open_loop_parity(gctx, oracle):
    n_batches = int(os.environ.get("GNSSCORR_FUZZ_BATCHES", "60"))
    rng = np.random.Generator(np.random.PCG64(20261004))
    worst = 0.0
    for i in range(n_batches):
        fmt = ["GC_IQ_F32", "GC_IQ_F32", "GC_IQ_I16", "GC_IQ_I8"][i % 4]
        worst = max(worst, _one_batch(gctx, oracle, rng, fmt, high_dyn=(i % 5 == 4)))
    print("worst error / tolerance over %d channel-epochs: %.3f" % (n_batches * 48, worst))


def test_randomised_acquisition_parity(gctx, oracle):
    """PCPS engine against the oracle over random block sizes (every radix mix the planner produces, primes up to 61
    included), zero-padded and bit-transition layouts, 1-3 dwells, both statistics, 1-3 satellites."""
    import gnsscorr
    n_cases = int(os.environ.get("GNSSCORR_FUZZ_ACQ_CASES", "40"))
    rng = np.random.Generator(np.random.PCG64(77077))
    done = unsupported = 0
    while done < n_cases:
        spms = int(rng.choice([200, 256, 250, 341, 400, 500, 511, 610, 1000, 1023, 1024, 1331, 2000, 2046, int(rng.integers(100, 2500))]))
        fs = spms * 1000
        ms_per_code = int(rng.choice([1, 1, 4]))
        sampled_ms = int(rng.choice([1, 2, 4])) if ms_per_code == 1 else int(rng.choice([4, 8]))
        bt = bool(rng.random() < 0.2)
        max_dwells = 1 if bt else int(rng.integers(1, 4))
        use_cfar = bool(rng.random() < 0.5)
        dmax = int(rng.choice([500, 1000, 2000]))
        dstep = int(rng.choice([125, 250, 500]))
        n_sats = int(rng.integers(1, 4))
        c = dict(fs_in=fs, sampled_ms=sampled_ms, ms_per_code=ms_per_code, samples_per_ms=np.float32(fs) * np.float32(0.001),
            samples_per_code=float(spms * ms_per_code), samples_per_chip=max(1, spms // 1000 + 1), doppler_max=dmax, doppler_step=dstep,
            max_dwells=max_dwells, bit_transition_flag=bt, use_cfar=use_cfar)
        try:
            acq = gnsscorr.PcpsAcquisition(gctx, n_sats, **c)
        except gnsscorr.GnsscorrError as e:
            # only a block whose shortest row (a large prime factor) does not fit the LDS is refused, never mis-computed
            assert "factorisation" in str(e), e
            unsupported += 1
            continue
        consumed = acq.consumed_samples
        code_len = acq.fft_size // 2 if bt else consumed
        n_total = consumed * max_dwells
        x = ((rng.standard_normal(n_total) + 1j * rng.standard_normal(n_total)) * np.sqrt(0.5)).astype(np.complex64)
        orcs = []
        for s in range(n_sats):
            period = np.sign(rng.standard_normal(spms * ms_per_code)).astype(np.float32)
            code = np.tile(period, -(-code_len // period.size))[:code_len].astype(np.complex64)
            if s == 0:  # the first satellite is in the signal
                delay, dopp = int(rng.integers(0, period.size)), float(rng.uniform(-dmax, dmax))
                t = np.arange(n_total)
                x += (0.4 * np.tile(np.roll(period, delay), -(-n_total // period.size))[:n_total] * np.exp(2j * np.pi * dopp * t / fs)).astype(np.complex64)
            acq.set_local_code(s, code)
            p = oracle.pcps(**c)
            p.set_local_code(code)
            orcs.append(p)
        for d in range(max_dwells):
            blk = x[d * consumed:(d + 1) * consumed]
            res = acq.dwell(blk)
            for s in range(n_sats):
                q = orcs[s].core(blk)
                r = res[s]
                assert (r.indext, r.doppler_hz, r.doppler_index) == (q.indext, q.doppler, q.doppler_index), (c, s, d)
                assert r.mag == pytest.approx(q.mag, rel=1e-4) and r.test_statistics == pytest.approx(q.test_statistics, rel=2e-4), (c, s, d)
        grid, ref = acq.grid(0), orcs[0].grid()
        assert np.max(np.abs(grid - ref)) <= 1e-4 * ref.max(), c
        if max_dwells > 1:
            # the same search with its dwells enqueued back to back: they run in pairs (gc_acquisition.hip) -- not a bit may differ
            import torch
            d_x = torch.from_numpy(x.view(np.float32)).cuda()
            torch.cuda.synchronize()
            acq.reset()
            for d in range(max_dwells):
                acq.dwell_enqueue(d_x.data_ptr() + 8 * consumed * d, 0)
            again = acq.fetch_results(0)
            for s in range(n_sats):
                assert (again[s].indext, again[s].doppler_hz, again[s].mag, again[s].test_statistics) == (res[s].indext, res[s].doppler_hz, res[s].mag, res[s].test_statistics), (c, s)
            assert np.array_equal(acq.grid(0), grid), c
        acq.close()
        done += 1
    print("acquisition cases: %d checked, %d sizes refused" % (done, unsupported))


def test_randomised_complex_and_16bit_codes(gctx, oracle):
    """The complex-chip (Cpu_Multicorrelator) and 16-bit (Cpu_Multicorrelator_16sc) paths over random set-ups."""
    import gnsscorr
    import torch
    rng = np.random.Generator(np.random.PCG64(4242))
    for i in range(int(os.environ.get("GNSSCORR_FUZZ_BATCHES", "60")) // 2):
        sc16 = (i % 2 == 1)
        n_taps = int(rng.integers(1, 9))
        L = int(rng.choice([1, 7, 511, 1023, 2046, int(rng.integers(3, 7000))]))
        shifts = np.sort(rng.uniform(-2.0, 2.0, n_taps)).astype(np.float32)
        n_sig, n_epochs = 12000, 32
        raw = rng.standard_normal((n_sig, 2))
        b = gnsscorr.TrackingBatch(gctx, 1, n_taps, L)
        if sc16:
            q = np.round(raw * 40).astype(np.int16)
            code = np.stack([np.sign(rng.standard_normal(L)) * rng.integers(1, 3), rng.integers(-1, 2, L)], 1).astype(np.int16)
            b.set_16sc(True)
            b.set_code_16sc(0, code, shifts)
        else:
            q = raw.astype(np.float32)
            code = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
            b.set_complex_codes(True)
            b.set_code_complex(0, code, shifts)
        d = torch.from_numpy(q).cuda()
        b.set_input_dev(0, d.data_ptr(), n_sig)
        recs, refs = [], []
        for _ in range(n_epochs):
            n = int(rng.choice([0, 1, 255, 256, 257, 512, 1000, 4000, int(rng.integers(0, 5000))]))
            off = int(rng.integers(0, n_sig - n + 1))
            a = [np.float32(rng.uniform(-7, 7)), np.float32(rng.uniform(-0.05, 0.05)), np.float32(rng.uniform(-2 * L, 2 * L)), np.float32(rng.uniform(0.01, 1.5))]
            recs.append(gnsscorr.epoch_params(off, float(a[0]), float(a[1]), float(a[2]), float(a[3]), n))
            if sc16:
                refs.append(oracle.multicorrelator_16sc(q[off:], code, shifts, a[0], a[1], a[2], a[3], n))
            else:
                refs.append(oracle.multicorrelator_cc(q[off:].reshape(-1).view(np.complex64), code, shifts, a[0], a[1], a[2], a[3], n))
        out = b.run(n_epochs, gnsscorr.epoch_params_array(recs))[0]
        b.close()
        for k in range(n_epochs):
            n = recs[k].n_samples
            if sc16:
                ref, exact = refs[k]
                want = ref.astype(np.int32) if np.array_equal(ref.astype(np.int32), exact) else np.clip(exact, -32768, 32767)
                # a sample whose rotated value rounds the other way moves a sum by one chip: |re| + |im| <= 3 here
                assert np.abs(out[k].astype(np.int32) - want).max() <= 3 * 3, (i, k, n, out[k], ref, exact)
            else:
                tol = 6e-5 * np.sqrt(max(n, 1)) + 1e-4 * float(np.max(np.abs(refs[k]))) if n else 0.0
                assert float(np.max(np.abs(out[k] - refs[k]))) <= tol, (i, k, n, out[k], refs[k])


def test_randomised_ring_addressing(gctx, oracle):
    """Random ring capacities, push sizes and window positions: correlating from the ring (absolute sample numbers, wrap,
    mirrored head) must give what the same windows give from one linear buffer -- bit for bit when the capacity is a multiple of
    16 samples (the batched kernel counts a window's chunks from the 8-pair boundary below its first sample, so that every wave
    instruction reads whole cache lines: a window keeps its summation order where it keeps its position modulo 16 samples)."""
    import gnsscorr
    import torch
    rng = np.random.Generator(np.random.PCG64(909))
    code = oracle.gps_l1_ca_code(13).astype(np.float32)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    for case in range(12):
        fmt = [gnsscorr.GC_IQ_F32, gnsscorr.GC_IQ_I16, gnsscorr.GC_IQ_I8][case % 3]
        win = int(rng.integers(64, 3000))
        cap = int(rng.integers(2 * win, 6 * win))
        if case % 2 == 0:
            cap += -cap % 16
        total = int(rng.integers(3 * cap, 8 * cap))
        raw = rng.standard_normal((total, 2))
        if fmt == gnsscorr.GC_IQ_F32:
            q = raw.astype(np.float32)
            push_view = q.reshape(-1).view(np.complex64)
        elif fmt == gnsscorr.GC_IQ_I16:
            q = np.round(raw * 300).astype(np.int16)
            push_view = q
        else:
            q = np.clip(np.round(raw * 30), -128, 127).astype(np.int8)
            push_view = q
        d_lin = torch.from_numpy(q).cuda()
        lin = gnsscorr.TrackingBatch(gctx, 1, 3, 1023)
        ring = gnsscorr.IqStream(gctx, cap, win, iq_format=fmt)
        rb = gnsscorr.TrackingBatch(gctx, 1, 3, 1023)
        for b in (lin, rb):
            if fmt != gnsscorr.GC_IQ_F32:
                b.set_input_format(fmt)
            b.set_code(0, code, shifts)
            b.set_slices(1)
        lin.set_input_dev(0, d_lin.data_ptr(), total)
        rb.set_input_stream(0, ring)
        pushed = 0
        while pushed < total:
            m = int(min(rng.integers(1, cap + 1), total - pushed))
            assert ring.push(push_view[pushed:pushed + m]) == pushed
            pushed += m
            oldest, head, _ = ring.info()
            assert head == pushed and oldest == max(0, pushed - cap)
            recs = []
            for _ in range(6):
                n = int(rng.integers(0, win + 1))
                if head - oldest < n:
                    continue
                off = int(rng.integers(oldest, head - n + 1))
                recs.append(gnsscorr.epoch_params(off, float(np.float32(rng.uniform(0, 6))), float(np.float32(rng.uniform(-0.01, 0.01))),
                    float(np.float32(rng.uniform(-500, 500))), float(np.float32(rng.uniform(0.05, 0.5))), n))
            if not recs:
                continue
            params = gnsscorr.epoch_params_array([recs])
            got = rb.run(len(recs), params)
            want = lin.run(len(recs), params)
            if cap % 16 == 0:
                assert np.array_equal(got, want), (case, cap, win, pushed, [(r.sample_offset, r.n_samples) for r in recs])
            else:
                # another capacity moves wrapped windows against the 16-sample grid: same samples, another summation order
                assert np.max(np.abs(got - want)) <= 1e-5 * (1.0 + np.max(np.abs(want))), (case, cap, win, pushed)
        lin.close()
        rb.close()
        ring.close()


def test_randomised_level1_call_sequences(gctx, oracle):
    """The drop-in object under random call sequences: window length changes from call to call, the caller edits the
    shifts and the code table in place (pointers are retained, as in the reference), real / complex replicas alternate."""
    import gnsscorr
    rng = np.random.Generator(np.random.PCG64(31337))
    n_max = 30000
    sig = (rng.standard_normal(n_max + 64) + 1j * rng.standard_normal(n_max + 64)).astype(np.complex64)
    for obj in range(6):
        complex_code = (obj % 2 == 1)
        n_taps = int(rng.integers(1, 9))
        L = int(rng.choice([511, 1023, 2046, 4092]))
        code = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64) if complex_code else np.sign(rng.standard_normal(L)).astype(np.float32)
        shifts = np.sort(rng.uniform(-1, 1, n_taps)).astype(np.float32)
        out = np.zeros(n_taps, np.complex64)
        mc = gnsscorr.HipMulticorrelator(gctx) if complex_code else gnsscorr.HipMulticorrelatorRealCodes(gctx)
        if not complex_code:
            mc.set_high_dynamics_resampler(False)
        mc.init(n_max, n_taps)
        mc.set_local_code_and_taps(L, code, shifts)
        for call in range(25):
            n = int(rng.choice([0, 1, 100, 2000, 2048, 2049, 4000, 8000, 25000, int(rng.integers(0, n_max))]))
            off = int(rng.integers(0, 33))
            if rng.random() < 0.3:
                shifts[:] = np.sort(rng.uniform(-1, 1, n_taps)).astype(np.float32)   # edited in place, no setter call
            if rng.random() < 0.2:
                code[int(rng.integers(0, L))] *= -1                                   # so is the replica
            a = [np.float32(rng.uniform(-7, 7)), np.float32(rng.uniform(-0.02, 0.02)), np.float32(rng.uniform(-L, L)), np.float32(rng.uniform(0.02, 0.6))]
            mc.set_input_output_vectors(out, sig[off:])
            if complex_code:
                mc.Carrier_wipeoff_multicorrelator_resampler(float(a[0]), float(a[1]), float(a[2]), float(a[3]), n)
                ref = oracle.multicorrelator_cc(sig[off:], code, shifts, a[0], a[1], a[2], a[3], n)
            else:
                mc.Carrier_wipeoff_multicorrelator_resampler(float(a[0]), float(a[1]), 0.0, float(a[2]), float(a[3]), 0.0, n)
                ref = oracle.multicorrelator(sig[off:], code, shifts, a[0], a[1], a[2], a[3], n)
            tol = 6e-5 * np.sqrt(max(n, 1)) * (2.0 if complex_code else 1.0) + 1e-4 * float(np.max(np.abs(ref))) if n else 0.0
            assert float(np.max(np.abs(out - ref))) <= tol, (obj, call, n, off, out, ref)
        mc.free()
        mc.close()


def _random_loop_case(rng, veml=None, pilot=None, high_dyn=None):
    """A random signal + loop configuration for the closed-loop state machine (synchronisation, extension, pilot)."""
    from test_loop_sync_gpu import _stream
    force_pilot = pilot
    hd = high_dyn if high_dyn is not None else (int(rng.integers(2, 11)) if rng.uniform() < 0.3 else 0)  # Dll_Pll_Conf::high_dyn + smoother_length
    veml = bool(rng.integers(0, 2)) if veml is None else veml
    spc = 2 if veml else 1
    L = int(rng.integers(150, 1200)) * spc            # code samples per period
    fs = float(rng.integers(2000, 5000)) * 1000.0     # 1 ms code period
    N = int(round(fs * 0.001))
    code = (rng.integers(0, 2, L) * 2 - 1).astype(np.float32)
    data_code = (rng.integers(0, 2, L) * 2 - 1).astype(np.float32)
    kind = rng.choice(["secondary", "preamble", "single"]) if not force_pilot else "secondary"
    pilot = (kind == "secondary" and rng.uniform() < 0.7) if force_pilot is None else force_pilot
    ext = int(rng.integers(1, 6))
    y = dict(extend_correlation_symbols=ext, track_pilot=pilot, pll_bw_narrow_hz=float(rng.uniform(8, 20)), dll_bw_narrow_hz=float(rng.uniform(0.5, 2.0)),
        early_late_space_narrow_chips=float(rng.uniform(0.1, 0.4)), very_early_late_space_narrow_chips=float(rng.uniform(0.45, 0.6)))
    n_ep = 70 + 6 * ext
    shift = int(rng.integers(0, 12))
    if kind == "secondary":
        sec = "".join(rng.choice(["0", "1"], int(rng.integers(4, 31))))
        if len(set(sec)) == 1:
            sec = sec[:-1] + ("1" if sec[0] == "0" else "0")
        y.update(symbols_per_bit=int(rng.integers(1, 21)), secondary_code=sec)
        sym = np.roll(np.array([1.0 if c == "0" else -1.0 for c in sec]), shift)
    elif kind == "preamble":
        spb = int(rng.integers(2, 6))
        bits = [int(b) for b in rng.integers(0, 2, int(rng.integers(3, 7)))]
        pre = [1 if b else -1 for b in bits for _ in range(spb)]
        y.update(symbols_per_bit=spb, preamble_symbols=pre, bit_sync_min_time_s=float(rng.uniform(0.0, 0.02)))
        # random bits, the preamble, random bits -- in both polarities over the cases
        allbits = [int(b) for b in rng.integers(0, 2, 4)] + bits + [int(b) for b in rng.integers(0, 2, 40)]
        sym = np.roll(np.repeat(np.array(allbits) * 2.0 - 1.0, spb), shift) * (1.0 if rng.integers(0, 2) else -1.0)
    else:
        y.update(symbols_per_bit=1)
        sym = np.array([1.0])
    doppler = float(rng.uniform(-4000, 4000))
    delay = float(rng.integers(0, N))
    comps = [(code, sym, 1.0)]
    if pilot:
        comps.append((data_code, rng.integers(0, 2, 97) * 2.0 - 1.0, 1.0))
    x = _stream(comps, fs, N * (n_ep + 3), doppler, delay, float(rng.uniform(47, 52)), int(rng.integers(1, 1 << 30)), L * 1000.0)
    conf = dict(fs_in=fs, signal_carrier_freq_hz=1575.42e6, code_chip_rate_hz=L * 1000.0 / spc, code_period_s=0.001, carrier_lock_th=0.85,
        code_length_chips=L // spc, code_samples_per_chip=spc, vector_length=N, pull_in_time_s=0, veml=int(veml), pll_filter_order=int(rng.integers(2, 4)),
        dll_filter_order=int(rng.integers(1, 4)), enable_fll_pull_in=0, enable_fll_steady_state=int(rng.integers(0, 2)), cn0_samples=int(rng.integers(5, 21)),
        cn0_min=25, max_lock_fail=50, pll_bw_hz=float(rng.uniform(25, 45)), dll_bw_hz=float(rng.uniform(1, 3)), fll_bw_hz=10.0,
        early_late_space_chips=float(rng.uniform(0.2, 0.5)), very_early_late_space_chips=float(rng.uniform(0.55, 0.7)), acq_samplestamp_samples=0,
        sample_counter=0, acq_delay_samples=delay, acq_doppler_hz=doppler + float(rng.uniform(-3, 3)),
        high_dyn_smoother_length=hd)
    return x, code, (data_code if pilot else None), conf, y, n_ep, (5 if veml else 3)


def _agreeing_prefix(rec, ref):
    """Number of leading periods whose block boundaries agree.  The block length is floor(T_prn + remainder) in double precision
    (dll_pll_veml_tracking.cc:1009): when that sum falls within ~1e-9 of a whole sample, the device's libm and numpy may round it
    to different sides -- one sample, once in tens of thousands of periods -- and the two runs are different (equally valid) runs
    from there on."""
    for k in range(min(len(rec), len(ref))):
        if int(rec[k]["sample_counter"]) != ref[k]["sample_counter"]:
            assert abs(int(rec[k]["sample_counter"]) - ref[k]["sample_counter"]) == 1, (k, int(rec[k]["sample_counter"]), ref[k]["sample_counter"])
            return k
    return min(len(rec), len(ref))


def test_randomised_loop_state_machine(gctx, oracle):
    """Closed-loop engine vs the Python restatement over random synchronisation set-ups: random replicas and code lengths,
    3 / 5 taps, secondary codes of 4..30 symbols with and without a pilot + data component, preambles in both polarities,
    1..5 symbol integrations, loop-filter orders, FLL assistance, the high-dynamics kernels and rate smoothers.  State sequence and block
    boundaries must be identical."""
    import gnsscorr
    import torch
    from closed_loop_ref import run as ref_run
    from test_loop_sync_gpu import _compare, _conf, _sync
    seed = int(os.environ.get("GNSSCORR_FUZZ_SEED", "20240611"))
    rng = np.random.Generator(np.random.PCG64(seed + 5))
    reached = {2: 0, 3: 0, 4: 0}
    forks = compared = 0
    for case in range(int(os.environ.get("GNSSCORR_FUZZ_LOOP_CASES", "24"))):
        x, code, data_code, conf, y, n_ep, n_taps = _random_loop_case(rng)
        ref = ref_run(oracle, x, code, conf, n_ep, sync=y, data_code=data_code)
        d = torch.from_numpy(x.view(np.float32)).cuda()
        loop = gnsscorr.TrackingLoop(gctx, 1, code.size)
        loop.set_input_dev(0, d.data_ptr(), x.size)
        loop.set_sync(0, _sync(gnsscorr, y), data_code)
        loop.start(0, _conf(gnsscorr, **conf), code)
        # the same epochs in two launches: the state machine's state lives on the device between them
        k = int(rng.integers(1, n_ep))
        rec = np.concatenate([loop.run(k)[0], loop.run(n_ep - k)[0]])
        loop.close()
        keep_n = _agreeing_prefix(rec, ref)
        forks += keep_n < len(ref)
        compared += keep_n
        try:
            _compare(rec[:keep_n], ref[:keep_n], n_taps, tol=5e-3, abs_tol=4.0 * float(np.abs(x).max()))  # up to two samples across a chip edge
        except AssertionError as e:
            raise AssertionError("case %d (seed %d): %s\nsync %r" % (case, seed, e, y)) from e
        reached[int(rec["state"][len(ref) - 1])] = reached.get(int(rec["state"][len(ref) - 1]), 0) + 1
    assert reached[3] + reached[4] >= 4, reached  # a fair share of the cases did synchronise
    # several channels of one engine, each with its own signal, rates, synchronisation data and block length
    for group in range(int(os.environ.get("GNSSCORR_FUZZ_LOOP_GROUPS", "6"))):
        veml, pilot = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        hd = int(rng.integers(2, 11)) if rng.uniform() < 0.3 else 0  # one engine, one high_dyn mode (the smoother length may differ)
        cases = [_random_loop_case(rng, veml=veml, pilot=pilot, high_dyn=(int(rng.integers(2, 11)) if hd else 0)) for _ in range(int(rng.integers(2, 6)))]
        refs = [ref_run(oracle, x, code, conf, n_ep, sync=y, data_code=dc) for x, code, dc, conf, y, n_ep, _ in cases]
        loop = gnsscorr.TrackingLoop(gctx, len(cases), max(c[1].size for c in cases))
        keep = []
        for ch, (x, code, dc, conf, y, n_ep, _) in enumerate(cases):
            d = torch.from_numpy(x.view(np.float32)).cuda()
            keep.append(d)
            loop.set_input_dev(ch, d.data_ptr(), x.size)
            loop.set_sync(ch, _sync(gnsscorr, y), dc)
            loop.start(ch, _conf(gnsscorr, **conf), code)
        n_max = max(c[5] for c in cases)
        k = int(rng.integers(1, n_max))
        rec = np.concatenate([loop.run(k), loop.run(n_max - k)], axis=1)
        loop.close()
        for ch, (x, code, dc, conf, y, n_ep, n_taps) in enumerate(cases):
            keep_n = _agreeing_prefix(rec[ch], refs[ch])
            forks += keep_n < len(refs[ch])
            compared += keep_n
            try:
                _compare(rec[ch, :keep_n], refs[ch][:keep_n], n_taps, tol=5e-3, abs_tol=4.0 * float(np.abs(x).max()))  # up to two samples across a chip edge
            except AssertionError as e:
                raise AssertionError("group %d channel %d (seed %d): %s\nsync %r" % (group, ch, seed, e, y)) from e
            # past the end of its input a channel produces invalid records and keeps its state
            assert np.all(rec[ch, len(refs[ch]) + 1:]["valid"] == 0)
    assert forks <= 1 + compared // 10000, (forks, compared)  # one-sample forks of the block length are rare events


# Bit-generator state of seed 4242 (GNSSCORR_FUZZ_LOOP_CASES=500) just before group 43 of the loop fuzz: the soak run of
# round 1 failed there (channel 2, period 77: code_error_chips -0.5907 on the device, -0.6766 in the restatement).
SEED_4242_GROUP_43 = {"bit_generator": "PCG64", "state": {"state": 171073727812857451284120285393411247032,
    "inc": 219547397827042044247403813135624673319}, "has_uint32": 1, "uinteger": 2462273828}


def _edge_candidates(r, x, n_taps, width=3e-4):
    """Samples of the reference's window of one period whose chip phase step * n + shift_t - rem (the resampler's index
    expression, volk_gnsssdr_32f_xn_resampler_32f_xn.h:77-94) lies within `width` of a whole chip for tap t: the only
    samples a last-bit difference in the float32 NCO scalars can move to the neighbouring chip.  Returns [(t, n, |x[n]|)]."""
    rem, step, n = float(r["args"][2]), float(r["args"][3]), int(r["args"][4])
    i = np.arange(n, dtype=np.float64)
    out = []
    for t in range(n_taps):
        ph = step * i + float(r["shifts"][t]) - rem
        d = np.abs(ph - np.round(ph))
        for k in np.nonzero(d < width)[0]:
            out.append((t, int(k), float(abs(x[r["pos"] + int(k)]))))
    return out


def test_seed_4242_group_43_is_a_chip_edge_event(gctx, oracle):
    """The recorded fuzz failure, rebuilt from the generator state (pilot + data component, 5 taps, 3-symbol integration,
    20-symbol secondary code), run once.  What it establishes, period by period:
    * up to the first period whose correlator outputs differ the two runs agree to float rounding;
    * at that period every differing tap differs by 2 |x[n]| of ONE sample n (or two) whose chip phase sits within 3e-4 chip
      of an edge -- the signature of a float32 NCO scalar that differs in its last bit (device libm vs numpy in the double
      precision loop state), never a wrong replica, sign or window;
    * in EVERY period the device's discriminator outputs are the reference formulas applied to the device's own accumulators
      (checked inside _compare), and each accumulator is the signed sum of the device's own correlator outputs of the
      integration (save_correlation_results, :1073-1125) -- i.e. states 3 / 4 compute the right thing on what they are given;
    * the code error then differs by at most 2 |dE| / (|E| + |L|), which is the gate _compare uses.
    Measured on MI355X (round 2, gpurun_out/seed4242_report.txt): the runs agree to rounding up to period 57 (state 2); there the L
    tap alone differs, by 1.78375 = 2 |x[n]| (1.78317) of one of its 3 edge samples; at the recorded period 77 (3-symbol
    integration, the loop far off its peak: pe 15.6, pl 80.7 against |P| 533) the accumulators differ by at most 6.7 = 1.89 max|x|,
    which allows 2 sqrt(2) 6.7 / 96.2 = 0.197 in the code error; the observed gap is 0.086.  States 3 / 4 are not at fault."""
    import gnsscorr
    import torch
    from closed_loop_ref import run as ref_run
    from test_loop_sync_gpu import _compare, _conf, _sync
    rng = np.random.Generator(np.random.PCG64(0))
    rng.bit_generator.state = SEED_4242_GROUP_43
    veml, pilot = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    hd = int(rng.integers(2, 11)) if rng.uniform() < 0.3 else 0
    cases = [_random_loop_case(rng, veml=veml, pilot=pilot, high_dyn=(int(rng.integers(2, 11)) if hd else 0)) for _ in range(int(rng.integers(2, 6)))]
    assert (veml, pilot, hd, len(cases)) == (True, True, 0, 4)
    assert abs(cases[2][4]["pll_bw_narrow_hz"] - 15.100609601049388) < 1e-12 and cases[2][4]["secondary_code"] == "01110101010101000011"
    refs = [ref_run(oracle, x, code, conf, n_ep, sync=y, data_code=dc) for x, code, dc, conf, y, n_ep, _ in cases]
    loop = gnsscorr.TrackingLoop(gctx, len(cases), max(c[1].size for c in cases))
    keep = []
    for ch, (x, code, dc, conf, y, n_ep, _) in enumerate(cases):
        d = torch.from_numpy(x.view(np.float32)).cuda()
        keep.append(d)
        loop.set_input_dev(ch, d.data_ptr(), x.size)
        loop.set_sync(ch, _sync(gnsscorr, y), dc)
        loop.start(ch, _conf(gnsscorr, **conf), code)
    n_max = max(c[5] for c in cases)
    k_split = int(rng.integers(1, n_max))
    rec = np.concatenate([loop.run(k_split), loop.run(n_max - k_split)], axis=1)
    loop.close()
    x, code, dc, conf, y, n_ep, n_taps = cases[2]
    ref, g = refs[2], rec[2]
    keep_n = _agreeing_prefix(g, ref)
    assert keep_n == len(ref)  # no block-length fork in this case
    xmax = float(np.abs(x).max())
    gc = g["corr"][:, 0:2 * n_taps:2] + 1j * g["corr"][:, 1:2 * n_taps:2]
    rc = np.array([r["corr"] for r in ref])
    scale = np.abs(rc[:, n_taps // 2])
    d = np.abs(gc[:len(ref)] - rc)
    noise = 1e-4 * scale[:, None] + 5e-3          # float rounding of a 2000..5000-sample sum (an edge sample moves a tap by ~2)
    differs = np.nonzero(np.any(d > noise, axis=1))[0]
    assert differs.size > 0, "the device and the restatement agree everywhere: the recorded failure did not reproduce"
    k0 = int(differs[0])
    # (1) before k0: agreement to rounding, loop outputs included
    for k in range(k0):
        assert abs(float(g["code_error_chips"][k]) - ref[k]["cerr"]) < 1e-3 and abs(float(g["carrier_doppler_hz"][k]) - ref[k]["doppler"]) < 1e-3, k
    # (2) at k0: each differing tap is off by 2 |x[n]| for edge samples n of that tap
    cand = _edge_candidates(ref[k0], x, n_taps)
    report = ["first differing period %d (state %d), |P| %.1f, max|x| %.2f, %d edge candidates" % (k0, ref[k0]["state_in"], scale[k0], xmax, len(cand))]
    for t in range(n_taps):
        if d[k0, t] <= noise[k0, 0]:
            continue
        mine = [c for c in cand if c[0] == t]
        singles = [2 * c[2] for c in mine]
        pairs = [2 * (a[2] + b[2]) for i, a in enumerate(mine) for b in mine[i + 1:]] + [2 * abs(a[2] - b[2]) for i, a in enumerate(mine) for b in mine[i + 1:]]
        best = min(singles, key=lambda v: abs(v - d[k0, t])) if singles else float("nan")
        report.append("  tap %d: |dev - ref| = %.5f; nearest 2|x[n]| over %d edge samples = %.5f" % (t, d[k0, t], len(mine), best))
        # a pair of edge samples can add with any relative phase: bounded by the sum, attributed when one sample explains it
        explained = any(abs(v - d[k0, t]) <= 2e-3 * v + noise[k0, 0] for v in singles) or any(d[k0, t] <= v + noise[k0, 0] for v in pairs)
        assert mine and explained, "\n".join(report)
    # (3) every period: accumulators are the signed sums of the device's own outputs
    sec = y["secondary_code"]
    acc = np.zeros(5, np.complex64)
    sym = 0
    for k in range(len(ref)):
        st_in = ref[k]["state_in"]
        if st_in == 2:
            acc[:] = gc[k].astype(np.complex64)
            nxt_reset = ref[k]["state"] != 2
        else:
            sign = -1.0 if sec[sym] != "0" else 1.0
            sym = (sym + 1) % len(sec)
            acc = (acc + np.complex64(sign) * gc[k].astype(np.complex64)).astype(np.complex64)
            nxt_reset = st_in == 4
        ga = g["accu"][k, 0::2] + 1j * g["accu"][k, 1::2]
        assert np.max(np.abs(ga - acc)) <= 1e-6 * (1.0 + np.max(np.abs(acc))), (k, ga, acc)
        if nxt_reset:
            acc[:] = 0
            if st_in == 2:
                sym = 0
    # (4) the recorded period: the gap in the code error is what the accumulator gap allows
    k77 = 77
    r77 = ref[k77]
    ga = g["accu"][k77, 0::2] + 1j * g["accu"][k77, 1::2]
    dacc = float(np.max(np.abs(ga - r77["accu"])))
    pe = float(np.hypot(abs(r77["accu"][0]), abs(r77["accu"][1])))
    pl = float(np.hypot(abs(r77["accu"][4]), abs(r77["accu"][3])))
    gap = abs(float(g["code_error_chips"][k77]) - r77["cerr"])
    report.append("period 77: cerr device %.5f, restatement %.5f (gap %.5f); max |d accu| %.3f = %.2f max|x|; pe %.2f pl %.2f; bound 2*sqrt(2)*dacc/(pe+pl) = %.5f"
        % (float(g["code_error_chips"][k77]), r77["cerr"], gap, dacc, dacc / xmax, pe, pl, 2 * np.sqrt(2.0) * dacc / (pe + pl)))
    print("\n".join(report))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "seed4242_report.txt"), "w") as f:
            f.write("\n".join(report) + "\n")
    # pe, pl are each the norm of two taps, each tap off by at most dacc: |d pe| <= sqrt(2) dacc
    assert gap <= 2 * np.sqrt(2.0) * dacc / (pe + pl) + 1e-4, report[-1]
    _compare(g[:keep_n], ref[:keep_n], n_taps, tol=5e-3, abs_tol=4.0 * xmax)
