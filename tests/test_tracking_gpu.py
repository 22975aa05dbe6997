"""GPU parity of the tracking multicorrelator (HIP, through the C ABI) against the
CPU oracle on identical seeded inputs.

Tolerance: north_star asks for <= 1e-4 relative (float32) on the E/P/L
accumulators; the metric is max_t |gpu[t] - oracle[t]| / |P_oracle| on
signal-bearing channels.  TOL below is that bound.
"""
import numpy as np
import pytest

from helpers import open_loop_params, rel_err, synth_stream

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _run_level1(gctx, sig, code, shifts, p, high_dyn=False, rate=(0.0, 0.0)):
    import gnsscorr
    mc = gnsscorr.HipMulticorrelatorRealCodes(gctx)
    mc.set_high_dynamics_resampler(high_dyn)
    n = p["n"]
    mc.init(2 * n, len(shifts))
    out = np.zeros(len(shifts), np.complex64)
    window = np.ascontiguousarray(sig[p["sample_offset"]:p["sample_offset"] + n])
    mc.set_local_code_and_taps(len(code), code, shifts)
    mc.set_input_output_vectors(out, window)
    assert mc.Carrier_wipeoff_multicorrelator_resampler(float(p["rem_carr"]), float(p["phase_step"]), float(rate[0]),
        float(p["rem_code"]), float(p["code_step"]), float(rate[1]), n)
    mc.free()
    mc.close()
    return out


@pytest.mark.parametrize("fs,n", [(2_000_000, 2000), (4_000_000, 4000), (25_000_000, 25000)])  # 2000: the single-workgroup call path
def test_gps_l1_single_correlator_matches_oracle(gctx, oracle, fs, n):
    """cfg1/cfg2 shape: GPS L1 C/A, 3 taps, one code period, drop-in class."""
    code = oracle.gps_l1_ca_code(7).astype(np.float32)
    sig, truth = synth_stream([code], fs, 4 * n, seed=1001, cn0_db_hz=(48.0, 48.0) if n < 4000 else (45.0, 45.0))
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    for p in open_loop_params(truth[0], fs, 1023, n, 3):
        ref = oracle.multicorrelator(sig[p["sample_offset"]:], code, shifts, p["rem_carr"], p["phase_step"],
            p["rem_code"], p["code_step"], n)
        got = _run_level1(gctx, sig, code, shifts, p)
        assert np.abs(ref[1]) > 0.5 * truth[0]["amp"] * n  # the prompt correlator sees the signal
        assert rel_err(got, ref, 1) <= TOL, (got, ref)


def test_batch_matches_oracle_many_channels(gctx, oracle):
    """32 channels x 8 epochs in one launch, shared RF stream, arbitrary (odd/even) window starts."""
    import gnsscorr
    import torch
    fs, n, n_ch, n_ep = 25_000_000, 25000, 32, 8
    codes = [oracle.gps_l1_ca_code(prn).astype(np.float32) for prn in range(1, n_ch + 1)]
    sig, truth = synth_stream(codes, fs, n * n_ep + 64, seed=1002, cn0_db_hz=(42.0, 48.0))
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    d_sig = torch.from_numpy(sig.view(np.float32)).cuda()
    batch = gnsscorr.TrackingBatch(gctx, n_ch, 3, 1023)
    params = []
    refs = np.zeros((n_ch, n_ep, 3), np.complex64)
    for ch in range(n_ch):
        batch.set_code(ch, codes[ch], shifts)
        batch.set_input_dev(ch, d_sig.data_ptr(), sig.size)
        recs = []
        for k, p in enumerate(open_loop_params(truth[ch], fs, 1023, n, n_ep)):
            off = p["sample_offset"] + (ch + k) % 7  # odd and even starts: exercises the 8-byte aligned head
            p = dict(p, sample_offset=off)
            recs.append(gnsscorr.epoch_params(off, float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]),
                float(p["code_step"]), n))
            refs[ch, k] = oracle.multicorrelator(sig[off:], codes[ch], shifts, p["rem_carr"], p["phase_step"],
                p["rem_code"], p["code_step"], n)
        params.append(recs)
    got = batch.run(n_ep, gnsscorr.epoch_params_array(params))
    batch.close()
    worst = 0.0
    for ch in range(n_ch):
        for k in range(n_ep):
            worst = max(worst, rel_err(got[ch, k], refs[ch, k], 1))
    assert worst <= TOL, worst


@pytest.mark.parametrize("fmt", ["f32", "i16", "i8"])
def test_samples_next_to_a_window_never_enter_it(gctx, oracle, fmt):
    """The batched kernel fetches the ragged first / last chunk of a window like the interior ones -- whole 16-byte pieces, up to 15
    samples before the window and up to a chunk behind it, wherever those lie inside the channel's buffer -- and drops them in
    registers (trk_device.hpp, mask_window).  Here every sample outside the windows is poison (NaN / +-Inf for float input, full
    scale for the integer formats): windows at the very start of the buffer, at its very end, odd offsets, lengths that end anywhere
    in a chunk, one and several slices.  Any leak shows as a NaN or as an error far beyond the tolerance."""
    import gnsscorr
    import torch
    fs = 25_000_000
    code = oracle.gps_l1_ca_code(11).astype(np.float32)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    rng = np.random.Generator(np.random.PCG64(4711))
    lengths = [25000, 24576, 513, 511, 1, 7000, 12345]
    gap = 40
    starts, pos = [], 0  # the first window starts at sample 0, the last one ends with the buffer
    for n in lengths:
        starts.append(pos)
        pos += n + gap + int(rng.integers(0, 9))
    total = starts[-1] + lengths[-1]
    sig, truth = synth_stream([code], fs, total, seed=77, cn0_db_hz=(50.0, 50.0))
    inside = np.zeros(total, bool)
    for s0, n in zip(starts, lengths):
        inside[s0:s0 + n] = True
    if fmt == "f32":
        raw = sig.copy()
        poison = np.array([np.nan + 1j * np.inf, -np.inf + 1j * np.nan, np.nan + 1j * np.nan], np.complex64)
        raw[~inside] = poison[np.arange((~inside).sum()) % 3]
        dev, iq_fmt, clean = torch.from_numpy(raw.view(np.float32)).cuda(), gnsscorr.GC_IQ_F32, sig
    else:
        scale, lim, dt = (300.0, 32767, np.int16) if fmt == "i16" else (25.0, 127, np.int8)
        q = np.clip(np.round(sig.view(np.float32).reshape(-1, 2) * scale), -lim, lim).astype(dt)
        clean = (q[:, 0].astype(np.float32) + 1j * q[:, 1].astype(np.float32)).astype(np.complex64)
        q[~inside] = [lim, -lim]
        dev, iq_fmt = torch.from_numpy(q).cuda(), (gnsscorr.GC_IQ_I16 if fmt == "i16" else gnsscorr.GC_IQ_I8)
    recs, refs = [], []
    for s0, n in zip(starts, lengths):
        a = (float(np.float32(rng.uniform(0, 6))), float(np.float32(rng.uniform(-0.01, 0.01))), float(np.float32(rng.uniform(-400, 400))),
            float(np.float32(1.023e6 / fs)))
        recs.append(gnsscorr.epoch_params(s0, a[0], a[1], a[2], a[3], n))
        refs.append(oracle.multicorrelator(clean[s0:], code, shifts, np.float32(a[0]), np.float32(a[1]), np.float32(a[2]), np.float32(a[3]), n))
    for slices in (1, 3):
        b = gnsscorr.TrackingBatch(gctx, 1, 3, 1023)
        if iq_fmt != gnsscorr.GC_IQ_F32:
            b.set_input_format(iq_fmt)
        b.set_code(0, code, shifts)
        b.set_input_dev(0, dev.data_ptr(), total)
        b.set_slices(slices)
        got = b.run(len(recs), gnsscorr.epoch_params_array([recs]))[0]
        b.close()
        assert np.all(np.isfinite(got.view(np.float32))), (fmt, slices)
        for k, ref in enumerate(refs):
            tol = 1e-4 * float(np.max(np.abs(ref))) + 6e-5 * np.sqrt(lengths[k]) * float(np.max(np.abs(clean)))
            assert float(np.max(np.abs(got[k] - ref))) <= tol, (fmt, slices, k, got[k], ref)


def test_sliced_epochs_equal_unsliced(gctx, oracle):
    """Cutting an epoch into slices only changes the summation order."""
    import gnsscorr
    import torch
    fs, n = 25_000_000, 25000
    code = oracle.gps_l1_ca_code(3).astype(np.float32)
    sig, truth = synth_stream([code], fs, 2 * n, seed=5, cn0_db_hz=(45.0, 45.0))
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    d_sig = torch.from_numpy(sig.view(np.float32)).cuda()
    p = open_loop_params(truth[0], fs, 1023, n, 1)[0]
    rec = gnsscorr.epoch_params_array([gnsscorr.epoch_params(3, float(p["rem_carr"]), float(p["phase_step"]),
        float(p["rem_code"]), float(p["code_step"]), n)])
    outs = []
    for slices in (1, 2, 7, 24):
        b = gnsscorr.TrackingBatch(gctx, 1, 3, 1023)
        b.set_code(0, code, shifts)
        b.set_input_dev(0, d_sig.data_ptr(), sig.size)
        b.set_slices(slices)
        outs.append(b.run(1, rec)[0, 0])
        b.close()
    ref = oracle.multicorrelator(sig[3:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n)
    for o in outs:
        assert rel_err(o, ref, 1) <= TOL


def test_level1_calls_from_many_threads_are_batched_and_exact(gctx, oracle):
    """The drop-in object driven the way the flowgraph drives it: one correlator per thread, every thread reading its own position
    of ONE stream buffer, all calling at once (ctypes releases the GIL).  Every result must equal the same call made alone (the
    slice count depends on the window length only and every path keeps the caller pointer's alignment phase, so a batch changes
    nothing), and the same holds with the buffer registered (no staging copies)."""
    import threading
    import gnsscorr
    fs, n, n_thr, n_calls = 25_000_000, 25000, 12, 6
    code = oracle.gps_l1_ca_code(4).astype(np.float32)
    sig, truth = synth_stream([code], fs, 4 * n, seed=99, cn0_db_hz=(45.0, 45.0))
    shifts = [np.array([-0.5, 0.0, 0.5], np.float32) for _ in range(n_thr)]
    outs = [np.zeros(3, np.complex64) for _ in range(n_thr)]
    offs = [(977 * t) % (2 * n) for t in range(n_thr)]
    mcs = []
    for t in range(n_thr):
        mc = gnsscorr.HipMulticorrelatorRealCodes(gctx)
        mc.set_high_dynamics_resampler(False)
        mc.init(2 * n, 3)
        mc.set_local_code_and_taps(1023, code, shifts[t])
        mc.set_input_output_vectors(outs[t], sig[offs[t]:])
        mcs.append(mc)
    args = [(float(np.float32(0.1 * t)), float(np.float32(2e-3)), 0.0, float(np.float32(-3.25 * t)), float(np.float32(1023.0 / n)), 0.0, n) for t in range(n_thr)]
    alone = []
    for t in range(n_thr):
        mcs[t].Carrier_wipeoff_multicorrelator_resampler(*args[t])
        alone.append(outs[t].copy())
        ref = oracle.multicorrelator(sig[offs[t]:], code, shifts[t], np.float32(args[t][0]), np.float32(args[t][1]), np.float32(args[t][3]), np.float32(args[t][4]), n)
        assert float(np.max(np.abs(alone[t] - ref))) <= 6e-5 * np.sqrt(n) + 1e-4 * float(np.max(np.abs(ref)))
    for registered in (False, True):
        if registered:
            gctx.register_host_buffer(sig)
        b0, r0, s0, _ = gctx.correlator_batch_stats()
        bad = []
        barrier = threading.Barrier(n_thr)

        def work(t):
            barrier.wait()
            for _ in range(n_calls):
                outs[t][:] = 0
                mcs[t].Carrier_wipeoff_multicorrelator_resampler(*args[t])
                if not np.array_equal(outs[t], alone[t]):
                    bad.append((t, outs[t].copy()))

        threads = [threading.Thread(target=work, args=(t,)) for t in range(n_thr)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        b1, r1, s1, _ = gctx.correlator_batch_stats()
        assert not bad, bad[:2]
        # every call went through the batcher; how many shared a launch depends on how the interpreter interleaves the threads
        # (the C++ self-test measures the batching itself with native threads)
        assert r1 - r0 == n_thr * n_calls and b1 - b0 <= r1 - r0
        if registered:
            gctx.unregister_host_buffer(sig)
    for mc in mcs:
        mc.close()
