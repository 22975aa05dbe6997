"""GPU parity of the tracking multicorrelator for the other signals / modes of the hot path:
Galileo E1 (5 taps, 2 samples per chip, 4 ms), BeiDou B1I, the high-dynamics kernels, window
edge cases, and size-independent properties at BASELINE's full sizes."""
import os

import numpy as np
import pytest

from helpers import open_loop_params, rel_err, synth_stream

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def _batch_one(gctx, sig, code, shifts, recs, high_dyn=False, slices=0):
    import gnsscorr
    import torch
    d_sig = torch.from_numpy(np.ascontiguousarray(sig).view(np.float32)).cuda()
    b = gnsscorr.TrackingBatch(gctx, 1, len(shifts), len(code), high_dyn=high_dyn)
    b.set_code(0, code, shifts)
    b.set_input_dev(0, d_sig.data_ptr(), sig.size)
    if slices:
        b.set_slices(slices)
    out = b.run(len(recs), gnsscorr.epoch_params_array(recs))
    b.close()
    return out[0]


def test_galileo_e1_five_taps_4ms(gctx, oracle):
    """cfg3 shape: sinBOC(1,1) replica at 2 samples/chip (L = 8184), VE/E/P/L/VL, N = 100000."""
    import gnsscorr
    e1b = np.load(os.path.join(G, "galileo_e1_codes.npz"))["e1b"]
    code = oracle.galileo_e1_sinboc11(e1b[10])
    fs, n = 25_000_000, 100000
    sig, truth = synth_stream([code], fs, 2 * n + 16, seed=1003, cn0_db_hz=(44.0, 44.0), chip_rate=2 * 1.023e6)
    shifts = np.array([-1.2, -0.3, 0.0, 0.3, 1.2], np.float32)  # +-0.6, +-0.15 chips x 2 samples/chip
    recs, refs = [], []
    for k, p in enumerate(open_loop_params(truth[0], fs, 8184, n, 2)):
        off = p["sample_offset"] + k  # second window starts on an odd sample (0.08 chip off: still on the BOC main peak)
        recs.append(gnsscorr.epoch_params(off, float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n))
        refs.append(oracle.multicorrelator(sig[off:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n))
    got = _batch_one(gctx, sig, code, shifts, recs)
    got_sliced = _batch_one(gctx, sig, code, shifts, recs, slices=9)
    for k in range(2):
        assert abs(refs[k][2]) > 0.5 * truth[0]["amp"] * n
        assert rel_err(got[k], refs[k], 2) <= TOL
        assert rel_err(got_sliced[k], refs[k], 2) <= TOL


def test_launch_window_smaller_than_the_code_table(gctx, oracle):
    """A launch's LDS code window is sized by what one slice of a NOMINAL epoch touches (L = 8184, N = 100000: two slices, ~4.3 k floats
    instead of 8248), so a record outside that bound has neither its window nor the whole table in LDS and reads the table from global
    memory (trk_epoch, GTAB).  One batch of four records on one channel: a nominal one (windowed), one whose code step is 2.6 x nominal
    (its slices touch ~11000 chips: global path), one with a negative step (not monotone: global path), and a nominal one again; all
    against the oracle, and equal to what the pre-sizing launch (set_slices(-1): whole table in LDS) gives."""
    import gnsscorr
    e1b = np.load(os.path.join(G, "galileo_e1_codes.npz"))["e1b"]
    code = oracle.galileo_e1_sinboc11(e1b[3])
    fs, n = 25_000_000, 100000
    sig, truth = synth_stream([code], fs, 4 * n + 64, seed=1404, cn0_db_hz=(45.0, 45.0), chip_rate=2 * 1.023e6)
    shifts = np.array([-1.2, -0.3, 0.0, 0.3, 1.2], np.float32)
    nominal = open_loop_params(truth[0], fs, 8184, n, 4)
    steps = [float(nominal[0]["code_step"]), 2.6 * float(nominal[1]["code_step"]), -float(nominal[2]["code_step"]), float(nominal[3]["code_step"])]
    recs, refs = [], []
    for k, p in enumerate(nominal):
        off = p["sample_offset"] + (3 if k == 1 else 0)
        recs.append(gnsscorr.epoch_params(off, float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), steps[k], n))
        refs.append(oracle.multicorrelator(sig[off:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], np.float32(steps[k]), n))
    # device-resident records (gc_trk_batch_run_dev): the host knows only the nominal length, so the launch gets the small window; with a
    # host-side array (gc_trk_batch_run) the engine reads the records' code steps and sizes the window for the largest
    import torch
    d_sig = torch.from_numpy(np.ascontiguousarray(sig).view(np.float32)).cuda()
    d_par = torch.from_numpy(gnsscorr.epoch_params_array([recs]).view(np.uint8)).cuda()
    d_out = torch.zeros(4 * 5, 2, dtype=torch.float32, device="cuda")
    b = gnsscorr.TrackingBatch(gctx, 1, 5, len(code))
    b.set_code(0, code, shifts)
    b.set_input_dev(0, d_sig.data_ptr(), sig.size)
    b.set_nominal_length(n)
    torch.cuda.synchronize()
    b.run_dev(4, d_par.data_ptr(), d_out.data_ptr())
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(np.complex64).reshape(4, 5)
    b.close()
    host = _batch_one(gctx, sig, code, shifts, recs)
    old = _batch_one(gctx, sig, code, shifts, recs, slices=-1)
    scale = abs(refs[0][2])
    assert scale > 0.5 * truth[0]["amp"] * n
    for k in range(4):
        assert np.max(np.abs(got[k] - refs[k])) <= TOL * scale, k
        assert np.max(np.abs(host[k] - refs[k])) <= TOL * scale, k
        assert np.max(np.abs(old[k] - refs[k])) <= TOL * scale, k
    # the two nominal records are on the correlation peak; the others decorrelate (wrong chip rate / direction)
    assert abs(got[3][2]) > 0.5 * truth[0]["amp"] * n and abs(got[1][2]) < 0.1 * abs(got[0][2]) and abs(got[2][2]) < 0.1 * abs(got[0][2])


def test_beidou_b1i_three_taps(gctx, oracle):
    import gnsscorr
    code = oracle.beidou_b1i_code(6).astype(np.float32)
    fs, n = 25_000_000, 25000
    sig, truth = synth_stream([code], fs, 3 * n, seed=1005, cn0_db_hz=(44.0, 44.0), chip_rate=2.046e6, carrier_freq=1561.098e6)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    recs, refs = [], []
    for p in open_loop_params(truth[0], fs, 2046, n, 3):
        recs.append(gnsscorr.epoch_params(p["sample_offset"], float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n))
        refs.append(oracle.multicorrelator(sig[p["sample_offset"]:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n))
    got = _batch_one(gctx, sig, code, shifts, recs)
    for k in range(3):
        assert rel_err(got[k], refs[k], 1) <= TOL


def test_oracle_epl_golden_vectors(gctx, oracle):
    """The committed E/P/L fixtures (oracle-generated, tests/golden/oracle_epl.npz)."""
    import gnsscorr
    z = np.load(os.path.join(G, "oracle_epl.npz"))
    e1b = np.load(os.path.join(G, "galileo_e1_codes.npz"))["e1b"]
    cfgs = [("gps_4m", 4000000, 4000, oracle.gps_l1_ca_code(1).astype(np.float32), [-0.5, 0, 0.5], 1.023e6, 1),
            ("gps_25m", 25000000, 25000, oracle.gps_l1_ca_code(9).astype(np.float32), [-0.5, 0, 0.5], 1.023e6, 1),
            ("bds_25m", 25000000, 25000, oracle.beidou_b1i_code(6).astype(np.float32), [-0.5, 0, 0.5], 2.046e6, 1),
            ("gal_25m", 25000000, 100000, oracle.galileo_e1_sinboc11(e1b[10]), [-1.2, -0.3, 0, 0.3, 1.2], 2.046e6, 2)]
    for name, fs, n, code, shifts, chip_rate, prompt in cfgs:
        sig, _ = synth_stream([code], fs, 3 * n, seed=int(z[name + "_seed"]), cn0_db_hz=(44.0, 44.0), chip_rate=chip_rate)
        shifts = np.array(shifts, np.float32)
        recs = [gnsscorr.epoch_params(int(r[0]), float(np.float32(r[1])), float(np.float32(r[2])), float(np.float32(r[3])),
            float(np.float32(r[4])), n) for r in z[name + "_scalars"]]
        got = _batch_one(gctx, sig, code, shifts, recs)
        for k, want in enumerate(z[name + "_out"]):
            assert rel_err(got[k], want, prompt) <= TOL, name


@pytest.mark.parametrize("n_taps", [1, 2, 4, 8])
def test_other_tap_counts(gctx, oracle, n_taps):
    import gnsscorr
    code = oracle.gps_l1_ca_code(21).astype(np.float32)
    fs, n = 4_000_000, 4000
    sig, truth = synth_stream([code], fs, 2 * n, seed=77, cn0_db_hz=(50.0, 50.0))
    shifts = np.linspace(-0.7, 0.7, n_taps).astype(np.float32) if n_taps > 1 else np.array([0.0], np.float32)
    p = open_loop_params(truth[0], fs, 1023, n, 1)[0]
    rec = [gnsscorr.epoch_params(1, float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n)]
    ref = oracle.multicorrelator(sig[1:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n)
    got = _batch_one(gctx, sig, code, shifts, rec)[0]
    assert np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 63, 511, 512, 513, 1025])
def test_tiny_and_ragged_windows(gctx, oracle, n):
    """Empty, single-sample and chunk-boundary windows, odd and even starts."""
    import gnsscorr
    code = oracle.gps_l1_ca_code(2).astype(np.float32)
    rng = np.random.Generator(np.random.PCG64(n + 1))
    sig = (rng.standard_normal(2048) + 1j * rng.standard_normal(2048)).astype(np.complex64)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    for off in (0, 1, 6, 7):
        rec = [gnsscorr.epoch_params(off, 0.3, 0.01, -3.25, 0.2557, n)]
        got = _batch_one(gctx, sig, code, shifts, rec)[0]
        ref = oracle.multicorrelator(sig[off:], code, shifts, np.float32(0.3), np.float32(0.01), np.float32(-3.25), np.float32(0.2557), n)
        scale = max(1.0, float(np.sqrt(max(n, 1))))
        assert np.max(np.abs(got - ref)) <= 1e-5 * scale, (n, off, got, ref)


def test_multi_period_window_uses_the_wrapping_path(gctx, oracle):
    """Index span > LDS window (several code periods per integration): generic modulo path."""
    import gnsscorr
    code = oracle.gps_l1_ca_code(30).astype(np.float32)
    fs, n = 4_000_000, 20000  # 5 code periods
    sig, truth = synth_stream([code], fs, n + 8, seed=9, cn0_db_hz=(45.0, 45.0))
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    p = open_loop_params(truth[0], fs, 1023, n, 1)[0]
    rec = [gnsscorr.epoch_params(0, float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n)]
    ref = oracle.multicorrelator(sig, code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n)
    got = _batch_one(gctx, sig, code, shifts, rec)[0]
    assert abs(ref[1]) > 0.5 * truth[0]["amp"] * n
    assert rel_err(got, ref, 1) <= TOL
    # negative code step (never produced by the tracking loop, but legal for the kernel)
    rec = [gnsscorr.epoch_params(0, 0.1, 0.002, 5.5, -0.2557, 4000)]
    ref = oracle.multicorrelator(sig, code, shifts, np.float32(0.1), np.float32(0.002), np.float32(5.5), np.float32(-0.2557), 4000)
    got = _batch_one(gctx, sig, code, shifts, rec)[0]
    assert np.max(np.abs(got - ref)) <= 1e-5 * np.sqrt(4000) + TOL * np.max(np.abs(ref))


def test_high_dynamics_resampler_and_rotator(gctx, oracle):
    """Tracking_XX.high_dyn = true: high-dynamics resampler (tap 0 resampled with the quadratic term,
    other taps sample-shifted copies) + high-dynamic rotator (carrier phase-rate term)."""
    import gnsscorr
    code = oracle.gps_l1_ca_code(14).astype(np.float32)
    fs, n = 25_000_000, 25000
    sig, truth = synth_stream([code], fs, 2 * n, seed=31, cn0_db_hz=(46.0, 46.0))
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    p = open_loop_params(truth[0], fs, 1023, n, 1)[0]
    carr_rate = np.float32(2 * np.pi * 50.0 / fs / fs)  # 50 Hz/s
    code_rate = np.float32(1e-12)
    ref = oracle.multicorrelator(sig, code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n,
        phase_rate_step=carr_rate, code_rate_step=code_rate, high_dyn=True)
    rec = [gnsscorr.epoch_params(0, float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n,
        carr_phase_rate_step_rad=float(carr_rate), code_phase_rate_step_chips=float(code_rate))]
    got = _batch_one(gctx, sig, code, shifts, rec, high_dyn=True)[0]
    assert abs(ref[1]) > 0.5 * truth[0]["amp"] * n
    assert rel_err(got, ref, 1) <= TOL, (got, ref)
    # drop-in class: default-constructed object has the high-dynamics flag SET (reference ctor, .cc:49)
    mc = gnsscorr.HipMulticorrelatorRealCodes(gctx)
    mc.init(2 * n, 3)
    out = np.zeros(3, np.complex64)
    mc.set_local_code_and_taps(1023, code, shifts)
    mc.set_input_output_vectors(out, sig)
    mc.Carrier_wipeoff_multicorrelator_resampler(float(p["rem_carr"]), float(p["phase_step"]), float(carr_rate),
        float(p["rem_code"]), float(p["code_step"]), float(code_rate), n)
    assert rel_err(out, ref, 1) <= TOL
    # 6-argument overload: plain rotator, resampler still follows the flag
    mc.Carrier_wipeoff_multicorrelator_resampler(float(p["rem_carr"]), float(p["phase_step"]),
        float(p["rem_code"]), float(p["code_step"]), float(code_rate), n)
    idx = oracle.resampler_indices(p["rem_code"], p["code_step"], shifts, 1023, n, rate=code_rate)
    wiped = sig[:n].astype(np.complex128) * np.exp(-1j * (float(p["rem_carr"]) + float(p["phase_step"]) * np.arange(n)))
    want = np.array([np.sum(wiped * code[idx[t]]) for t in range(3)])
    assert np.max(np.abs(out - want)) <= TOL * abs(want[1])
    mc.close()


def test_shifts_pointer_is_reread_between_calls(gctx, oracle):
    """The reference keeps the caller's shifts pointer and the tracking block edits it in place when it
    narrows the correlator spacing (dll_pll_veml_tracking.cc:1753-1764)."""
    import gnsscorr
    code = oracle.gps_l1_ca_code(8).astype(np.float32)
    fs, n = 4_000_000, 4000
    sig, truth = synth_stream([code], fs, 2 * n, seed=4, cn0_db_hz=(48.0, 48.0))
    p = open_loop_params(truth[0], fs, 1023, n, 1)[0]
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    mc = gnsscorr.HipMulticorrelatorRealCodes(gctx)
    mc.set_high_dynamics_resampler(False)
    mc.init(2 * n, 3)
    out = np.zeros(3, np.complex64)
    mc.set_local_code_and_taps(1023, code, shifts)
    mc.set_input_output_vectors(out, sig)
    args = (float(p["rem_carr"]), float(p["phase_step"]), 0.0, float(p["rem_code"]), float(p["code_step"]), 0.0, n)
    mc.Carrier_wipeoff_multicorrelator_resampler(*args)
    wide = out.copy()
    shifts[0], shifts[2] = -0.15, 0.15  # edited in place, no setter call
    mc.Carrier_wipeoff_multicorrelator_resampler(*args)
    ref = oracle.multicorrelator(sig, code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n)
    assert rel_err(out, ref, 1) <= TOL
    assert abs(out[0]) > abs(wide[0])  # narrower spacing: early/late move up the correlation triangle
    mc.close()


def test_full_size_properties_cfg2(gctx, oracle):
    """BASELINE configs[1] at full width (32 channels, 25 Msps, 64 epochs, shared stream): properties
    that need no oracle pass -- linearity in the input, tap symmetry on a noiseless signal, and a
    spot check of 8 random channel-epochs against the oracle."""
    import gnsscorr
    import torch
    fs, n, n_ch, n_ep = 25_000_000, 25000, 32, 64
    codes = [oracle.gps_l1_ca_code(prn).astype(np.float32) for prn in range(1, n_ch + 1)]
    sig, truth = synth_stream(codes, fs, n * n_ep + 8, seed=1002, cn0_db_hz=(40.0, 48.0))
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    d1 = torch.from_numpy(sig.view(np.float32)).cuda()
    d2 = (d1 * 2.0).contiguous()
    params = []
    for ch in range(n_ch):
        params.append([gnsscorr.epoch_params(p["sample_offset"], float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]),
            float(p["code_step"]), n) for p in open_loop_params(truth[ch], fs, 1023, n, n_ep)])
    arr = gnsscorr.epoch_params_array(params)
    outs = []
    for d in (d1, d2):
        b = gnsscorr.TrackingBatch(gctx, n_ch, 3, 1023)
        for ch in range(n_ch):
            b.set_code(ch, codes[ch], shifts)
            b.set_input_dev(ch, d.data_ptr(), sig.size)
        outs.append(b.run(n_ep, arr))
        b.close()
    assert np.array_equal(outs[1], outs[0] * 2)  # scaling by 2 is exact in binary floating point
    pm = np.abs(outs[0][:, :, 1])
    for ch in range(n_ch):
        assert pm[ch].mean() > 0.6 * truth[ch]["amp"] * n
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(8):
        ch, k = int(rng.integers(n_ch)), int(rng.integers(n_ep))
        p = open_loop_params(truth[ch], fs, 1023, n, n_ep)[k]
        ref = oracle.multicorrelator(sig[p["sample_offset"]:], codes[ch], shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n)
        assert rel_err(outs[0][ch, k], ref, 1) <= TOL


@pytest.mark.parametrize("fmt_name,np_type,scale", [("GC_IQ_I16", np.int16, 600.0), ("GC_IQ_I8", np.int8, 24.0)])
def test_integer_iq_formats_equal_float_path_on_converted_samples(gctx, oracle, fmt_name, np_type, scale):
    """cshort / cbyte input (what front-ends deliver; the reference converts them to gr_complex with a plain cast
    before its float correlators, pcps_acquisition.cc:676-679 / data_type_adapter): the engine converts on load,
    so results must equal the float path fed with the converted samples."""
    import gnsscorr
    import torch
    fs, n, n_ep = 25_000_000, 25000, 3
    code = oracle.gps_l1_ca_code(12).astype(np.float32)
    sig, truth = synth_stream([code], fs, n * n_ep + 16, seed=21, cn0_db_hz=(46.0, 46.0))
    q = np.clip(np.round(sig.view(np.float32) * scale), np.iinfo(np_type).min, np.iinfo(np_type).max).astype(np_type)  # interleaved re, im
    sig_f = q.astype(np.float32).view(np.complex64)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    d_q = torch.from_numpy(q).cuda()
    b = gnsscorr.TrackingBatch(gctx, 1, 3, 1023)
    b.set_input_format(getattr(gnsscorr, fmt_name))
    b.set_code(0, code, shifts)
    recs, refs = [], []
    for k, p in enumerate(open_loop_params(truth[0], fs, 1023, n, n_ep)):
        off = p["sample_offset"] + k * 3  # odd and even starts
        recs.append(gnsscorr.epoch_params(off, float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n))
        refs.append(oracle.multicorrelator(sig_f[off:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n))
    b.set_input_dev(0, d_q.data_ptr(), sig_f.size)
    got = b.run(n_ep, gnsscorr.epoch_params_array(recs))[0]
    b.close()
    for k in range(n_ep):
        assert abs(refs[k][1]) > 0.4 * truth[0]["amp"] * scale * n
        assert rel_err(got[k], refs[k], 1) <= TOL


@pytest.mark.parametrize("case", ["max_code_length", "huge_code_phase", "zero_code_step", "far_taps", "fast_carrier"])
def test_parameter_extremes(gctx, oracle, case):
    """Parameters at the edges of what the kernel accepts; every one must still walk the reference's chip indices
    (the oracle's are pinned bit for bit) and sum to the oracle's values."""
    import gnsscorr
    rng = np.random.Generator(np.random.PCG64(["max_code_length", "huge_code_phase", "zero_code_step", "far_taps", "fast_carrier"].index(case) + 70))
    n = 6000
    sig = (rng.standard_normal(n + 16) + 1j * rng.standard_normal(n + 16)).astype(np.complex64)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    L, rem, step, carr = 1023, np.float32(-7.75), np.float32(0.2557), np.float32(0.01)
    if case == "max_code_length":
        L = 15936  # largest table the 64 KB LDS window takes (kMaxLdsTableFloats - 64)
        step = np.float32(2.5)
    elif case == "huge_code_phase":
        rem = np.float32(-1.0e6)  # hundreds of code periods away: float32 has 1/16-chip resolution there, as in the reference
    elif case == "zero_code_step":
        step = np.float32(0.0)
    elif case == "far_taps":
        shifts = np.array([-3000.25, 0.0, 2999.5], np.float32)  # taps several code periods apart: no LDS window, wrapping path
    elif case == "fast_carrier":
        carr = np.float32(3.0)  # close to pi rad per sample
    code = np.sign(rng.standard_normal(L)).astype(np.float32)
    rec = [gnsscorr.epoch_params(3, 0.7, float(carr), float(rem), float(step), n)]
    ref = oracle.multicorrelator(sig[3:], code, shifts, np.float32(0.7), carr, rem, step, n)
    got = _batch_one(gctx, sig, code, shifts, rec)[0]
    assert np.max(np.abs(got - ref)) <= 2e-5 * np.sqrt(n) + TOL * np.max(np.abs(ref)), (case, got, ref)


def _run_group(gctx, d_sig, n_sig, codes, shifts, truths, fs, L, n, n_ep):
    """One batched launch: channel ch tracks codes[ch] on the shared device stream with open-loop records from its truth."""
    import gnsscorr
    params = [[gnsscorr.epoch_params(p["sample_offset"], float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n)
        for p in open_loop_params(truths[ch], fs, L, n, n_ep)] for ch in range(len(codes))]
    b = gnsscorr.TrackingBatch(gctx, len(codes), len(shifts), L)
    for ch, code in enumerate(codes):
        b.set_code(ch, code, shifts)
        b.set_input_dev(ch, d_sig.data_ptr(), n_sig)
    out = b.run(n_ep, gnsscorr.epoch_params_array(params))
    b.close()
    return out


def test_full_size_properties_cfg3_galileo_32_channels(gctx, oracle):
    """BASELINE configs[2] at full width: Galileo E1 sinBOC(1,1) replicas (L = 8184, 2 samples per chip), 32 channels, 25 Msps,
    5 taps VE/E/P/L/VL at -+0.6 / -+0.15 chips, N = 100000 (4 ms coherent), 6 periods, all channels on one RF stream
    (gnss_flowgraph.cc:496-499).  Exact linearity in the input, every prompt sees its signal, and 8 random channel-periods
    against the oracle at 1e-4 of the prompt."""
    import torch
    e1b = np.load(os.path.join(G, "galileo_e1_codes.npz"))["e1b"]
    fs, n, n_ch, n_ep = 25_000_000, 100000, 32, 6
    codes = [oracle.galileo_e1_sinboc11(e1b[prn]) for prn in range(n_ch)]
    sig, truth = synth_stream(codes, fs, n * n_ep + 8, seed=1003, cn0_db_hz=(40.0, 48.0), chip_rate=2 * 1.023e6)
    shifts = np.array([-1.2, -0.3, 0.0, 0.3, 1.2], np.float32)
    d1 = torch.from_numpy(sig.view(np.float32)).cuda()
    d2 = (d1 * 2.0).contiguous()
    outs = [_run_group(gctx, d, sig.size, codes, shifts, truth, fs, 8184, n, n_ep) for d in (d1, d2)]
    assert np.array_equal(outs[1], outs[0] * 2)
    pm = np.abs(outs[0][:, :, 2])
    for ch in range(n_ch):
        assert pm[ch].mean() > 0.6 * truth[ch]["amp"] * n
        # E and L on the main lobe of the BOC autocorrelation, VE / VL on the far side of its zero crossing
        assert np.abs(outs[0][ch, :, 1]).mean() > np.abs(outs[0][ch, :, 0]).mean()
    rng = np.random.Generator(np.random.PCG64(6))
    for _ in range(8):
        ch, k = int(rng.integers(n_ch)), int(rng.integers(n_ep))
        p = open_loop_params(truth[ch], fs, 8184, n, n_ep)[k]
        ref = oracle.multicorrelator(sig[p["sample_offset"]:], codes[ch], shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n)
        assert rel_err(outs[0][ch, k], ref, 2) <= TOL


def test_cfg5_per_gpu_mix_gps_galileo_beidou(gctx, oracle):
    """One GPU's share of BASELINE configs[4] (256 channels round-robin over 8 GPUs = 32 per GPU): 16 GPS L1 C/A (3 taps,
    L = 1023) + 8 Galileo E1 (5 taps, L = 8184, 4 ms) on the L1 stream and 8 BeiDou B1I (3 taps, L = 2046) on the B1 stream
    (another carrier, Beidou_B1I.h:51), three launches like bench.py's hybrid step.  Two channel-periods per signal are checked
    against the oracle; all prompts see their signals."""
    import torch
    e1b = np.load(os.path.join(G, "galileo_e1_codes.npz"))["e1b"]
    fs, n = 25_000_000, 25000
    n_ep = 8
    gps = [oracle.gps_l1_ca_code(prn).astype(np.float32) for prn in range(1, 17)]
    gal = [oracle.galileo_e1_sinboc11(e1b[prn]) for prn in range(8)]
    bds = [oracle.beidou_b1i_code(prn).astype(np.float32) for prn in range(6, 14)]
    # L1 stream: GPS C/A at 1.023 Mcps and Galileo sinBOC samples at 2.046 M code samples per second, one noise floor
    s_gps, t_gps = synth_stream(gps, fs, n * n_ep + 8, seed=1005, cn0_db_hz=(40.0, 48.0), noise=False)
    s_gal, t_gal = synth_stream(gal, fs, n * n_ep + 8, seed=2005, cn0_db_hz=(40.0, 48.0), chip_rate=2 * 1.023e6)
    l1 = (s_gps + s_gal).astype(np.complex64)
    b1, t_bds = synth_stream(bds, fs, n * n_ep + 8, seed=3005, cn0_db_hz=(40.0, 48.0), chip_rate=2.046e6, carrier_freq=1561.098e6)
    d_l1 = torch.from_numpy(l1.view(np.float32)).cuda()
    d_b1 = torch.from_numpy(b1.view(np.float32)).cuda()
    s3 = np.array([-0.5, 0.0, 0.5], np.float32)
    s5 = np.array([-1.2, -0.3, 0.0, 0.3, 1.2], np.float32)
    groups = [("gps", d_l1, l1, gps, s3, t_gps, 1023, n, n_ep, 1), ("galileo", d_l1, l1, gal, s5, t_gal, 8184, 4 * n, n_ep // 4, 2),
        ("beidou", d_b1, b1, bds, s3, t_bds, 2046, n, n_ep, 1)]
    rng = np.random.Generator(np.random.PCG64(7))
    for name, d, sig, codes, shifts, truths, L, n_len, k_ep, pi in groups:
        out = _run_group(gctx, d, sig.size, codes, shifts, truths, fs, L, n_len, k_ep)
        for ch in range(len(codes)):
            assert np.abs(out[ch, :, pi]).mean() > 0.6 * truths[ch]["amp"] * n_len, (name, ch)
        for _ in range(2):
            ch, k = int(rng.integers(len(codes))), int(rng.integers(k_ep))
            p = open_loop_params(truths[ch], fs, L, n_len, k_ep)[k]
            ref = oracle.multicorrelator(sig[p["sample_offset"]:], codes[ch], shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n_len)
            assert rel_err(out[ch, k], ref, pi) <= TOL, (name, ch, k)
