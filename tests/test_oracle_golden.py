"""CPU tests (no GPU): the oracle against the committed golden vectors.

ref_*.npz come from the REFERENCE's own sources compiled into oracle/_ref
(tests/golden/make_golden.py); kat_* are the captures and gates of the
reference's own acquisition tests.  These tests are what "pins" the oracle.
"""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_resampler_chip_indices_match_compiled_reference(oracle):
    z = np.load(os.path.join(G, "ref_resampler.npz"))
    n = int(z["n_cases"])
    assert n >= 20
    for i in range(n):
        L, N = (int(v) for v in z["c%d_params" % i])
        rem, step, rate = (np.float32(v) for v in z["c%d_f" % i])
        shifts = z["c%d_shifts" % i]
        got = oracle.resampler_indices(rem, step, shifts, L, N)
        assert np.array_equal(got, z["c%d_idx" % i].astype(np.int32)), "case %d" % i
        if not np.isnan(rate):
            got = oracle.resampler_indices(rem, step, shifts, L, N, rate=rate)
            assert np.array_equal(got, z["c%d_idx_hd" % i].astype(np.int32)), "hd case %d" % i


def test_complex_chip_resampler_walks_the_same_indices(oracle):
    """make_golden.py asserted that the reference's volk_gnsssdr_32fc_xn_resampler_32fc_xn_generic
    (Cpu_Multicorrelator) gathers exactly the indices stored for the real-code resampler."""
    z = np.load(os.path.join(G, "ref_resampler.npz"))
    assert int(z["complex_chip_resampler_checked"]) == 1 and int(z["int16_chip_resampler_checked"]) == 1
    for i in range(int(z["n_cases"])):
        L, N = (int(v) for v in z["c%d_params" % i])
        rem, step, _ = (np.float32(v) for v in z["c%d_f" % i])
        ramp = np.arange(L, dtype=np.float32)
        got = oracle.resampler_cc((ramp + 1j * (L - ramp)).astype(np.complex64), rem, step, z["c%d_shifts" % i], N)
        idx = z["c%d_idx" % i].astype(np.int32)
        assert np.array_equal(got.real.astype(np.int32), idx) and np.array_equal(got.imag.astype(np.int32), L - idx), "case %d" % i


def test_prn_generators_match_compiled_reference(oracle):
    z = np.load(os.path.join(G, "ref_codes.npz"))
    for k, prn in enumerate(z["gps_prn"]):
        assert np.array_equal(oracle.gps_l1_ca_code(int(prn)), z["gps_chips"][k])
    assert np.array_equal(oracle.gps_l1_ca_code(5, 7), z["gps_chips_shift7"])
    for k, prn in enumerate(z["bds_prn"]):
        assert np.array_equal(oracle.beidou_b1i_code(int(prn)), z["bds_chips"][k])
    assert np.array_equal(oracle.glonass_l1_ca_code(), z["glo_chips"])
    assert np.array_equal(oracle.glonass_l1_ca_code(100), z["glo_chips_shift100"])
    for fs in (4000000, 25000000, 2048000):
        assert np.array_equal(oracle.glonass_l1_ca_code_sampled(fs).real, z["glo_sampled_fs%d" % fs])
        assert np.array_equal(oracle.gps_l1_ca_code_sampled(1, fs).real, z["gps_sampled_fs%d_prn1" % fs])
        assert np.array_equal(oracle.gps_l1_ca_code_sampled(19, fs).real, z["gps_sampled_fs%d_prn19" % fs])
        assert np.array_equal(oracle.beidou_b1i_code_sampled(6, fs).real, z["bds_sampled_fs%d_prn6" % fs])


def test_gps_ca_code_properties(oracle):
    """IS-GPS-200: first 10 chips of PRN 1 are 1100100000 (octal 1440); balanced Gold codes."""
    c = oracle.gps_l1_ca_code(1)
    assert "".join("1" if v == 1 else "0" for v in c[:10]) == "1100100000"
    for prn in range(1, 33):
        c = oracle.gps_l1_ca_code(prn).astype(np.int64)
        assert abs(int(c.sum())) == 1
        ac = np.array([np.dot(c, np.roll(c, k)) for k in (1, 7, 100, 511)])
        assert set(ac.tolist()) <= {-1, 63, -65}


def test_sincos_and_index_max_match_compiled_reference(oracle):
    z = np.load(os.path.join(G, "ref_sincos.npz"))
    for i, inc in enumerate(z["phase_inc"]):
        got = oracle.sincos(float(inc), 25000)
        assert np.array_equal(got.view(np.float32), z["out%d" % i].view(np.float32))
    assert oracle.index_max(z["imax_in"]) == int(z["imax_out"]) == 17


def test_galileo_e1_codes_and_sinboc(oracle):
    z = np.load(os.path.join(G, "galileo_e1_codes.npz"))
    e1b, e1c = z["e1b"], z["e1c"]
    assert e1b.shape == e1c.shape == (50, 4092)
    assert set(np.unique(e1b).tolist()) == {-1, 1}
    # Galileo OS SIS ICD Annex C: E1-B PRN 1 starts with hex F5D710130573..., E1-C PRN 1 with B39340...
    def first_hex(row, n):
        bits = (1 - row[:4 * n]) // 2
        return "".join("%X" % int("".join(str(b) for b in bits[4 * i:4 * i + 4]), 2) for i in range(n))
    assert first_hex(e1b[0], 12) == "F5D710130573"
    s = oracle.galileo_e1_sinboc11(e1b[3])
    assert np.array_equal(s[0::2], e1b[3].astype(np.float32)) and np.array_equal(s[1::2], -e1b[3].astype(np.float32))
    # sampled replica: 4 Msps -> 16000 samples per 4 ms code, +-1 valued without CBOC
    c = oracle.galileo_e1_code_sampled(e1b[0], 4000000, cboc=False)
    assert c.size == 16000 and set(np.unique(c).tolist()) == {-1.0, 1.0}
    c = oracle.galileo_e1_code_sampled(e1b[0], 4000000, cboc=True)
    assert c.size == 16000 and len(np.unique(np.abs(c))) == 2


def test_fft_matches_numpy(oracle):
    rng = np.random.Generator(np.random.PCG64(11))
    for n in (1, 2, 5, 8, 1000, 4000, 2046, 25000):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        f = np.fft.fft(x)
        assert np.max(np.abs(oracle.fft(x) - f)) <= 1e-12 * max(1.0, np.max(np.abs(f)))
        assert np.max(np.abs(oracle.fft(x, inverse=True) - np.fft.ifft(x) * n)) <= 1e-12 * max(1.0, np.max(np.abs(f)))


def _kat(name):
    k = json.load(open(os.path.join(G, "kat_expected.json")))[name]
    x = np.fromfile(os.path.join(G, k["file"]), np.complex64)
    return k, x


def test_pcps_gps_l1_known_answer(oracle):
    """GpsL1CaPcpsAcquisitionTest.ValidationOfResults gates (reference test :281-356)."""
    k, x = _kat("gps_l1_ca")
    fs = k["fs"]
    p = oracle.pcps(fs_in=fs, sampled_ms=1, ms_per_code=1, samples_per_ms=np.float32(fs) * np.float32(0.001),
        samples_per_code=4000.0, samples_per_chip=4, doppler_max=k["doppler_max"], doppler_step=k["doppler_step"])
    assert (p.fft_size, p.num_doppler_bins) == (4000, 100)
    p.set_local_code(oracle.gps_l1_ca_code_sampled(k["prn"], fs))
    r = p.core(x)
    g = k["reference_test"]
    assert abs(r.acq_delay_samples - g["expected_delay_samples"]) * 1023 / 4000 < g["max_delay_error_chips"]
    assert abs(r.acq_doppler_hz - g["expected_doppler_hz"]) <= g["max_doppler_error_hz"]
    assert r.test_statistics > k["threshold"]
    assert (r.indext, r.doppler) == (k["oracle"]["indext"], k["oracle"]["doppler"])
    assert r.test_statistics == pytest.approx(k["oracle"]["test_statistics"], rel=1e-6)


def test_pcps_galileo_e1_known_answer(oracle):
    """GalileoE1PcpsAmbiguousAcquisitionTest.ValidationOfResults gates (reference test :293-358)."""
    k, x = _kat("galileo_e1")
    fs = k["fs"]
    e1b = np.load(os.path.join(G, "galileo_e1_codes.npz"))["e1b"]
    p = oracle.pcps(fs_in=fs, sampled_ms=4, ms_per_code=4, samples_per_ms=np.float32(fs) * np.float32(0.001),
        samples_per_code=16000.0, samples_per_chip=4, doppler_max=k["doppler_max"], doppler_step=k["doppler_step"])
    assert (p.fft_size, p.num_doppler_bins) == (16000, 80)
    p.set_local_code(oracle.galileo_e1_code_sampled(e1b[k["prn"] - 1], fs, cboc=False).astype(np.complex64))
    r = p.core(x)
    g = k["reference_test"]
    assert abs(r.acq_delay_samples - g["expected_delay_samples"]) * 1023 / 4000 < g["max_delay_error_chips"]
    assert abs(r.acq_doppler_hz - g["expected_doppler_hz"]) <= g["max_doppler_error_hz"]
    assert r.test_statistics > k["threshold"]


def test_pcps_glonass_l1_real_capture(oracle):
    """The NT1065 GLONASS L1 capture of the reference's GLONASS tracking tests: a PCPS search with the GLONASS C/A
    replica on frequency channel 0 lands on the acquisition hand-over those tests hard-code (delay 1343 samples,
    Doppler -2750 Hz for PRN 11; glonass_l1_ca_dll_pll_tracking_test.cc:134-167).  Satellites on other FDMA channels
    need d_old_freq = DFRQ1_GLO * k (pcps_acquisition.cc:276-293)."""
    k, x = _kat("glonass_l1_ca")
    fs = k["fs"]
    conf = dict(fs_in=fs, sampled_ms=1, ms_per_code=1, samples_per_ms=np.float32(fs) * np.float32(0.001), samples_per_code=6625.0,
        samples_per_chip=13, doppler_max=k["doppler_max"], doppler_step=k["doppler_step"])
    code = oracle.glonass_l1_ca_code_sampled(fs)
    assert code.size == 6625 and x.size >= 4 * 6625 - 1
    g = k["reference_test"]
    for ms in range(3):  # the satellite is there in every millisecond of the capture
        p = oracle.pcps(**conf)
        p.set_local_code(code)
        r = p.core(x[ms * 6625:])
        assert abs(r.acq_delay_samples - g["expected_delay_samples"]) * 511 / 6625 < g["max_delay_error_chips"]
        assert abs(r.acq_doppler_hz - g["expected_doppler_hz"]) <= g["max_doppler_error_hz"]
    for kc, want in k["oracle_by_frequency_channel"].items():
        p = oracle.pcps(**conf)
        p.set_local_code(code)
        p.set_frequency_offset(k["dfrq1_glo_hz"] * int(kc))
        r = p.core(x)
        assert (r.indext, r.doppler) == (want["indext"], want["doppler"])
        assert r.test_statistics == pytest.approx(want["test_statistics"], rel=1e-6)
    # without the FDMA offset the channel -3 satellite is not found where it is
    p = oracle.pcps(**conf)
    p.set_local_code(code)
    assert p.core(x).indext != k["oracle_by_frequency_channel"]["-3"]["indext"]


def test_pcps_galileo_e1_real_capture(oracle):
    """Real Galileo E1 signal (GSoC 2012 roof capture): the reference ships a MATLAB analysis of its acquisition grid beside
    the file -- PRN 11 at 13873 samples / 9500 Hz and PRN 12 at 10583 samples / 7250 Hz.  Delays must match exactly and the
    Doppler magnitudes exactly (that 2012 listing uses the opposite Doppler sign of today's pcps_acquisition)."""
    k, x = _kat("galileo_e1_real_capture")
    fs = k["fs"]
    z = np.load(os.path.join(G, "galileo_e1_codes.npz"))
    stats = {}
    for prn in (11, 12, 19, 20):
        p = oracle.pcps(fs_in=fs, sampled_ms=4, ms_per_code=4, samples_per_ms=np.float32(fs) * np.float32(0.001), samples_per_code=16000.0,
            samples_per_chip=4, doppler_max=k["doppler_max"], doppler_step=k["doppler_step"])
        assert (p.fft_size, p.num_doppler_bins) == (16000, 160)
        p.set_local_code(oracle.galileo_e1_code_sampled(z["e1b"][prn - 1], fs, cboc=False).astype(np.complex64))
        r = p.core(x)
        stats[prn] = r.test_statistics
        want = k["oracle_by_prn"][str(prn)]
        assert (r.indext, r.doppler) == (want["indext"], want["doppler"])
        if str(prn) in k["reference_analysis"]:
            a = k["reference_analysis"][str(prn)]
            assert r.indext == a["delay_samples"] and abs(r.doppler) == a["abs_doppler_hz"]
    assert min(stats[11], stats[12]) > 2.0 * max(stats[19], stats[20])


def test_pcps_grid_values_against_the_reference_matlab_analysis(oracle):
    """VALUE-level pin of the search grid from numbers the reference itself ships: next to the GSoC 2012 capture lies the output
    of src/utils/matlab/plot_acq_grid_gsoc.m on the acquisition dump of that very file -- peak magnitude, noise floor and their
    ratio for Galileo PRN 11 (23.0285 / 1.8919 / 10.8538 dB) and PRN 12 (16.5534 / 1.9020 / 9.3968 dB), E1C replica, 160 bins of
    125 Hz, 4 ms.  The oracle's grid, put through the same statistics (tests/helpers.py::gsoc_grid_statistics), must give the same
    gain within 0.02 dB and, with the |IFFT(FFT.FFT*)| gain of N^2 removed, the same peak and floor within 0.3 %: the 2012
    receiver is not today's pcps_acquisition bit for bit (Doppler sign, replica digitisation), so this pins the grid to ~2e-3,
    not to the 1e-4 the oracle-vs-GPU comparison holds; measured: 10.8541 dB and 9.3888 dB, peaks 0.02 % and 0.2 % off."""
    from helpers import gsoc_grid_statistics
    k, x = _kat("galileo_e1_real_capture")
    fs = k["fs"]
    z = np.load(os.path.join(G, "galileo_e1_codes.npz"))
    for prn in (11, 12):
        a = k["reference_analysis"][str(prn)]
        p = oracle.pcps(fs_in=fs, sampled_ms=4, ms_per_code=4, samples_per_ms=np.float32(fs) * np.float32(0.001), samples_per_code=16000.0,
            samples_per_chip=4, doppler_max=k["doppler_max"], doppler_step=k["doppler_step"])
        p.set_local_code(oracle.galileo_e1_code_sampled(z["e1c"][prn - 1], fs, cboc=False, is_e1c=True).astype(np.complex64))
        p.core(x)
        peak, floor_, gain, row, col = gsoc_grid_statistics(p.grid(), fs, k["doppler_step"])
        assert col == a["delay_samples"] and abs(-k["doppler_max"] + k["doppler_step"] * row) == a["abs_doppler_hz"]
        assert abs(gain - a["gain_db"]) < 0.02, (prn, gain)
        n2 = float(p.fft_size) ** 2
        assert abs(peak / n2 / a["maximum_correlation_peak"] - 1.0) < 3e-3, (prn, peak / n2)
        assert abs(floor_ / n2 / a["noise_floor"] - 1.0) < 3e-3, (prn, floor_ / n2)


def test_pcps_second_peak_and_dwells(oracle):
    """Two non-coherent dwells switch the statistic to first/second peak (pcps_acquisition.cc:152-159)
    and accumulate |.|^2 (:737-738)."""
    k, x = _kat("gps_l1_ca")
    fs = k["fs"]
    p = oracle.pcps(fs_in=fs, sampled_ms=1, ms_per_code=1, samples_per_ms=np.float32(fs) * np.float32(0.001),
        samples_per_code=4000.0, samples_per_chip=4, doppler_max=5000, doppler_step=250, max_dwells=2)
    assert not p.use_cfar
    p.set_local_code(oracle.gps_l1_ca_code_sampled(1, fs))
    r1 = p.core(x[:4000])
    g1 = p.grid()
    r2 = p.core(x[4000:])
    g2 = p.grid()
    assert np.all(g2 >= g1) and r2.mag > r1.mag
    assert r1.indext == r2.indext == 524
    assert r2.second_peak_fixed > 0 and r2.mag / r2.second_peak_fixed > 3.0
    assert r2.test_statistics == pytest.approx(r2.mag / r2.second_peak)


def test_multicorrelator_consistent_with_acquisition_cell(oracle):
    """Cross-check of the two restatements: |P|^2 of the tracking correlator at the acquisition
    peak (delay 524, Doppler bin 1700 Hz) equals the PCPS grid cell there (same correlation)."""
    k, x = _kat("gps_l1_ca")
    fs, n = k["fs"], 4000
    p = oracle.pcps(fs_in=fs, sampled_ms=1, ms_per_code=1, samples_per_ms=np.float32(fs) * np.float32(0.001),
        samples_per_code=4000.0, samples_per_chip=4, doppler_max=5000, doppler_step=100)
    sampled = oracle.gps_l1_ca_code_sampled(1, fs)
    p.set_local_code(sampled)
    r = p.core(x)
    cell = p.grid()[r.doppler_index, r.indext]
    # correlate the same block with the sampled code delayed by indext and the bin's wipe-off frequency
    w = p.wipeoffs()[r.doppler_index]
    ref = np.sum(x[:n].astype(np.complex128) * w[:n] * np.roll(sampled.real, r.indext))
    # the unnormalised IFFT carries a factor N on the correlation, N^2 on its squared magnitude
    assert abs(cell - (n * abs(ref)) ** 2) <= 2e-4 * cell
    # and through the tracking restatement: a 4000-chip table (the sampled code), step 1 chip/sample
    code = sampled.real.astype(np.float32)
    out = oracle.multicorrelator(x, code, np.array([0.0], np.float32), np.float32(0.0),
        np.float32(2 * np.pi * r.doppler / fs), np.float32(r.indext), np.float32(1.0), n)
    assert abs((n * abs(out[0])) ** 2 - cell) <= 2e-3 * cell


def test_oracle_epl_regression(oracle):
    """The oracle reproduces its own committed E/P/L vectors (guards the restatement against edits)."""
    import sys
    from helpers import synth_stream
    z = np.load(os.path.join(G, "oracle_epl.npz"))
    e1b = np.load(os.path.join(G, "galileo_e1_codes.npz"))["e1b"]
    cfgs = [("gps_4m", 4000000, 4000, oracle.gps_l1_ca_code(1).astype(np.float32), [-0.5, 0, 0.5], 1.023e6),
            ("gps_25m", 25000000, 25000, oracle.gps_l1_ca_code(9).astype(np.float32), [-0.5, 0, 0.5], 1.023e6)]
    for name, fs, n, code, shifts, chip_rate in cfgs:
        sig, _ = synth_stream([code], fs, 3 * n, seed=int(z[name + "_seed"]), cn0_db_hz=(44.0, 44.0), chip_rate=chip_rate)
        for row, want in zip(z[name + "_scalars"], z[name + "_out"]):
            off = int(row[0])
            got = oracle.multicorrelator(sig[off:], code, np.array(shifts, np.float32), np.float32(row[1]), np.float32(row[2]),
                np.float32(row[3]), np.float32(row[4]), n)
            assert np.array_equal(got.view(np.float32), want.view(np.float32))


def test_survey_probe_of_the_genuine_reference(oracle):
    """SURVEY.md section 8c records a run of the REFERENCE's own Cpu_Multicorrelator_Real_Codes (compiled
    during the survey): a noiseless PRN-1 input at 4 Msps, 3 taps at -0.5/0/+0.5 chips, N = 4000 gave
    P = (4000,0), E = (2004,0), L = (1992,0).  The oracle's rotator + accumulate restatement reproduces it."""
    code = oracle.gps_l1_ca_code(1).astype(np.float32)
    step = np.float32(1023.0 / 4000)
    idx = oracle.resampler_indices(np.float32(0), step, np.array([0], np.float32), 1023, 4000)[0]
    sig = code[idx].astype(np.complex64)
    out = oracle.multicorrelator(sig, code, np.array([-0.5, 0, 0.5], np.float32), np.float32(0), np.float32(0), np.float32(0), step, 4000)
    assert out.tolist() == [2004 + 0j, 4000 + 0j, 1992 + 0j]
