"""GPU parity of the PCPS acquisition engine (HIP, through the C ABI) against the CPU
oracle and against the reference's own known-answer tests.

Bars: Doppler row and code-phase index EXACT; peak magnitude, input power and test statistic
within 1e-4 relative (north_star tolerance); the whole search grid within 1e-4 of the peak
(FFTW's own float32 rounding is not reproducible, so the grid is compared with the oracle's
float64-FFT restatement -- SURVEY.md section 8c)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def _kat(name):
    k = json.load(open(os.path.join(G, "kat_expected.json")))[name]
    return k, np.fromfile(os.path.join(G, k["file"]), np.complex64)


def _conf(fs, ms, ms_per_code, spcode, dmax, dstep, **kw):
    return dict(fs_in=fs, sampled_ms=ms, ms_per_code=ms_per_code, samples_per_ms=np.float32(fs) * np.float32(0.001),
        samples_per_code=spcode, samples_per_chip=int(np.ceil(np.float32(9.7752e-07) * np.float32(fs))),
        doppler_max=dmax, doppler_step=dstep, **kw)


def _check(r, q, cfar=True):
    assert (r.indext, r.doppler_hz, r.doppler_index) == (q.indext, q.doppler, q.doppler_index)
    assert r.mag == pytest.approx(q.mag, rel=TOL)
    assert r.test_statistics == pytest.approx(q.test_statistics, rel=2 * TOL)
    if cfar:
        assert r.input_power == pytest.approx(q.input_power, rel=TOL)
    assert r.acq_delay_samples == q.acq_delay_samples and r.acq_doppler_hz == q.acq_doppler_hz


def test_gps_l1_known_answer_and_grid(gctx, oracle):
    import gnsscorr
    k, x = _kat("gps_l1_ca")
    c = _conf(k["fs"], 1, 1, 4000.0, k["doppler_max"], k["doppler_step"])
    code = oracle.gps_l1_ca_code_sampled(k["prn"], k["fs"])
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    assert (acq.fft_size, acq.consumed_samples, acq.num_doppler_bins) == (4000, 4000, 100)
    acq.set_local_code(0, code)
    r = acq.dwell(x)[0]
    g = k["reference_test"]  # gates of GpsL1CaPcpsAcquisitionTest.ValidationOfResults
    assert abs(r.acq_delay_samples - g["expected_delay_samples"]) * 1023 / 4000 < g["max_delay_error_chips"]
    assert abs(r.acq_doppler_hz - g["expected_doppler_hz"]) <= g["max_doppler_error_hz"]
    assert r.test_statistics > k["threshold"]
    p = oracle.pcps(**c)
    p.set_local_code(code)
    q = p.core(x)
    _check(r, q)
    grid, ref = acq.grid(0), p.grid()
    assert np.max(np.abs(grid - ref)) <= TOL * ref.max()
    acq.close()


def test_galileo_e1_known_answer_4ms(gctx, oracle):
    import gnsscorr
    k, x = _kat("galileo_e1")
    e1b = np.load(os.path.join(G, "galileo_e1_codes.npz"))["e1b"]
    c = _conf(k["fs"], 4, 4, 16000.0, k["doppler_max"], k["doppler_step"])
    for cboc in (False, True):
        code = oracle.galileo_e1_code_sampled(e1b[0], k["fs"], cboc=cboc).astype(np.complex64)
        acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
        assert (acq.fft_size, acq.num_doppler_bins) == (16000, 80)
        acq.set_local_code(0, code)
        r = acq.dwell(x)[0]
        g = k["reference_test"]
        assert abs(r.acq_delay_samples - g["expected_delay_samples"]) * 1023 / 4000 < g["max_delay_error_chips"]
        assert abs(r.acq_doppler_hz - g["expected_doppler_hz"]) <= g["max_doppler_error_hz"]
        p = oracle.pcps(**c)
        p.set_local_code(code)
        _check(r, p.core(x))
        acq.close()


def test_two_dwells_noncoherent_first_vs_second_peak(gctx, oracle):
    """max_dwells=2: CFAR off (pcps_acquisition.cc:152-159), grids accumulate, statistic =
    first/second peak with the reference's N-byte scratch copy (:647)."""
    import gnsscorr
    k, x = _kat("gps_l1_ca")
    c = _conf(k["fs"], 1, 1, 4000.0, 5000, 250, max_dwells=2)
    code = oracle.gps_l1_ca_code_sampled(1, k["fs"])
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    acq.set_local_code(0, code)
    p = oracle.pcps(**c)
    p.set_local_code(code)
    for rep in range(2):  # a second search after reset() behaves like the first but for the stale scratch
        for d in range(2):
            r = acq.dwell(x[4000 * d:])[0]
            q = p.core(x[4000 * d:])
            _check(r, q, cfar=False)
            assert r.second_peak == pytest.approx(q.second_peak, rel=TOL)
            assert r.second_peak_full_row == pytest.approx(q.second_peak_fixed, rel=TOL)
        assert np.max(np.abs(acq.grid(0) - p.grid())) <= TOL * p.grid().max()
        acq.reset()
        p.reset_grid()
    acq.close()


def test_bit_transition_flag_doubles_the_fft(gctx, oracle):
    import gnsscorr
    k, x = _kat("gps_l1_ca")
    c = _conf(k["fs"], 1, 1, 4000.0, 5000, 500, bit_transition_flag=True)
    code = oracle.gps_l1_ca_code_sampled(1, k["fs"])
    code2 = np.concatenate([code, code])
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    assert (acq.fft_size, acq.consumed_samples) == (16000, 8000)
    acq.set_local_code(0, code2)
    p = oracle.pcps(**c)
    p.set_local_code(code2)
    r, q = acq.dwell(x)[0], p.core(x)
    _check(r, q)
    acq.close()


def test_batched_satellites_25msps(gctx, oracle):
    """cfg4 shape at reduced width: 25 Msps (N = 25000 = 2^3 5^5), 4 PRNs searched at once on one block,
    2 present and 2 absent; 2 dwells."""
    import gnsscorr
    from helpers import synth_stream
    fs, n = 25_000_000, 25000
    prns = [3, 11, 17, 25]
    codes = [oracle.gps_l1_ca_code(p).astype(np.float32) for p in prns]
    x, truth = synth_stream(codes[:2], fs, 2 * n, seed=1004, cn0_db_hz=(46.0, 48.0))
    c = _conf(fs, 1, 1, 25000.0, 5000, 500, max_dwells=2)
    acq = gnsscorr.PcpsAcquisition(gctx, len(prns), **c)
    assert (acq.fft_size, acq.num_doppler_bins) == (25000, 20)
    orcs = []
    for s, prn in enumerate(prns):
        code = oracle.gps_l1_ca_code_sampled(prn, fs)
        acq.set_local_code(s, code)
        p = oracle.pcps(**c)
        p.set_local_code(code)
        orcs.append(p)
    for d in range(2):
        res = acq.dwell(x[d * n:])
        for s in range(len(prns)):
            q = orcs[s].core(x[d * n:])
            _check(res[s], q, cfar=False)
    # present satellites: delay and Doppler agree with the truth; absent ones have a flat grid
    for s in range(2):
        t = truth[s]
        expect = (-t["tau0"] * fs / 1.023e6) % n
        assert min(abs(res[s].indext - expect), n - abs(res[s].indext - expect)) <= 25
        assert abs(res[s].doppler_hz - t["doppler"]) <= 500
        assert res[s].test_statistics > 2.5
    for s in range(2, 4):
        assert res[s].test_statistics < 2.0
    acq.close()


def test_num_doppler_bins_override_41(gctx):
    import gnsscorr
    c = _conf(25_000_000, 1, 1, 25000.0, 5000, 250, num_doppler_bins_override=41)
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    assert acq.num_doppler_bins == 41
    acq.close()
    c = _conf(25_000_000, 1, 1, 25000.0, 5000, 250)
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    assert acq.num_doppler_bins == 40  # ceil(2*5000/250), pcps_acquisition.cc:326
    acq.close()


def test_dwell_without_code_is_a_state_error(gctx):
    import gnsscorr
    c = _conf(4_000_000, 1, 1, 4000.0, 5000, 500)
    acq = gnsscorr.PcpsAcquisition(gctx, 2, **c)
    acq.set_local_code(0, np.ones(4000, np.complex64))
    with pytest.raises(gnsscorr.GnsscorrError) as ei:
        acq.dwell(np.zeros(4000, np.complex64))
    assert ei.value.status == gnsscorr.GC_ERR_STATE
    acq.close()


def test_all_zero_input_gives_index_zero(gctx, oracle):
    """Empty signal: every |.|^2 is 0, strict '>' keeps row 0 / index 0 (index_max semantics)."""
    import gnsscorr
    c = _conf(4_000_000, 1, 1, 4000.0, 5000, 500, use_cfar=False, max_dwells=1)
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    acq.set_local_code(0, oracle.gps_l1_ca_code_sampled(1, 4_000_000))
    r = acq.dwell(np.zeros(4000, np.complex64))[0]
    assert (r.indext, r.doppler_index, r.mag) == (0, 0, 0.0)
    acq.close()


def test_cshort_input_block(gctx, oracle):
    """item_type cshort: the block converts lv_16sc_t to gr_complex at the head of acquisition_core
    (pcps_acquisition.cc:676-679); device int16 input must give the float search of the converted block."""
    import gnsscorr
    import torch
    k, x = _kat("gps_l1_ca")
    q = np.round(x.view(np.float32) * 4000.0).astype(np.int16)
    xf = q.astype(np.float32).view(np.complex64)
    c = _conf(k["fs"], 1, 1, 4000.0, 5000, 250)
    code = oracle.gps_l1_ca_code_sampled(1, k["fs"])
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    acq.set_local_code(0, code)
    acq.set_input_format(gnsscorr.GC_IQ_I16)
    d = torch.from_numpy(q).cuda()
    r = acq.dwell_dev(d.data_ptr())[0]
    p = oracle.pcps(**c)
    p.set_local_code(code)
    _check(r, p.core(xf))
    assert r.indext == 524
    acq.close()


# FFT sizes that walk every instantiated stage list of the packed row kernel (acq_rows2_registry in
# acq_kernels.hip) and the general row kernel (prime radices 11 / 31, or a stage list outside the registry)
@pytest.mark.parametrize("n", [2048, 5456, 6250, 8184, 12000, 24000, 30000, 32000, 32768, 40000, 50000, 64000, 65536, 80000, 100000, 256000, 400000])
def test_fft_sizes_against_oracle(gctx, oracle, n):
    import gnsscorr
    from helpers import synth_stream
    fs = n * 1000
    chips = oracle.gps_l1_ca_code(9).astype(np.float32)
    x, truth = synth_stream([chips], fs, n, seed=n, cn0_db_hz=(50.0, 50.0), doppler_max=900.0)
    c = dict(fs_in=fs, sampled_ms=1, ms_per_code=1, samples_per_ms=np.float32(fs) * np.float32(0.001), samples_per_code=float(n),
        samples_per_chip=int(np.ceil(np.float32(9.7752e-07) * np.float32(fs))), doppler_max=1000, doppler_step=500)
    code = oracle.gps_l1_ca_code_sampled(9, fs)
    assert code.size == n
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    assert (acq.fft_size, acq.num_doppler_bins) == (n, 4)
    acq.set_local_code(0, code)
    r = acq.dwell(x)[0]
    p = oracle.pcps(**c)
    p.set_local_code(code)
    q = p.core(x)
    _check(r, q)
    expect = (-truth[0]["tau0"] * fs / 1.023e6) % n
    assert min(abs(r.indext - expect), n - abs(r.indext - expect)) <= n / 1023.0 + 1
    grid, ref = acq.grid(0), p.grid()
    assert np.max(np.abs(grid - ref)) <= TOL * ref.max()
    acq.close()


def test_frequency_offset_glonass_fdma(gctx, oracle):
    """d_old_freq: the GLONASS FDMA channel offset the reference adds to every Doppler bin in set_local_code()
    (pcps_acquisition.cc:242-247, :276-293, :371-380).  A 511-chip GLONASS-shaped code on frequency channel k = +3
    (3 x 562500 Hz away from the L1 centre) is found at its true Doppler once the offset is installed."""
    import gnsscorr
    fs, n = 8_000_000, 8000
    rng = np.random.Generator(np.random.PCG64(511))
    chips = np.sign(rng.standard_normal(511)).astype(np.float32)
    k_channel, dfrq1 = 3, 562500
    doppler, delay = -1830.0, 2345
    i = np.arange(n)
    idx = np.floor((i - delay) * 0.511e6 / fs).astype(np.int64) % 511
    x = (0.15 * chips[idx] * np.exp(2j * np.pi * (k_channel * dfrq1 + doppler) * i / fs)
        + (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(0.5)).astype(np.complex64)
    code = chips[np.floor(i * 0.511e6 / fs).astype(np.int64) % 511].astype(np.complex64)
    c = dict(fs_in=fs, sampled_ms=1, ms_per_code=1, samples_per_ms=np.float32(fs) * np.float32(0.001), samples_per_code=float(n),
        samples_per_chip=16, doppler_max=5000, doppler_step=250)
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    acq.set_local_code(0, code)
    r0 = acq.dwell(x)[0]            # without the offset the satellite is 1.7 MHz outside the grid
    acq.reset()
    acq.set_frequency_offset(k_channel * dfrq1)
    r = acq.dwell(x)[0]
    p = oracle.pcps(**c)
    p.set_local_code(code)
    p.set_frequency_offset(k_channel * dfrq1)
    q = p.core(x)
    _check(r, q)
    assert r.indext == delay and abs(r.doppler_hz - doppler) <= 250 and r.test_statistics > 3 * r0.test_statistics  # 1 ms: the peak is one bin wide
    acq.close()


def test_glonass_l1_real_capture(gctx, oracle):
    """Real GLONASS L1 data (the capture of the reference's GLONASS tracking tests): the engine with its own GLONASS
    replica generator and the FDMA offset finds what the oracle finds, cell for cell, on three frequency channels, and
    channel 0 is the acquisition hand-over the reference tests hard-code (1343 samples, -2750 Hz)."""
    import gnsscorr
    k, x = _kat("glonass_l1_ca")
    fs = k["fs"]
    c = dict(fs_in=fs, sampled_ms=1, ms_per_code=1, samples_per_ms=np.float32(fs) * np.float32(0.001), samples_per_code=6625.0,
        samples_per_chip=13, doppler_max=k["doppler_max"], doppler_step=k["doppler_step"])
    code = gnsscorr.glonass_l1_ca_code_gen_complex_sampled(fs)
    assert np.array_equal(code, oracle.glonass_l1_ca_code_sampled(fs))
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    assert (acq.fft_size, acq.num_doppler_bins) == (6625, 80)  # 6625 = 5^3 * 53: a prime radix in the row FFT
    acq.set_local_code(0, code)
    for kc, want in k["oracle_by_frequency_channel"].items():
        acq.reset()
        acq.set_frequency_offset(k["dfrq1_glo_hz"] * int(kc))
        r = acq.dwell(x)[0]
        p = oracle.pcps(**c)
        p.set_local_code(code)
        p.set_frequency_offset(k["dfrq1_glo_hz"] * int(kc))
        _check(r, p.core(x))
        assert (r.indext, r.doppler_hz) == (want["indext"], want["doppler"])
    g = k["reference_test"]
    acq.reset()
    acq.set_frequency_offset(0)
    r = acq.dwell(x[6625:])[0]
    assert abs(r.acq_delay_samples - g["expected_delay_samples"]) * 511 / 6625 < g["max_delay_error_chips"]
    assert abs(r.acq_doppler_hz - g["expected_doppler_hz"]) <= g["max_doppler_error_hz"]
    acq.close()


def _engine_against_oracle_everywhere(acq, p, x, bins):
    """One search on `acq` and on the oracle block `p`, then every stage of the engine against the oracle's: the wipe-off rows
    (the oracle's s32f_sincos table is pinned to the compiled reference), FFT(x * wipeoff[bin]) of a few bins, the code spectrum,
    the WHOLE grid, the row maxima the statistics kernel scans, and the result."""
    r = acq.dwell(x)[0]
    q = p.core(x)
    N = acq.fft_size
    W = p.wipeoffs()
    for b in bins:
        w = acq.peek(acq.PEEK_WIPEOFF, b)
        # sincosf of the GPU and of the host libm may differ in the last bit; the running phase itself is exact
        assert np.max(np.abs(w - W[b])) <= 2e-6, "wipe-off row %d" % b
        X = np.fft.fft(x[:N].astype(np.complex128) * W[b].astype(np.complex128))
        assert np.max(np.abs(acq.peek(acq.PEEK_SPECTRUM, b) - X)) <= 2e-6 * np.max(np.abs(X)) * np.sqrt(N), "spectrum of bin %d" % b
    fc = p.fft_codes()
    assert np.max(np.abs(acq.peek(acq.PEEK_CODE, 0) - fc)) <= 2e-6 * np.max(np.abs(fc)) * np.sqrt(N)
    grid, ref = acq.grid(0), p.grid()
    bad = np.argwhere(np.abs(grid - ref) > TOL * ref.max())
    assert bad.size == 0, "grid differs in %d cells, first (bin, index) %s" % (len(bad), bad[:4].tolist())
    rm = acq.peek(acq.PEEK_ROW_MAX, 0)
    assert np.array_equal(rm[:, 1].astype(np.int64), grid.argmax(axis=1)) and np.array_equal(rm[:, 0], grid.max(axis=1))
    _check(r, q)
    return r


def test_engine_reuse_after_frequency_offset_whole_grid(gctx, oracle):
    """The sequence of the one red run of round 3 (`adapter_selftest` GLONASS block, pcps_acquisition.cc:239-247, :296-310, :371-380):
    ONE engine searches frequency channel 0, has its wipe-off tables rebuilt for channel -4 (set_frequency_offset) and its code
    re-installed (set_local_code), and searches again -- at N = 6625 = 5^3 x 53 (general row kernel, prime radix 53) on the real
    capture.  Every stage is compared with the oracle after each search, not only the peak cell; the second search's peak is the
    one the self-test expects (delay 502, 3000 Hz)."""
    import gnsscorr
    k, x = _kat("glonass_l1_ca")
    fs = k["fs"]
    c = dict(fs_in=fs, sampled_ms=1, ms_per_code=1, samples_per_ms=np.float32(fs) * np.float32(0.001), samples_per_code=6625.0,
        samples_per_chip=13, doppler_max=k["doppler_max"], doppler_step=k["doppler_step"])
    code = gnsscorr.glonass_l1_ca_code_gen_complex_sampled(fs)
    x = x[:6625]
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    acq.set_local_code(0, code)
    p = oracle.pcps(**c)
    p.set_local_code(code)
    bins = (0, 28, 29, 50, 51, 52, 53, 54, 79)
    r = _engine_against_oracle_everywhere(acq, p, x, bins)
    assert (r.indext, r.doppler_hz) == (1344, -2750)
    for kc, want in ((-4, (502, 3000)), (0, (1344, -2750)), (-4, (502, 3000))):
        acq.reset()
        acq.set_frequency_offset(k["dfrq1_glo_hz"] * kc)
        acq.set_local_code(0, code)
        p = oracle.pcps(**c)
        p.set_local_code(code)
        p.set_frequency_offset(k["dfrq1_glo_hz"] * kc)
        r = _engine_against_oracle_everywhere(acq, p, x, bins)
        assert (r.indext, r.doppler_hz) == want
    acq.close()


def test_engine_reuse_after_frequency_offset_whole_grid_25msps(gctx, oracle):
    """The same reuse sequence at N = 25000 (the pair row kernel and the 25-point column pass of the bench configuration), 40 bins:
    search, rebuild the tables with an intermediate-frequency offset the signal really has, re-install the code, search."""
    import gnsscorr
    from helpers import synth_stream
    fs, N = 25_000_000, 25000
    code_chips = oracle.gps_l1_ca_code(7)
    x, truth = synth_stream([code_chips], fs, N, seed=2504, cn0_db_hz=(47.0, 47.0), doppler_max=4000.0)
    offset = 1_250_000
    xo = (x * np.exp(2j * np.pi * offset * np.arange(N) / fs)).astype(np.complex64)
    c = _conf(fs, 1, 1, float(N), 5000, 250)  # 40 bins (the oracle block has the reference's bin count, no override)
    code = oracle.gps_l1_ca_code_sampled(7, fs)
    acq = gnsscorr.PcpsAcquisition(gctx, 1, **c)
    acq.set_local_code(0, code)
    bins = (0, 19, 20, 21, 39)
    for off, sig in ((0, x), (offset, xo), (0, x), (offset, xo)):
        acq.reset()
        acq.set_frequency_offset(off)
        acq.set_local_code(0, code)
        p = oracle.pcps(**c)
        p.set_local_code(code)
        p.set_frequency_offset(off)
        r = _engine_against_oracle_everywhere(acq, p, sig, bins)
        assert abs(r.doppler_hz - truth[0]["doppler"]) <= 250
    acq.close()


def test_galileo_e1_real_capture_batched(gctx, oracle):
    """Real Galileo E1 data (GSoC 2012 roof capture, with the MATLAB analysis the reference ships beside it): PRN 11 and 12
    present, 19 and 20 absent, searched in one batched call on a 160-bin grid; every result equals the oracle's cell for
    cell and the present satellites sit where the analysis puts them."""
    import gnsscorr
    k, x = _kat("galileo_e1_real_capture")
    fs = k["fs"]
    e1b = np.load(os.path.join(G, "galileo_e1_codes.npz"))["e1b"]
    c = dict(fs_in=fs, sampled_ms=4, ms_per_code=4, samples_per_ms=np.float32(fs) * np.float32(0.001), samples_per_code=16000.0,
        samples_per_chip=4, doppler_max=k["doppler_max"], doppler_step=k["doppler_step"])
    prns = [11, 12, 19, 20]
    acq = gnsscorr.PcpsAcquisition(gctx, len(prns), **c)
    orcs = []
    for s_, prn in enumerate(prns):
        code = gnsscorr.galileo_e1_code_gen_complex_sampled("1B", False, prn, fs)
        assert np.array_equal(code.real, oracle.galileo_e1_code_sampled(e1b[prn - 1], fs, cboc=False))
        acq.set_local_code(s_, code)
        p = oracle.pcps(**c)
        p.set_local_code(code)
        orcs.append(p)
    res = acq.dwell(x)
    for s_, prn in enumerate(prns):
        _check(res[s_], orcs[s_].core(x))
    for prn, a in ((11, k["reference_analysis"]["11"]), (12, k["reference_analysis"]["12"])):
        r = res[prns.index(prn)]
        assert r.indext == a["delay_samples"] and abs(r.doppler_hz) == a["abs_doppler_hz"]
    assert min(res[0].test_statistics, res[1].test_statistics) > 2.0 * max(res[2].test_statistics, res[3].test_statistics)
    acq.close()


def test_cfg4_full_width_32_prns_41_bins_2_dwells(gctx, oracle):
    """BASELINE configs[3] at full width, through the path bench.py times (dwell_enqueue x 2 + fetch_results on a caller stream):
    GPS L1 C/A, 25 Msps, N = 25000, 32 PRNs x 41 Doppler bins x 2 non-coherent dwells; PRN 1-16 present at C/N0 in [38, 48] dB-Hz,
    PRN 17-32 absent, CFAR off because max_dwells = 2 (pcps_acquisition.cc:152-159).  41 bins come from the
    reference's own formula with doppler_max = 5125 (ceil(10250 / 250), :326), so the oracle searches the same grid.  With the
    default 140 MB inter-pass buffer the 32 satellites run as 16 + 16 (gc_acquisition.hip); every satellite of both batches is
    checked against the oracle's acquisition_core (:668-770) after EACH dwell: peak cell exact, magnitudes and statistics <= 1e-4."""
    import gnsscorr
    import torch
    from helpers import synth_stream
    fs, n = 25_000_000, 25000
    prns = list(range(1, 33))
    chips = [oracle.gps_l1_ca_code(p).astype(np.float32) for p in prns[:16]]
    x, truth = synth_stream(chips, fs, 2 * n, seed=1004, cn0_db_hz=(38.0, 48.0))
    c = _conf(fs, 1, 1, 25000.0, 5125, 250, max_dwells=2)
    acq = gnsscorr.PcpsAcquisition(gctx, 32, **c)
    assert (acq.fft_size, acq.num_doppler_bins) == (25000, 41)
    orcs = []
    for s, prn in enumerate(prns):
        code = oracle.gps_l1_ca_code_sampled(prn, fs)
        acq.set_local_code(s, code)
        p = oracle.pcps(**c)
        p.set_local_code(code)
        orcs.append(p)
    d_x = torch.from_numpy(x.view(np.float32)).cuda()
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    per_dwell = []
    for rep in range(2):  # the second search, after reset(), must store over the first one's grid (no memset in between)
        acq.reset()
        for d in range(2):
            acq.dwell_enqueue(d_x.data_ptr() + 8 * n * d, st.cuda_stream)
            if rep == 0:
                per_dwell.append(acq.fetch_results(st.cuda_stream))
        last = acq.fetch_results(st.cuda_stream)
    for s in range(32):
        orcs[s].reset_grid()
        for d in range(2):
            q = orcs[s].core(x[d * n:])
            _check(per_dwell[d][s], q, cfar=False)
            assert per_dwell[d][s].second_peak == pytest.approx(q.second_peak, rel=TOL)
        _check(last[s], q, cfar=False)
        # the second search enqueued its dwells back to back: they ran as ONE pair (gc_acquisition.hip, inv_pending: one row pass over both
        # dwells' spectra, one column pass that writes the grid once).  Same additions in the same order: not a bit may differ
        a_, b_ = per_dwell[1][s], last[s]
        assert (a_.indext, a_.doppler_hz, a_.mag, a_.test_statistics, a_.second_peak) == (b_.indext, b_.doppler_hz, b_.mag, b_.test_statistics, b_.second_peak)
    # a satellite of each batch, whole grid
    for s in (3, 29):
        ref = orcs[s].grid()
        assert np.max(np.abs(acq.grid(s) - ref)) <= TOL * ref.max()
    # the strong half of the present satellites is found where the truth puts it; nothing absent beats them
    stats = np.array([r.test_statistics for r in last])
    strong = [s for s in range(16) if truth[s]["cn0"] >= 44.0]
    assert len(strong) >= 4
    for s in strong:
        expect = (-truth[s]["tau0"] * fs / 1.023e6) % n
        assert min(abs(last[s].indext - expect), n - abs(last[s].indext - expect)) <= 25 and abs(last[s].doppler_hz - truth[s]["doppler"]) <= 250
    assert min(stats[strong]) > max(stats[16:])
    acq.close()


def test_uneven_satellite_batches(gctx, oracle, monkeypatch):
    """The inter-pass buffer forced down to 1 MB (GNSSCORR_ACQ_Q_MB, read when the engine is created): 7 satellites x 10 bins at
    N = 4000 (320 KB per satellite) run as 3 + 3 + 1 -- the ragged last batch -- and 2 dwells accumulate across them."""
    import gnsscorr
    from helpers import synth_stream
    monkeypatch.setenv("GNSSCORR_ACQ_Q_MB", "1")
    fs, n = 4_000_000, 4000
    prns = [2, 5, 9, 14, 21, 27, 30]
    chips = [oracle.gps_l1_ca_code(p).astype(np.float32) for p in prns[:4]]
    x, truth = synth_stream(chips, fs, 2 * n, seed=77, cn0_db_hz=(44.0, 50.0), doppler_max=2000.0)
    c = _conf(fs, 1, 1, 4000.0, 2500, 500, max_dwells=2)
    acq = gnsscorr.PcpsAcquisition(gctx, len(prns), **c)
    assert acq.num_doppler_bins == 10
    orcs = []
    for s, prn in enumerate(prns):
        code = oracle.gps_l1_ca_code_sampled(prn, fs)
        acq.set_local_code(s, code)
        p = oracle.pcps(**c)
        p.set_local_code(code)
        orcs.append(p)
    for d in range(2):
        res = acq.dwell(x[d * n:])
        for s in range(len(prns)):
            _check(res[s], orcs[s].core(x[d * n:]), cfar=False)
    for s in range(len(prns)):
        ref = orcs[s].grid()
        assert np.max(np.abs(acq.grid(s) - ref)) <= TOL * ref.max()
    assert min(r.test_statistics for r in res[:4]) > max(r.test_statistics for r in res[4:])
    acq.close()


def _pairs_with_small_buffer_and_cshort_input(gctx, oracle, monkeypatch):
    """The pair path where it splits differently from the per-dwell path: a 1 MB inter-pass buffer holds three satellites of one
    dwell but only ONE satellite pair-wise (7 batches instead of 3 + 3 + 1), input blocks are lv_16sc_t (converted per dwell), the
    second search changes a code in between (the held-back dwell must be searched with the code of its time)."""
    import gnsscorr
    import torch
    from helpers import synth_stream
    fs, n = 4_000_000, 4000
    prns = [2, 5, 9, 14, 21, 27, 30]
    chips = [oracle.gps_l1_ca_code(p).astype(np.float32) for p in prns[:4]]
    x, _ = synth_stream(chips, fs, 3 * n, seed=91, cn0_db_hz=(44.0, 50.0), doppler_max=2000.0)
    xi = np.clip(np.round(x.view(np.float32) * 200.0), -32768, 32767).astype(np.int16)
    c = _conf(fs, 1, 1, 4000.0, 2500, 500, max_dwells=3)
    monkeypatch.setenv("GNSSCORR_ACQ_Q_MB", "1")
    eng = []
    for _ in range(2):
        a = gnsscorr.PcpsAcquisition(gctx, len(prns), **c)
        a.set_input_format(gnsscorr.GC_IQ_I16)
        for s_, prn in enumerate(prns):
            a.set_local_code(s_, oracle.gps_l1_ca_code_sampled(prn, fs))
        eng.append(a)
    d_x = torch.from_numpy(xi).cuda()
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    out = []
    for k_, a in enumerate(eng):
        per_dwell = k_ == 1  # the second engine completes every dwell by itself (gc_acq_flush), as acquisition_core does
        a.reset()
        a.dwell_enqueue(d_x.data_ptr(), st.cuda_stream)
        if per_dwell:
            a.flush(st.cuda_stream)
        a.set_local_code(6, oracle.gps_l1_ca_code_sampled(31, fs))  # between the dwells of what would have been a pair
        a.dwell_enqueue(d_x.data_ptr() + 4 * n, st.cuda_stream)
        if per_dwell:
            a.flush(st.cuda_stream)
        a.dwell_enqueue(d_x.data_ptr() + 8 * n, st.cuda_stream)
        out.append(a.fetch_results(st.cuda_stream))
    for s_ in range(len(prns)):
        assert np.array_equal(eng[0].grid(s_), eng[1].grid(s_)), s_
        r1, r0 = out[0][s_], out[1][s_]
        assert (r1.indext, r1.doppler_hz, r1.mag, r1.test_statistics) == (r0.indext, r0.doppler_hz, r0.mag, r0.test_statistics)
    for a in eng:
        a.close()


@pytest.mark.parametrize("n_dwells", [2, 3, 4, 5])
def test_dwell_pairs_equal_per_dwell_processing(gctx, oracle, monkeypatch, n_dwells):
    """Dwells enqueued back to back are searched in pairs (the first of a pair is held back until the second arrives; an odd one
    and anything followed by a fetch run alone): 2 = one pair, 3 = pair + accumulating single, 4 = pair + accumulating pair,
    5 = pair + pair + single.  A second engine completes every dwell by itself (gc_acq_flush behind each one: inverse passes and
    the statistics evaluation), as acquisition_core does (pcps_acquisition.cc:668-770, :747-755): grids and results must be
    IDENTICAL, and the per-dwell engine is checked against the oracle."""
    import gnsscorr
    import torch
    from helpers import synth_stream
    fs, n = 4_000_000, 4000
    prns = [3, 7, 11, 19, 23]
    chips = [oracle.gps_l1_ca_code(p).astype(np.float32) for p in prns[:3]]
    x, _ = synth_stream(chips, fs, n_dwells * n, seed=500 + n_dwells, cn0_db_hz=(40.0, 46.0), doppler_max=2000.0)
    c = _conf(fs, 1, 1, 4000.0, 2500, 500, max_dwells=n_dwells)
    engines = []
    for _ in range(2):
        a = gnsscorr.PcpsAcquisition(gctx, len(prns), **c)
        for s, prn in enumerate(prns):
            a.set_local_code(s, oracle.gps_l1_ca_code_sampled(prn, fs))
        engines.append(a)
    d_x = torch.from_numpy(x.view(np.float32)).cuda()
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    res = []
    for k_, a in enumerate(engines):
        a.reset()
        for d in range(n_dwells):
            a.dwell_enqueue(d_x.data_ptr() + 8 * n * d, st.cuda_stream)
            if k_ == 1:
                a.flush(st.cuda_stream)  # per-dwell processing
        res.append(a.fetch_results(st.cuda_stream))
    for s, prn in enumerate(prns):
        g1, g0 = engines[0].grid(s), engines[1].grid(s)
        assert np.array_equal(g1, g0), (s, np.max(np.abs(g1 - g0)))
        r1, r0 = res[0][s], res[1][s]
        assert (r1.indext, r1.doppler_hz, r1.mag, r1.test_statistics, r1.second_peak) == (r0.indext, r0.doppler_hz, r0.mag, r0.test_statistics, r0.second_peak)
        p = oracle.pcps(**c)
        p.set_local_code(oracle.gps_l1_ca_code_sampled(prn, fs))
        for d in range(n_dwells):
            q = p.core(x[d * n:])
        _check(r0, q, cfar=False)
        assert r0.second_peak == pytest.approx(q.second_peak, rel=TOL)
    if n_dwells == 3:
        _pairs_with_small_buffer_and_cshort_input(gctx, oracle, monkeypatch)
    # a held-back dwell is not lost when the caller looks before its partner arrives: fetch after the first dwell, then go on
    a = engines[0]
    a.reset()
    a.dwell_enqueue(d_x.data_ptr(), st.cuda_stream)
    first = a.fetch_results(st.cuda_stream)
    b = engines[1]
    b.reset()
    b.dwell_enqueue(d_x.data_ptr(), st.cuda_stream)
    assert [(r.indext, r.mag, r.test_statistics) for r in first] == [(r.indext, r.mag, r.test_statistics) for r in b.fetch_results(st.cuda_stream)]
    a.dwell_enqueue(d_x.data_ptr() + 8 * n, st.cuda_stream)
    b.dwell_enqueue(d_x.data_ptr() + 8 * n, st.cuda_stream)
    assert np.array_equal(a.grid(0), b.grid(0))
    for e in engines:
        e.close()


def test_gpu_grid_against_the_reference_matlab_analysis(gctx, oracle):
    """The same reference-held figures (plot_acq_grid_gsoc.m on the GSoC 2012 capture: PRN 11 10.8538 dB, PRN 12 9.3968 dB of peak
    over noise floor; tests/test_oracle_golden.py has the details) from the GPU's own grid, with the engine's own E1C replica."""
    import gnsscorr
    from helpers import gsoc_grid_statistics
    k, x = _kat("galileo_e1_real_capture")
    fs = k["fs"]
    c = dict(fs_in=fs, sampled_ms=4, ms_per_code=4, samples_per_ms=np.float32(fs) * np.float32(0.001), samples_per_code=16000.0,
        samples_per_chip=4, doppler_max=k["doppler_max"], doppler_step=k["doppler_step"])
    acq = gnsscorr.PcpsAcquisition(gctx, 2, **c)
    for s_, prn in enumerate((11, 12)):
        acq.set_local_code(s_, gnsscorr.galileo_e1_code_gen_complex_sampled("1C", False, prn, fs))
    acq.dwell(x)
    for s_, prn in enumerate((11, 12)):
        a = k["reference_analysis"][str(prn)]
        peak, floor_, gain, row, col = gsoc_grid_statistics(acq.grid(s_), fs, k["doppler_step"])
        assert col == a["delay_samples"] and abs(-k["doppler_max"] + k["doppler_step"] * row) == a["abs_doppler_hz"]
        assert abs(gain - a["gain_db"]) < 0.02, (prn, gain)
        n2 = float(acq.fft_size) ** 2
        assert abs(peak / n2 / a["maximum_correlation_peak"] - 1.0) < 3e-3 and abs(floor_ / n2 / a["noise_floor"] - 1.0) < 3e-3, (prn, peak / n2, floor_ / n2)
    acq.close()


def test_flush_completes_held_back_work_without_a_fetch(gctx, oracle):
    """gc_acq_flush (pcps_acquisition.cc:747-755 evaluates after every dwell): enqueues a held-back dwell's inverse passes and the
    statistics kernel, copies nothing; idempotent; results after flush + fetch equal the results of a plain fetch; flushing an idle
    engine is a no-op."""
    import gnsscorr
    import torch
    from helpers import synth_stream
    fs, n = 4_000_000, 4000
    prns = [4, 8, 15]
    chips = [oracle.gps_l1_ca_code(p).astype(np.float32) for p in prns[:2]]
    x, _ = synth_stream(chips, fs, 2 * n, seed=77, cn0_db_hz=(44.0, 48.0), doppler_max=1500.0)
    c = _conf(fs, 1, 1, 4000.0, 2500, 500, max_dwells=2)
    d_x = torch.from_numpy(x.view(np.float32)).cuda()
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    engines = []
    for _ in range(2):
        a = gnsscorr.PcpsAcquisition(gctx, len(prns), **c)
        for s, prn in enumerate(prns):
            a.set_local_code(s, oracle.gps_l1_ca_code_sampled(prn, fs))
        engines.append(a)
    a, b = engines
    a.flush(st.cuda_stream)  # nothing pending
    for e in engines:
        e.reset()
        e.dwell_enqueue(d_x.data_ptr(), st.cuda_stream)
    a.flush(st.cuda_stream)
    a.flush(st.cuda_stream)  # idempotent
    ra, rb = a.fetch_results(st.cuda_stream), b.fetch_results(st.cuda_stream)
    assert [(r.indext, r.doppler_hz, r.mag, r.test_statistics) for r in ra] == [(r.indext, r.doppler_hz, r.mag, r.test_statistics) for r in rb]
    for e in engines:
        e.dwell_enqueue(d_x.data_ptr() + 8 * n, st.cuda_stream)
    a.flush(st.cuda_stream)
    ra, rb = a.fetch_results(st.cuda_stream), b.fetch_results(st.cuda_stream)
    for s in range(len(prns)):
        assert np.array_equal(a.grid(s), b.grid(s))
        assert (ra[s].indext, ra[s].mag, ra[s].test_statistics, ra[s].second_peak) == (rb[s].indext, rb[s].mag, rb[s].test_statistics, rb[s].second_peak)
    for e in engines:
        e.close()
