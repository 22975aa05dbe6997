"""GPU parity of the complex-chip correlator: the image of the
reference's Cpu_Multicorrelator (tracking/libs/cpu_multicorrelator.cc:82-130; GLONASS L1/L2 and
GPS L1 C-Aid trackers) against the oracle's restatement of
volk_gnsssdr_32fc_xn_resampler_32fc_xn + volk_gnsssdr_32fc_x2_rotator_dot_prod_32fc_xn.
Tolerance: 1e-4 of |Prompt| (float32, north_star); the resampler part is pinned bit for bit on the
CPU side (tests/test_oracle_golden.py), the rotator part is "parity unpinned" like the real-code one."""
import numpy as np
import pytest

from helpers import open_loop_params, rel_err, synth_stream

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _level1(gctx, sig, code, shifts, p, n):
    import gnsscorr
    corr = gnsscorr.HipMulticorrelator(gctx)
    out = np.zeros(len(shifts), np.complex64)
    shifts = shifts.copy()
    assert corr.init(2 * n, len(shifts))
    assert corr.set_local_code_and_taps(len(code), code, shifts)
    assert corr.set_input_output_vectors(out, sig[p["sample_offset"]:])
    assert corr.Carrier_wipeoff_multicorrelator_resampler(float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n)
    res = out.copy()
    assert corr.free()
    corr.close()
    return res


def test_level1_gps_ca_complex_code_4msps(gctx, oracle):
    """The C-Aid GPS tracker's replica: gps_l1_ca_code_gen_complex gives (+-1, 0) chips."""
    import gnsscorr
    chips = oracle.gps_l1_ca_code(7).astype(np.float32)
    code = chips.astype(np.complex64)
    fs, n = 4_000_000, 4000
    sig, truth = synth_stream([chips], fs, 3 * n, seed=1001, cn0_db_hz=(45.0, 45.0))
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    for p in open_loop_params(truth[0], fs, 1023, n, 3):
        ref = oracle.multicorrelator_cc(sig[p["sample_offset"]:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n)
        got = _level1(gctx, sig, code, shifts, p, n)
        assert abs(ref[1]) > 0.5 * truth[0]["amp"] * n
        assert rel_err(got, ref, 1) <= TOL
        # zero imaginary chips: the real-code correlator must give the same sums
        real = gnsscorr.HipMulticorrelatorRealCodes(gctx)
        real.set_high_dynamics_resampler(False)
        out = np.zeros(3, np.complex64)
        real.init(2 * n, 3)
        sh = shifts.copy()
        real.set_local_code_and_taps(1023, chips, sh)
        real.set_input_output_vectors(out, sig[p["sample_offset"]:])
        real.Carrier_wipeoff_multicorrelator_resampler(float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), 0.0, n)
        assert rel_err(got, out, 1) <= 2e-6
        real.close()


def test_level1_truly_complex_chips(gctx, oracle):
    """Chips with both parts non-zero and non-unit modulus exercise the full complex MAC."""
    rng = np.random.Generator(np.random.PCG64(77))
    L, fs, n = 511, 4_000_000, 8000
    chips = np.sign(rng.standard_normal(L)).astype(np.float32)
    code = (chips * np.exp(1j * rng.uniform(0, 2 * np.pi, L)) * rng.uniform(0.5, 1.5, L)).astype(np.complex64)
    sig, truth = synth_stream([chips], fs, 2 * n, seed=78, cn0_db_hz=(50.0, 50.0), chip_rate=0.511e6)
    shifts = np.array([-0.5, 0.0, 0.5, 1.0], np.float32)
    for p in open_loop_params(truth[0], fs, L, n, 2):
        ref = oracle.multicorrelator_cc(sig[p["sample_offset"]:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n)
        got = _level1(gctx, sig, code, shifts, p, n)
        scale = np.abs(ref).max()
        assert np.max(np.abs(got - ref)) / scale <= TOL


@pytest.mark.parametrize("fmt", ["f32", "i16"])
def test_batch_complex_codes(gctx, oracle, fmt):
    """GLONASS-shaped batch: 511-chip complex replicas, 3 channels x 4 epochs, float and int16 IQ."""
    import gnsscorr
    import torch
    rng = np.random.Generator(np.random.PCG64(5))
    L, fs, n, n_epochs = 511, 6_250_000, 6250, 4
    codes = [np.sign(rng.standard_normal(L)).astype(np.float32) for _ in range(3)]
    sig, truth = synth_stream(codes, fs, n_epochs * n + 8, seed=6, cn0_db_hz=(46.0, 50.0), chip_rate=0.511e6, carrier_freq=1602e6)
    if fmt == "i16":
        q = np.round(sig.view(np.float32) * 64.0).astype(np.int16)
        sig = q.astype(np.float32).view(np.complex64)
        d_sig = torch.from_numpy(q).cuda()
    else:
        d_sig = torch.from_numpy(sig.view(np.float32).copy()).cuda()
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    b = gnsscorr.TrackingBatch(gctx, 3, 3, L)
    b.set_complex_codes(True)
    if fmt == "i16":
        b.set_input_format(gnsscorr.GC_IQ_I16)
    recs, refs = [], []
    for ch in range(3):
        ccode = (codes[ch] * (1 + 0.25j)).astype(np.complex64)
        b.set_code_complex(ch, ccode, shifts)
        b.set_input_dev(ch, d_sig.data_ptr(), sig.size)
        ps = open_loop_params(truth[ch], fs, L, n, n_epochs)
        recs.append([gnsscorr.epoch_params(p["sample_offset"], float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n) for p in ps])
        refs.append([oracle.multicorrelator_cc(sig[p["sample_offset"]:], ccode, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n) for p in ps])
    out = b.run(n_epochs, gnsscorr.epoch_params_array(recs))
    for ch in range(3):
        for k in range(n_epochs):
            assert rel_err(out[ch, k], refs[ch][k], 1) <= TOL
    b.close()


def test_complex_code_state_errors(gctx):
    import gnsscorr
    shifts = np.zeros(3, np.float32)
    c = gnsscorr.HipMulticorrelatorRealCodes(gctx)
    c.init(100, 3)
    out, sig = np.zeros(3, np.complex64), np.zeros(100, np.complex64)
    c.set_local_code_and_taps(10, np.ones(10, np.float32), shifts)
    c.set_input_output_vectors(out, sig)
    lib = gnsscorr.load_library()
    assert lib.gc_correlator_carrier_wipeoff_multicorrelator_resampler_5(c._h, 0.0, 0.0, 0.0, 0.1, 100) == gnsscorr.GC_ERR_STATE
    c.close()
    b = gnsscorr.TrackingBatch(gctx, 1, 3, 1023, high_dyn=True)
    with pytest.raises(gnsscorr.GnsscorrError):
        b.set_complex_codes(True)
    b.close()
    b = gnsscorr.TrackingBatch(gctx, 1, 3, 1023)
    with pytest.raises(gnsscorr.GnsscorrError):
        b.set_code_complex(0, np.ones(1023, np.complex64), shifts)  # batch not switched yet
    b.set_complex_codes(True)
    with pytest.raises(gnsscorr.GnsscorrError):
        b.set_code(0, np.ones(1023, np.float32), shifts)
    b.close()
    with pytest.raises(gnsscorr.GnsscorrError):
        gnsscorr.TrackingBatch(gctx, 1, 3, 8184).set_complex_codes(True)  # would not fit the LDS window


def test_glonass_real_capture_through_the_complex_chip_correlator(gctx, oracle):
    """The reference's GLONASS trackers hold a Cpu_Multicorrelator with the complex GLONASS replica
    (glonass_l1_ca_dll_pll_tracking_cc.cc); here its image correlates the real NT1065 capture at the hand-over the
    reference's test uses (delay 1343 samples, Doppler -2750 Hz): prompt on the peak, early / late half a chip down."""
    import json
    import os
    import gnsscorr
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    k = json.load(open(os.path.join(G, "kat_expected.json")))["glonass_l1_ca"]
    x = np.fromfile(os.path.join(G, k["file"]), np.complex64)
    fs, n = k["fs"], 6625
    code = gnsscorr.glonass_l1_ca_code_gen_float().astype(np.complex64)  # glonass_l1_ca_code_gen_complex: (+-1, 0)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    delay = k["oracle_by_frequency_channel"]["0"]["indext"]
    dopp = float(k["reference_test"]["expected_doppler_hz"])
    step = np.float32(511.0 / n)
    mags = []
    for ms in range(3):
        p = dict(sample_offset=delay + ms * n, rem_carr=np.float32(0.0), phase_step=np.float32(2 * np.pi * dopp / fs), rem_code=np.float32(0.0), code_step=step)
        ref = oracle.multicorrelator_cc(x[p["sample_offset"]:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n)
        got = _level1(gctx, x, code, shifts, p, n)
        assert rel_err(got, ref, 1) <= TOL
        mags.append(np.abs(got))
    m = np.mean(mags, axis=0)
    assert m[1] > 1.3 * m[0] and m[1] > 1.3 * m[2] and m[1] > 800.0  # a correlation peak (sums of pure noise are ~ sqrt(6625) * 2.1 = 170)
