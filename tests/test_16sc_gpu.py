"""GPU parity of the 16-bit correlator: the image of the reference's
Cpu_Multicorrelator_16sc (tracking/libs/cpu_multicorrelator_16sc.cc:78-103) against the oracle's restatement of
volk_gnsssdr_16ic_xn_resampler_16ic_xn + volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn (generic).

This path is integer arithmetic fed by a float32 rotator.  What is exact here and what is not:
  * chip indices, the int16 products and their summation are exact integer work -- the GPU forms the sums in
    32 bits in any order and saturates once, which equals the reference's running saturating sum whenever no
    running sum leaves the int16 range (checked below against the oracle's unsaturated sums);
  * each rotated sample is rounded with rintf(): the reference rounds x*phase with its sequentially rounded
    phase (25000 dependent float multiplications), the GPU with the phase evaluated directly; both are within
    ~1e-6 of the true phase, so a sample's rounding differs only when x*phase falls within ~1e-3 LSB of a
    half-integer.  Bar: every component within MAX_LSB of the oracle, and the number of differing LSBs small.
The resampler index walk is pinned bit for bit by the compiled reference (tests/test_oracle_golden.py); the
rotator is "parity unpinned" (its header needs the generated volk_gnsssdr.h)."""
import numpy as np
import pytest

from helpers import open_loop_params, synth_stream

pytestmark = pytest.mark.gpu
MAX_LSB = 3


def _quantise(sig, scale):
    q = np.round(sig.view(np.float32).reshape(-1, 2) * scale)
    return np.clip(q, -32768, 32767).astype(np.int16)


def test_level1_16sc_gps_ca(gctx, oracle):
    import gnsscorr
    chips = oracle.gps_l1_ca_code(4).astype(np.float32)
    code = np.stack([chips, np.zeros_like(chips)], 1).astype(np.int16)  # gps_l1_ca_code_gen_complex -> lv_16sc_t
    fs, n = 4_000_000, 4000
    sig, truth = synth_stream([chips], fs, 4 * n, seed=21, cn0_db_hz=(50.0, 50.0))
    q = _quantise(sig, 16.0)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    total_diff = 0
    for p in open_loop_params(truth[0], fs, 1023, n, 4):
        ref, exact = oracle.multicorrelator_16sc(q[p["sample_offset"]:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n)
        assert np.array_equal(ref.astype(np.int32), exact), "the test input must not saturate the reference's running sums"
        corr = gnsscorr.HipMulticorrelator16sc(gctx)
        out = np.zeros((3, 2), np.int16)
        sh = shifts.copy()
        assert corr.init(2 * n, 3)
        assert corr.set_local_code_and_taps(1023, code, sh)
        assert corr.set_input_output_vectors(out, q[p["sample_offset"]:])
        assert corr.Carrier_wipeoff_multicorrelator_resampler(float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n)
        corr.close()
        assert abs(int(ref[1, 0])) + abs(int(ref[1, 1])) > 1000  # the prompt sees the signal
        d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= MAX_LSB, (out, ref)
        total_diff += int(d.sum())
    assert total_diff <= 8  # a handful of half-LSB rounding flips over 4 epochs x 3 taps x 4000 samples


def test_batch_16sc_complex_chips_and_slices(gctx, oracle):
    """Chips with both parts set (the int16 product wraps like the reference's), 25 Msps windows cut in slices."""
    import gnsscorr
    import torch
    rng = np.random.Generator(np.random.PCG64(9))
    L, fs, n, n_epochs = 1023, 25_000_000, 25000, 3
    chips = [np.sign(rng.standard_normal(L)).astype(np.float32) for _ in range(2)]
    sig, truth = synth_stream(chips, fs, n_epochs * n + 8, seed=10, cn0_db_hz=(47.0, 50.0))
    q = _quantise(sig, 4.0)
    d_sig = torch.from_numpy(q).cuda()
    shifts = np.array([-0.5, 0.0, 0.5, 1.0], np.float32)
    for slices in (0, 7):
        b = gnsscorr.TrackingBatch(gctx, 2, 4, L)
        b.set_16sc(True)
        if slices:
            b.set_slices(slices)
        recs, refs = [], []
        for ch in range(2):
            code = np.stack([chips[ch] * (1 + ch), chips[ch] * ch], 1).astype(np.int16)  # (1,0) chips, then (2,1) chips
            b.set_code_16sc(ch, code, shifts)
            b.set_input_dev(ch, d_sig.data_ptr(), q.shape[0])
            ps = open_loop_params(truth[ch], fs, L, n, n_epochs)
            recs.append([gnsscorr.epoch_params(p["sample_offset"], float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n) for p in ps])
            refs.append([oracle.multicorrelator_16sc(q[p["sample_offset"]:], code, shifts, p["rem_carr"], p["phase_step"], p["rem_code"], p["code_step"], n) for p in ps])
        out = b.run(n_epochs, gnsscorr.epoch_params_array(recs))
        assert out.dtype == np.int16 and out.shape == (2, n_epochs, 4, 2)
        for ch in range(2):
            for k in range(n_epochs):
                ref, exact = refs[ch][k]
                if np.array_equal(ref.astype(np.int32), exact):
                    assert np.abs(out[ch, k].astype(np.int32) - ref.astype(np.int32)).max() <= MAX_LSB
                else:
                    # the reference's running sum saturated on the way: only the single final saturation is defined here
                    assert np.abs(out[ch, k].astype(np.int32) - np.clip(exact, -32768, 32767)).max() <= MAX_LSB
        b.close()


def test_16sc_saturates_once(gctx, oracle):
    """A strong input drives the sums past int16: the result is the exact sum clamped to [-32768, 32767]."""
    import gnsscorr
    chips = oracle.gps_l1_ca_code(1).astype(np.float32)
    code = np.stack([chips, np.zeros_like(chips)], 1).astype(np.int16)
    n, step = 4000, np.float32(1023.0 / 4000.0)
    idx = np.floor(step * np.arange(n, dtype=np.float32)).astype(np.int64) % 1023
    q = np.stack([chips[idx] * 100, np.zeros(n)], 1).astype(np.int16)  # noiseless, amplitude 100: prompt = 400000
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    ref, exact = oracle.multicorrelator_16sc(q, code, shifts, 0.0, 0.0, 0.0, float(step), n)
    corr = gnsscorr.HipMulticorrelator16sc(gctx)
    out = np.zeros((3, 2), np.int16)
    corr.init(n, 3)
    corr.set_local_code_and_taps(1023, code, shifts.copy())
    corr.set_input_output_vectors(out, q)
    corr.Carrier_wipeoff_multicorrelator_resampler(0.0, 0.0, 0.0, float(step), n)
    corr.close()
    assert exact[1, 0] == 400000 and ref[1, 0] == 32767
    assert np.array_equal(out, np.clip(exact, -32768, 32767).astype(np.int16))
    # the prompt's running sum grows monotonically, so the reference ends on the same rail; the early / late sums
    # do not (a mismatching chip after the rail has been reached pulls the reference's running value back down,
    # e.g. 32667): that order dependence is what the single final saturation does not reproduce
    assert np.array_equal(out[1], ref[1])


def test_16sc_state_errors(gctx):
    import gnsscorr
    lib = gnsscorr.load_library()
    c = gnsscorr.HipMulticorrelator16sc(gctx)
    c.init(100, 3)
    c.set_local_code_and_taps(10, np.ones((10, 2), np.int16), np.zeros(3, np.float32))
    # float vectors with a 16-bit code
    real = gnsscorr.HipMulticorrelatorRealCodes.set_input_output_vectors
    real(c, np.zeros(3, np.complex64), np.zeros(100, np.complex64))
    assert lib.gc_correlator_carrier_wipeoff_multicorrelator_resampler_5(c._h, 0.0, 0.0, 0.0, 0.1, 100) == gnsscorr.GC_ERR_STATE
    c.close()
    b = gnsscorr.TrackingBatch(gctx, 1, 3, 1023, high_dyn=True)
    with pytest.raises(gnsscorr.GnsscorrError):
        b.set_16sc(True)
    b.close()
    b = gnsscorr.TrackingBatch(gctx, 1, 3, 1023)
    b.set_16sc(True)
    with pytest.raises(gnsscorr.GnsscorrError):
        b.set_input_format(gnsscorr.GC_IQ_F32)
    with pytest.raises(gnsscorr.GnsscorrError):
        b.set_code(0, np.ones(1023, np.float32), np.zeros(3, np.float32))
    b.set_16sc(False)  # back to the float correlator: float codes and any input format are accepted again
    b.set_code(0, np.ones(1023, np.float32), np.zeros(3, np.float32))
    b.set_input_format(gnsscorr.GC_IQ_I8)
    b.close()
