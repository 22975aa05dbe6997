"""CPU tests (no GPU): the product's own PRN generators (gc_codes.cpp, host side of the C ABI)
against golden vectors produced by the REFERENCE's generators compiled into oracle/_ref
(tests/golden/ref_codes.npz) and against the Galileo ICD data."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_gps_l1_ca_chips_and_sampled_codes_match_reference():
    import gnsscorr
    z = np.load(os.path.join(G, "ref_codes.npz"))
    for k, prn in enumerate(z["gps_prn"]):
        assert np.array_equal(gnsscorr.gps_l1_ca_code_gen_float(int(prn)), z["gps_chips"][k].astype(np.float32))
    assert np.array_equal(gnsscorr.gps_l1_ca_code_gen_float(5, 7), z["gps_chips_shift7"].astype(np.float32))
    for fs in (4000000, 25000000, 2048000):
        for prn in (1, 19):
            got = gnsscorr.gps_l1_ca_code_gen_complex_sampled(prn, fs)
            assert got.size == fs // 1000 and np.all(got.imag == 0)
            assert np.array_equal(got.real, z["gps_sampled_fs%d_prn%d" % (fs, prn)].astype(np.float32))
    with pytest.raises(gnsscorr.GnsscorrError):
        gnsscorr.gps_l1_ca_code_gen_float(0)


def test_beidou_b1i_chips_and_sampled_codes_match_reference():
    import gnsscorr
    z = np.load(os.path.join(G, "ref_codes.npz"))
    for k, prn in enumerate(z["bds_prn"]):
        assert np.array_equal(gnsscorr.beidou_b1i_code_gen_float(int(prn)), z["bds_chips"][k].astype(np.float32))
    for fs in (4000000, 25000000, 2048000):
        got = gnsscorr.beidou_b1i_code_gen_complex_sampled(6, fs)
        assert np.array_equal(got.real, z["bds_sampled_fs%d_prn6" % fs].astype(np.float32))
    with pytest.raises(gnsscorr.GnsscorrError):
        gnsscorr.beidou_b1i_code_gen_float(0)


def test_glonass_l1_ca_chips_and_sampled_codes_match_reference():
    """One 511-chip m-sequence for every GLONASS satellite (FDMA): glonass_l1_signal_processing.cc compiled into oracle/_ref."""
    import gnsscorr
    z = np.load(os.path.join(G, "ref_codes.npz"))
    chips = gnsscorr.glonass_l1_ca_code_gen_float()
    assert np.array_equal(chips, z["glo_chips"].astype(np.float32))
    assert np.array_equal(gnsscorr.glonass_l1_ca_code_gen_float(100), z["glo_chips_shift100"].astype(np.float32))
    assert int(chips.sum()) == 1  # maximal-length sequence: 256 ones, 255 zeros
    for fs in (4000000, 25000000, 2048000):
        got = gnsscorr.glonass_l1_ca_code_gen_complex_sampled(fs)
        assert got.size == fs // 1000 and np.all(got.imag == 0)
        assert np.array_equal(got.real, z["glo_sampled_fs%d" % fs].astype(np.float32))


def test_galileo_e1_generators(oracle):
    import gnsscorr
    z = np.load(os.path.join(G, "galileo_e1_codes.npz"))
    for sig, tab in (("1B", z["e1b"]), ("1C", z["e1c"])):
        for prn in (1, 11, 50):
            s = gnsscorr.galileo_e1_code_gen_sinboc11_float(sig, prn)
            assert np.array_equal(s[0::2], tab[prn - 1].astype(np.float32))
            assert np.array_equal(s[1::2], -tab[prn - 1].astype(np.float32))
    # sampled replicas agree with the oracle's restatement of galileo_e1_code_gen_float_sampled
    for fs in (4000000, 25000000, 2046000):
        for cboc in (False, True):
            got = gnsscorr.galileo_e1_code_gen_complex_sampled("1B", cboc, 1, fs)
            want = oracle.galileo_e1_code_sampled(z["e1b"][0], fs, cboc=cboc)
            assert got.size == want.size == int(fs * 0.004)
            assert np.array_equal(got.real, want) and np.all(got.imag == 0)
    got = gnsscorr.galileo_e1_code_gen_complex_sampled("1C", True, 7, 4000000, chip_shift=100)
    want = oracle.galileo_e1_code_sampled(z["e1c"][6], 4000000, cboc=True, is_e1c=True, chip_shift=100)
    assert np.array_equal(got.real, want)
    with pytest.raises(gnsscorr.GnsscorrError):
        gnsscorr.galileo_e1_code_gen_sinboc11_float("5X", 1)
    with pytest.raises(gnsscorr.GnsscorrError):
        gnsscorr.galileo_e1_code_gen_sinboc11_float("1B", 51)


# ---- the 10.23 / 0.5115 Mcps signals: GPS L2C M, GPS L5 I/Q, BeiDou B3I (fixtures from the reference's own generators
# compiled here, tests/golden/make_golden_wideband.py) and Galileo E5a (ICD memory codes) ----
def _wb():
    return np.load(os.path.join(G, "ref_codes_wideband.npz"))


def _checksum(c):
    w = (np.arange(10230, dtype=np.int64) * 2654435761 + 12345) % 1000003
    return np.int64(np.sum(((1 - c.astype(np.int64)) // 2) * w))


def test_gps_l2c_l5_and_beidou_b3i_codes_match_the_reference_generators():
    import gnsscorr
    z = _wb()
    for k, prn in enumerate(z["prns"]):
        prn = int(prn)
        assert np.array_equal(gnsscorr.gps_l2c_m_code_gen_float(prn).astype(np.int8), z["l2c_chips"][k]), prn
        assert np.array_equal(gnsscorr.gps_l5i_code_gen_float(prn).astype(np.int8), z["l5i_chips"][k]), prn
        assert np.array_equal(gnsscorr.gps_l5q_code_gen_float(prn).astype(np.int8), z["l5q_chips"][k]), prn
        assert np.array_equal(gnsscorr.beidou_b3i_code_gen_float(prn).astype(np.int8), z["b3i_chips"][k]), prn
    assert np.array_equal(gnsscorr.beidou_b3i_code_gen_float(9, 1234).astype(np.int8), z["b3i_chips_prn9_shift1234"])
    # every PRN, through a position-weighted checksum of the reference's code
    for prn in range(1, 51):
        assert _checksum(gnsscorr.gps_l2c_m_code_gen_float(prn)) == z["l2c_checksums"][prn - 1], prn
        assert _checksum(gnsscorr.gps_l5i_code_gen_float(prn)) == z["l5i_checksums"][prn - 1], prn
        assert _checksum(gnsscorr.gps_l5q_code_gen_float(prn)) == z["l5q_checksums"][prn - 1], prn
    for prn in range(1, 64):
        # the reference's B3I chips are +1 for a set bit: the checksum was taken on its int code
        assert _checksum(gnsscorr.beidou_b3i_code_gen_float(prn)) == z["b3i_checksums"][prn - 1], prn
    for bad in (0, 51):
        with pytest.raises(gnsscorr.GnsscorrError):
            gnsscorr.gps_l5i_code_gen_float(bad)
    with pytest.raises(gnsscorr.GnsscorrError):
        gnsscorr.beidou_b3i_code_gen_float(64)


def test_wideband_sampled_codes_match_the_reference_generators():
    """L2C / L5 digitise with a true ceil(), B3I with the (int)(x + 1) of the C/A generators: both are pinned."""
    import gnsscorr
    z = _wb()
    for fs in (12500000, 25000000, 10230000, 4000000):
        for name, fn, prn in (("l5i", gnsscorr.gps_l5i_code_gen_complex_sampled, 7), ("l5q", gnsscorr.gps_l5q_code_gen_complex_sampled, 7),
                ("b3i", gnsscorr.beidou_b3i_code_gen_complex_sampled, 19)):
            want = z["%s_sampled_fs%d_prn%d" % (name, fs, prn)]
            got = fn(prn, fs)
            assert got.size == want.size == fs // 1000 and np.all(got.imag == 0)
            assert np.array_equal(got.real.astype(np.int8), want), (name, fs)
    for fs in (4000000, 25000000, 2046000):
        want = z["l2c_sampled_fs%d_prn19" % fs]
        got = gnsscorr.gps_l2c_m_code_gen_complex_sampled(19, fs)
        assert got.size == want.size == fs // 50
        assert np.array_equal(got.real.astype(np.int8), want), fs


def test_galileo_e5a_codes_are_the_icd_memory_codes():
    """galileo_e5_signal_processing.cc does not compile here (GNU Radio header): pinned by the ICD data the reference holds."""
    import gnsscorr
    z = _wb()
    for prn in (1, 11, 36, 50):
        x = gnsscorr.galileo_e5_a_code_gen_complex_primary(prn, "5X")
        assert np.array_equal(x.real.astype(np.int8), z["e5a_i_chips"][prn - 1]) and np.array_equal(x.imag.astype(np.int8), z["e5a_q_chips"][prn - 1])
        i_only, q_only = gnsscorr.galileo_e5_a_code_gen_complex_primary(prn, "5I"), gnsscorr.galileo_e5_a_code_gen_complex_primary(prn, "5Q")
        assert np.array_equal(i_only, x.real.astype(np.complex64)) and np.array_equal(q_only, 1j * x.imag.astype(np.complex64))
    # at the chip rate the sampled code is the code itself, delayed by the chip shift
    s = gnsscorr.galileo_e5_a_code_gen_complex_sampled("5X", 11, 10230000, 100)
    assert np.array_equal(np.roll(gnsscorr.galileo_e5_a_code_gen_complex_primary(11, "5X"), 10230 - 100), s)
    # resampled: the nearest-neighbour resampler of gnss_signal_processing.cc picks whole complex chips
    s25 = gnsscorr.galileo_e5_a_code_gen_complex_sampled("5X", 11, 25000000)
    x = gnsscorr.galileo_e5_a_code_gen_complex_primary(11, "5X")
    assert s25.size == 25000 and s25[0] == x[0] and s25[-1] == x[-1] and set(np.unique(np.abs(s25.real))) == {1.0}
    idx = np.minimum(np.ceil(np.float32(1 / np.float32(25e6)) * np.arange(1, 25000, dtype=np.float32) / np.float32(1 / np.float32(10.23e6))).astype(np.int64) - 1, 10229)
    assert np.mean(s25[:-1] == x[idx]) > 0.999  # float32 index rule: all but chip-edge samples follow the plain formula
    # secondary codes
    assert gnsscorr.secondary_code("5I") == "".join(str(int(b)) for b in z["e5a_i_secondary"])
    assert gnsscorr.secondary_code("5Q", 47) == "".join(str(int(b)) for b in z["e5a_q_secondary"][46])
    with pytest.raises(gnsscorr.GnsscorrError, match="no E5a-Q secondary code"):
        gnsscorr.secondary_code("5Q", 48)
    assert gnsscorr.secondary_code("1C") == "0011100000001010110110010" and gnsscorr.secondary_code("L5I") == "0000110101"
    assert gnsscorr.secondary_code("B3") == gnsscorr.secondary_code("L5Q") == "00000100110101001110"


def test_loop_sync_for_signal_mirrors_the_block_constructor():
    """gc_loop_sync_for_signal: what dll_pll_veml_tracking's constructor / start_tracking derive per signal (:113-336, :631-705)."""
    import gnsscorr
    y = gnsscorr.loop_sync_for_signal("G", "1C", 7, track_pilot=True, extend_correlation_symbols=20)
    assert (y.symbols_per_bit, y.secondary_code_length, y.preamble_length_symbols, y.track_pilot, y.extend_correlation_symbols) == (20, 0, 160, 0, 20)
    pre = list(y.preamble_symbols)[:160]
    assert pre == [s for b in (1, 0, 0, 0, 1, 0, 1, 1) for s in [1 if b else -1] * 20] and y.bit_sync_min_time_s == 10.0
    y = gnsscorr.loop_sync_for_signal("G", "2S", 3)
    assert (y.symbols_per_bit, y.secondary_code_length, y.preamble_length_symbols) == (1, 0, 0)
    y = gnsscorr.loop_sync_for_signal("G", "L5", 3, track_pilot=True)
    assert (y.symbols_per_bit, y.track_pilot, y.secondary_code.decode()) == (10, 1, "00000100110101001110")
    assert gnsscorr.loop_sync_for_signal("G", "L5", 3).secondary_code.decode() == "0000110101"
    y = gnsscorr.loop_sync_for_signal("E", "1B", 11, track_pilot=True, extend_correlation_symbols=4)
    assert (y.symbols_per_bit, y.track_pilot, y.secondary_code_length, y.secondary_code.decode()) == (1, 1, 25, "0011100000001010110110010")
    assert gnsscorr.loop_sync_for_signal("E", "1B", 11).secondary_code_length == 0
    y = gnsscorr.loop_sync_for_signal("E", "5X", 12, track_pilot=True)
    assert y.secondary_code_length == 100 and y.secondary_code.decode() == gnsscorr.secondary_code("5Q", 12)
    assert gnsscorr.loop_sync_for_signal("E", "5X", 12).secondary_code_length == 0  # left to the telemetry decoder
    for sig in ("B1", "B3"):
        y = gnsscorr.loop_sync_for_signal("C", sig, 20)
        assert (y.symbols_per_bit, y.secondary_code.decode(), y.preamble_length_symbols) == (20, "00000100110101001110", 0)
        geo = gnsscorr.loop_sync_for_signal("C", sig, 3)
        assert (geo.symbols_per_bit, geo.secondary_code_length, geo.preamble_length_symbols) == (2, 0, 22)
        assert list(geo.preamble_symbols)[:22] == [s for b in (1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 0) for s in [1 if b else -1] * 2]
    with pytest.raises(gnsscorr.GnsscorrError, match="unknown system"):
        gnsscorr.loop_sync_for_signal("R", "1G", 1)
    with pytest.raises(gnsscorr.GnsscorrError, match="no E5a-Q secondary code"):
        gnsscorr.loop_sync_for_signal("E", "5X", 49, track_pilot=True)
