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
