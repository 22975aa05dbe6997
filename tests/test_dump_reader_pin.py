"""CPU test (row f2): the 96-byte tracking dump record is pinned by the REFERENCE'S OWN reader.

tests/golden/track_dump_gps_l1_ch0.dat is a dump this repository's `hip_dll_pll_veml_tracking` wrote on an MI355X;
tests/golden/ref_track_dump_fields.npz holds what `Tracking_Dump_Reader::read_binary_obs()` -- the reference's
src/tests/unit-tests/signal-processing-blocks/libs/tracking_dump_reader.cc, compiled from where it lies into oracle/_ref/libref_dump.so
by tests/golden/make_golden_dump.py in the build container -- returned for that file.  The numpy record dtype the GPU tests check fresh
dumps with (tests/test_adapter_gpu.py) must read the same values from the same bytes, field for field, and the same record count
(dll_pll_veml_tracking.cc:1196-1243 writes the record)."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# the dtype of tests/test_adapter_gpu.py::test_cpp_closed_loop_tracking_selftest
DUMP_RECORD = np.dtype([("abs_VE", "<f4"), ("abs_E", "<f4"), ("abs_P", "<f4"), ("abs_L", "<f4"), ("abs_VL", "<f4"),
    ("prompt_I", "<f4"), ("prompt_Q", "<f4"), ("PRN_start_sample_count", "<u8"), ("acc_carrier_phase_rad", "<f4"),
    ("carrier_doppler_hz", "<f4"), ("carrier_doppler_rate_hz_s", "<f4"), ("code_freq_chips", "<f4"),
    ("code_freq_rate_chips", "<f4"), ("carr_error_hz", "<f4"), ("carr_error_filt_hz", "<f4"), ("code_error_chips", "<f4"),
    ("code_error_filt_chips", "<f4"), ("CN0_SNV_dB_Hz", "<f4"), ("carrier_lock_test", "<f4"), ("aux1", "<f4"),
    ("aux2", "<f8"), ("PRN", "<u4")])


def test_numpy_dtype_reads_what_the_reference_reader_reads():
    ref = np.load(os.path.join(G, "ref_track_dump_fields.npz"))
    dump = np.fromfile(os.path.join(G, "track_dump_gps_l1_ch0.dat"), DUMP_RECORD)
    assert DUMP_RECORD.itemsize == 96
    assert os.path.getsize(os.path.join(G, "track_dump_gps_l1_ch0.dat")) == 96 * dump.size
    assert int(ref["num_epochs"]) == dump.size and dump.size >= 200  # Tracking_Dump_Reader::num_epochs(): size / 96
    names = [n for n in DUMP_RECORD.names]
    assert sorted(names) == sorted(k for k in ref.files if k != "num_epochs")
    for name in names:
        a, b = dump[name], ref[name]
        assert a.dtype == b.dtype, name
        assert a.tobytes() == b.tobytes(), name  # bit for bit (NaN-safe)
    # the file is a real tracking run, not zeros: PRN 1, Doppler pulling in towards the signal's 1680 Hz, increasing sample stamps
    assert np.all(dump["PRN"] == 1) and np.all(np.diff(dump["PRN_start_sample_count"].astype(np.int64)) >= 3999)
    assert abs(float(dump["carrier_doppler_hz"][-20:].mean()) - 1680.0) < 100.0 and np.all(dump["abs_P"] > 0)  # the first 256 ms: still pulling in


def test_the_reference_reader_itself_when_it_is_built():
    """In the build container (oracle/_ref present) the compiled reference reader is run again on the committed file."""
    import sys
    lib = os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "libref_dump.so")
    if not os.path.exists(lib):
        import pytest
        pytest.skip("oracle/_ref/libref_dump.so is built from /root/reference in the build container only")
    sys.path.insert(0, G)
    from make_golden_dump import read_with_reference_reader
    got = read_with_reference_reader(os.path.join(G, "track_dump_gps_l1_ch0.dat"))
    ref = np.load(os.path.join(G, "ref_track_dump_fields.npz"))
    for k in ref.files:
        assert np.asarray(got[k]).tobytes() == ref[k].tobytes(), k
