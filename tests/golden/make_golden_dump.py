#!/usr/bin/env python3
"""Generates tests/golden/ref_track_dump_fields.npz: what the REFERENCE'S OWN reader returns for a tracking dump written by this
repository's block image (BUILD container only: needs oracle/_ref/libref_dump.so = the reference's
src/tests/unit-tests/signal-processing-blocks/libs/tracking_dump_reader.cc compiled from where it lies, `make -C oracle ref`).

  tests/golden/track_dump_gps_l1_ch0.dat   the first 256 records (96 bytes each) of the dump `hip_dll_pll_veml_tracking` wrote on an
                                           MI355X while tracking PRN 1 of the synthetic 4 Msps signal of tracking_selftest.cpp
                                           (gpurun: GNSSCORR_SELFTEST_DUMP_DIR=... tracking_selftest; record layout of
                                           dll_pll_veml_tracking.cc:1196-1243).  A data file this repository's code produced.
  ref_track_dump_fields.npz                per field, the values Tracking_Dump_Reader::read_binary_obs() put into its members for that
                                           file, record by record, and num_epochs().  Arrays only.

tests/test_dump_reader_pin.py (CPU) then holds the repository's numpy record dtype -- the checker of tests/test_adapter_gpu.py -- to these
arrays field for field: the dump format is pinned by the reference's reader, not by a dtype typed from reading it."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
F32_FIELDS = ["abs_VE", "abs_E", "abs_P", "abs_L", "abs_VL", "prompt_I", "prompt_Q", "acc_carrier_phase_rad", "carrier_doppler_hz",
    "carrier_doppler_rate_hz_s", "code_freq_chips", "code_freq_rate_chips", "carr_error_hz", "carr_error_filt_hz", "code_error_chips",
    "code_error_filt_chips", "CN0_SNV_dB_Hz", "carrier_lock_test", "aux1"]


def read_with_reference_reader(path, max_records=1 << 20):
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_dump.so"))
    L.ref_dump_num_epochs.restype = C.c_longlong
    L.ref_dump_num_epochs.argtypes = [C.c_char_p]
    L.ref_dump_read.restype = C.c_longlong
    L.ref_dump_read.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    n_ep = L.ref_dump_num_epochs(path.encode())
    assert n_ep >= 0, "the reference reader cannot open %s" % path
    n = min(n_ep, max_records)
    f32 = np.zeros((n, 20), np.float32)
    u64 = np.zeros(n, np.uint64)
    f64 = np.zeros(n, np.float64)
    u32 = np.zeros(n, np.uint32)
    got = L.ref_dump_read(path.encode(), n, f32.ctypes.data, u64.ctypes.data, f64.ctypes.data, u32.ctypes.data)
    assert got == n, (got, n)
    out = {name: f32[:, i].copy() for i, name in enumerate(F32_FIELDS)}
    out["PRN_start_sample_count"] = u64
    out["aux2"] = f64
    out["PRN"] = u32
    out["num_epochs"] = np.array(n_ep, np.int64)
    return out


if __name__ == "__main__":
    dat = os.path.join(HERE, "track_dump_gps_l1_ch0.dat")
    fields = read_with_reference_reader(dat)
    np.savez_compressed(os.path.join(HERE, "ref_track_dump_fields.npz"), **fields)
    print("records", int(fields["num_epochs"]), "PRN", set(fields["PRN"].tolist()), "doppler tail", fields["carrier_doppler_hz"][-3:])
