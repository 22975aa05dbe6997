"""CPU tests (no GPU): the loop filters of row f1 against the REFERENCE's own filters compiled in oracle/_ref.

tests/golden/ref_loop_filters.npz (tests/golden/make_golden_loop.py, build container) holds scripted call sequences and the
outputs of the reference's Tracking_FLL_PLL_filter (tracking_FLL_PLL_filter.cc:55-133: the carrier loop filter of
dll_pll_veml_tracking, dll_pll_veml_tracking.cc:914-973), Tracking_2nd_DLL_filter and Tracking_2nd_PLL_filter.  Held to them
BIT FOR BIT (float32 bit patterns, 256 steps per script, bandwidth narrowed and correlation time changed mid-script):
  * tests/closed_loop_ref.py::Pll -- the checker of the device loop, so the GPU parity of carrier_doppler_hz rests on a pinned filter;
  * adapter/tracking_loop_maths.h::Tracking_FLL_PLL_filter (the host-loop image) and the second-order filters of
    adapter/hip_glonass_ca_dll_pll_tracking.h, compiled here with g++ -ffp-contract=off like the reference was.
The device loop's own copy (csrc/trk_closed_loop.hip::pll_get_carrier_error) is checked on the GPU against the pinned Pll fed
with the device's recorded discriminator outputs (tests/test_closed_loop_gpu.py::test_device_carrier_filter_against_the_pinned_filter)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
fp = C.POINTER(C.c_float)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_python_restatement_of_the_carrier_filter_is_bit_exact():
    from closed_loop_ref import Pll
    z = np.load(os.path.join(G, "ref_loop_filters.npz"))
    assert int(z["n_fll_pll"]) == 15
    for i in range(int(z["n_fll_pll"])):
        order, fll_bw, pll_bw, narrow, dopp, n_switch = z["fp%d_conf" % i]
        f = Pll(float(fll_bw), float(pll_bw), int(order))
        f.initialize(np.float32(dopp))
        fll, pll, T, want = z["fp%d_fll" % i], z["fp%d_pll" % i], z["fp%d_T" % i], z["fp%d_out" % i]
        got = np.zeros(len(want), np.float32)
        for k in range(len(want)):
            if k == int(n_switch):
                f.set_params(float(fll_bw), float(narrow), int(order))
            got[k] = f.get_carrier_error(fll[k], pll[k], T[k])
        assert np.array_equal(_bits(got), _bits(want)), ("script %d (order %d)" % (i, order), np.flatnonzero(_bits(got) != _bits(want))[:5])


@pytest.fixture(scope="module")
def capi(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("capi") / "libloopcapi.so")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-DWITH_SECOND_ORDER",
        "-I", os.path.join(ROOT, "gnss-sdr-1_amd", "adapter"), "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "loop_filter_capi.cpp"), "-o", so,
        "-L", os.path.join(ROOT, "gnss-sdr-1_amd"), "-lgnsscorr", "-Wl,-rpath," + os.path.join(ROOT, "gnss-sdr-1_amd")])
    L = C.CDLL(so)
    L.fp_run.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, fp, fp, fp, fp]
    L.s2_run.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, fp, fp]
    return L


def test_adapter_carrier_filter_is_bit_exact(capi):
    z = np.load(os.path.join(G, "ref_loop_filters.npz"))
    for i in range(int(z["n_fll_pll"])):
        order, fll_bw, pll_bw, narrow, dopp, n_switch = z["fp%d_conf" % i]
        fll, pll, T, want = (np.ascontiguousarray(z["fp%d_%s" % (i, k)], np.float32) for k in ("fll", "pll", "T", "out"))
        got = np.zeros_like(want)
        capi.fp_run(int(order), fll_bw, pll_bw, narrow, dopp, int(n_switch), len(want), fll.ctypes.data_as(fp), pll.ctypes.data_as(fp),
            T.ctypes.data_as(fp), got.ctypes.data_as(fp))
        assert np.array_equal(_bits(got), _bits(want)), "script %d (order %d)" % (i, order)


def test_adapter_second_order_filters_are_bit_exact(capi):
    z = np.load(os.path.join(G, "ref_loop_filters.npz"))
    for i in range(int(z["n_second_order"])):
        bw, pdi, bw2, pdi2 = z["s2_%d_conf" % i]
        e = np.ascontiguousarray(z["s2_%d_in" % i], np.float32)
        for which, name in ((0, "dll2"), (1, "pll2")):
            want = z["%s_%d_out" % (name, i)]
            got = np.zeros_like(want)
            capi.s2_run(which, bw, pdi, bw2, pdi2, len(e), e.ctypes.data_as(fp), got.ctypes.data_as(fp))
            assert np.array_equal(_bits(got), _bits(want)), "%s script %d" % (name, i)
