#!/usr/bin/env python3
"""Generates tests/golden/ref_loop_filters.npz: input / output sequences of the reference's loop filters that compile
from their own sources (oracle/_ref/libref_loop.so, `make -C oracle ref`; BUILD container only):

  Tracking_FLL_PLL_filter   src/algorithms/tracking/libs/tracking_FLL_PLL_filter.cc:55-133 -- the carrier loop filter
                            of dll_pll_veml_tracking (dll_pll_veml_tracking.cc:344, :564, :939-950, :1752)
  Tracking_2nd_DLL_filter   tracking_2nd_DLL_filter.cc:40-96   } code / carrier filters of the GLONASS and
  Tracking_2nd_PLL_filter   tracking_2nd_PLL_filter.cc:40-104  } carrier-aided blocks

Every case is a script of the calls the tracking block makes: set_params -> initialize -> get_carrier_error x n with the
block's three call shapes (FLL only during pull-in, FLL-aided PLL, PLL only), the correlation time changing where the block
extends its integration, and a second set_params where it narrows the bandwidth (:1752) WITHOUT re-initialising.  Inputs are
seeded float32 values of the magnitudes the discriminators produce; outputs are what the compiled reference returned.
The file holds arrays only (inputs, outputs, the scripts' scalars)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_loop.so"))
    L.ref_fll_pll_new.restype = C.c_void_p
    L.ref_fll_pll_delete.argtypes = [C.c_void_p]
    L.ref_fll_pll_set_params.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int]
    L.ref_fll_pll_initialize.argtypes = [C.c_void_p, C.c_float]
    L.ref_fll_pll_get_carrier_error.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    L.ref_fll_pll_get_carrier_error.restype = C.c_float
    for n in ("dll2", "pll2"):
        getattr(L, "ref_%s_new" % n).restype = C.c_void_p
        getattr(L, "ref_%s_new" % n).argtypes = [C.c_float]
        getattr(L, "ref_%s_delete" % n).argtypes = [C.c_void_p]
        getattr(L, "ref_%s_set_bw" % n).argtypes = [C.c_void_p, C.c_float]
        getattr(L, "ref_%s_set_pdi" % n).argtypes = [C.c_void_p, C.c_float]
        getattr(L, "ref_%s_initialize" % n).argtypes = [C.c_void_p]
    L.ref_dll2_get_code_nco.argtypes = [C.c_void_p, C.c_float]
    L.ref_dll2_get_code_nco.restype = C.c_float
    L.ref_pll2_get_carrier_nco.argtypes = [C.c_void_p, C.c_float]
    L.ref_pll2_get_carrier_nco.restype = C.c_float
    return L


def fll_pll_script(rng, n):
    """(fll, pll, T) per step, float32, in the block's three call shapes; a Doppler-like drift in the discriminators."""
    fll = np.zeros(n, np.float32)
    pll = np.zeros(n, np.float32)
    a, b = n // 4, n // 2
    fll[:a] = (rng.standard_normal(a) * 25.0 + np.linspace(40.0, 0.0, a)).astype(np.float32)          # pull-in: pure FLL (:939)
    fll[a:b] = (rng.standard_normal(b - a) * 6.0).astype(np.float32)                                  # FLL-aided PLL (:944)
    pll[a:] = (rng.standard_normal(n - a) * 0.03 + 0.02 * np.sin(np.arange(n - a) / 17.0)).astype(np.float32)  # [Hz]: atan / 2 pi
    return fll, pll


def main():
    L = load()
    rng = np.random.Generator(np.random.PCG64(20261004))
    out = {}
    cases = []
    # (order, fll_bw, pll_bw, pll_bw_narrow, T wide, T narrow, acquisition Doppler): the bandwidths / periods of tests/ and bench.py
    for order in (1, 2, 3):
        for fll_bw, pll_bw, narrow, t_wide, t_narrow, dopp in (
                (35.0, 40.0, 20.0, 0.001, 0.020, 1690.0),
                (10.0, 15.0, 10.0, 0.004, 0.012, -632.0),
                (10.0, 25.0, 15.0, 0.001, 0.010, 4375.5),
                (35.0, 2.0, 0.0, 0.001, 0.001, -4987.25),
                (0.0, 40.0, 10.0, 0.001, 0.005, 12.0)):
            cases.append((order, fll_bw, pll_bw, narrow, t_wide, t_narrow, dopp))
    n = 256
    for i, (order, fll_bw, pll_bw, narrow, t_wide, t_narrow, dopp) in enumerate(cases):
        fll, pll = fll_pll_script(rng, n)
        T = np.full(n, t_wide, np.float32)
        n_switch = 3 * n // 4  # narrow stage: set_params with the narrow bandwidth, longer correlation time, no initialize
        T[n_switch:] = np.float32(t_narrow)
        f = L.ref_fll_pll_new()
        L.ref_fll_pll_set_params(f, fll_bw, pll_bw, order)
        L.ref_fll_pll_initialize(f, dopp)
        y = np.zeros(n, np.float32)
        for k in range(n):
            if k == n_switch:
                L.ref_fll_pll_set_params(f, fll_bw, narrow, order)
            y[k] = L.ref_fll_pll_get_carrier_error(f, float(fll[k]), float(pll[k]), float(T[k]))
        L.ref_fll_pll_delete(f)
        out["fp%d_conf" % i] = np.array([order, fll_bw, pll_bw, narrow, dopp, n_switch], np.float64)
        out["fp%d_fll" % i], out["fp%d_pll" % i], out["fp%d_T" % i], out["fp%d_out" % i] = fll, pll, T, y
    out["n_fll_pll"] = np.int64(len(cases))
    # second-order filters: (bw, pdi, narrow bw, extended pdi)
    c2 = [(2.0, 0.001, 1.0, 0.010), (4.0, 0.001, 2.0, 0.005), (0.75, 0.004, 0.5, 0.004), (50.0, 0.001, 25.0, 0.001), (35.0, 0.001, 15.0, 0.002)]
    for i, (bw, pdi, bw2, pdi2) in enumerate(c2):
        e = (rng.standard_normal(n) * 0.05 + 0.03 * np.cos(np.arange(n) / 23.0)).astype(np.float32)
        for name, new, get in (("dll2", L.ref_dll2_new, L.ref_dll2_get_code_nco), ("pll2", L.ref_pll2_new, L.ref_pll2_get_carrier_nco)):
            f = new(pdi)
            getattr(L, "ref_%s_set_bw" % name)(f, bw)
            getattr(L, "ref_%s_initialize" % name)(f)
            y = np.zeros(n, np.float32)
            for k in range(n):
                if k == n // 2:  # the carrier-aided block's extended integration: new pdi and bandwidth, state kept
                    getattr(L, "ref_%s_set_pdi" % name)(f, pdi2)
                    getattr(L, "ref_%s_set_bw" % name)(f, bw2)
                y[k] = get(f, float(e[k]))
            getattr(L, "ref_%s_delete" % name)(f)
            out["%s_%d_out" % (name, i)] = y
        out["s2_%d_conf" % i] = np.array([bw, pdi, bw2, pdi2], np.float64)
        out["s2_%d_in" % i] = e
    out["n_second_order"] = np.int64(len(c2))
    path = os.path.join(HERE, "ref_loop_filters.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(cases), "FLL/PLL scripts,", len(c2), "second-order scripts")


if __name__ == "__main__":
    main()
