"""Experiments build only (collected by tests/test_experiments_gpu.py with GNSSCORR_LIB = libgnsscorr_exp.so): closed-loop code
periods cut into S workgroups, one launch per period (trk_closed_loop_slice_kernel)."""
import os

import numpy as np
import pytest

from test_closed_loop_gpu import GPS, _conf, _signal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("signal", ["gps", "galileo_pilot"])
def test_closed_loop_sliced_periods(gctx, oracle, signal):
    """Few channels on a big chip: a channel-period is cut into S workgroups, one launch per code period, the last slice to
    finish adds the partial sums in slice order and runs the loop maths (trk_closed_loop_slice_kernel;
    dll_pll_veml_tracking.cc:914-1070 on one lane as before).  Against the persistent one-workgroup-per-channel kernel (S = 1): the
    same block boundaries every period, correlator sums equal to float rounding (they are associated differently), Doppler within
    0.02 Hz.  For a given S: bit-identical records run after run, and whether 80 periods come from one call or from 2 x 40; an
    exhausted input gives the same invalid tail; a stopped channel gives standby records."""
    import gnsscorr
    import torch
    if signal == "gps":
        fs, n_ep, n_ch, L = 4e6, 80, 3, 1023
        code, x = _signal(oracle, 21, fs, 4000 * (n_ep + 3), 909, 2345.0, 777.0)
        conf = dict(GPS, acq_delay_samples=777.0, acq_doppler_hz=2350.0, acq_samplestamp_samples=0, sample_counter=0)
        sync, data_code = None, None
    else:
        G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
        z = np.load(os.path.join(G, "galileo_e1_codes.npz"))
        code = oracle.galileo_e1_sinboc11(z["e1c"][6])
        data_code = oracle.galileo_e1_sinboc11(z["e1b"][6])
        fs, n, n_ep, n_ch, L = 4e6, 16000, 40, 2, 8184
        rng = np.random.Generator(np.random.PCG64(4711))
        i = np.arange(n * (n_ep + 3))
        rate = 2.046e6 * (1 + 1000.0 / 1575.42e6) / fs
        idx = np.floor((8184.0 - 3000.0 * 2.046e6 / fs) + i * rate).astype(np.int64) % 8184
        amp = np.sqrt(10 ** 4.7 / fs)
        x = (amp * (code[idx] + data_code[idx]) * np.exp(1j * (2 * np.pi * 1000.0 * i / fs + 0.3))
            + (rng.standard_normal(i.size) + 1j * rng.standard_normal(i.size)) * np.sqrt(0.5)).astype(np.complex64)
        conf = dict(GPS, code_period_s=0.004, code_length_chips=4092, code_samples_per_chip=2, vector_length=n, veml=1, pll_bw_hz=15.0, dll_bw_hz=0.75,
            fll_bw_hz=10.0, early_late_space_chips=0.15, very_early_late_space_chips=0.6, acq_delay_samples=3000.0, acq_doppler_hz=1010.0,
            acq_samplestamp_samples=0, sample_counter=0)
        sync = gnsscorr.LoopSyncConf.make(extend_correlation_symbols=1, track_pilot=True, symbols_per_bit=1)
    d = torch.from_numpy(x.view(np.float32)).cuda()

    def make(slices, n_used=None):
        loop = gnsscorr.TrackingLoop(gctx, n_ch, L)
        loop.set_geometry(slices_per_channel=slices)
        for ch in range(n_ch):
            loop.set_input_dev(ch, d.data_ptr(), x.size if n_used is None else n_used)
            if sync is not None:
                loop.set_sync(ch, sync, data_code)
            if ch != 1:  # channel 1 is never started: standby records
                loop.start(ch, _conf(gnsscorr, **conf), code)
        return loop
    base_loop = make(1)
    base = base_loop.run(n_ep)
    base_loop.close()
    assert np.all(base["valid"][0] == 1) and np.all(base["valid"][1] == 0) and np.all(base["state"][1] == 0)
    for S in (2, 4, 8):
        a = make(S)
        one = a.run(n_ep)
        a.close()
        b = make(S)
        two = np.concatenate([b.run(n_ep // 2), b.run(n_ep - n_ep // 2)], axis=1)
        b.close()
        assert one.tobytes() == two.tobytes(), S   # reproducible, and independent of how the periods are split over calls
        assert np.array_equal(one["sample_counter"], base["sample_counter"]), S
        assert np.array_equal(one["current_prn_length_samples"], base["current_prn_length_samples"]) and np.array_equal(one["state"], base["state"])
        assert np.array_equal(one["valid"], base["valid"])
        assert np.max(np.abs(one["corr"] - base["corr"])) <= 2e-5 * np.max(np.abs(base["corr"])), S
        assert np.max(np.abs(one["carrier_doppler_hz"] - base["carrier_doppler_hz"])) < 0.02, S
        assert np.max(np.abs(one["prompt_data"] - base["prompt_data"])) <= 2e-5 * np.max(np.abs(base["corr"]))
    # an input that ends early: the tail of the launch finds no samples, sliced like persistent
    short = int(x.size * 0.6)
    p1, p8 = make(1, short), make(8, short)
    t1, t8 = p1.run(n_ep), p8.run(n_ep)
    p1.close()
    p8.close()
    assert np.array_equal(t1["valid"], t8["valid"]) and np.array_equal(t1["sample_counter"], t8["sample_counter"]) and np.any(t1["valid"][0] == 0)
