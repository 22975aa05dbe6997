"""GPU parity of the closed-loop tracking engine (correlations + DLL/PLL maths of dll_pll_veml_tracking in one
launch) against the Python restatement of the same loop running on the CPU oracle correlator
(tests/closed_loop_ref.py)."""
import os

import numpy as np
import pytest

from helpers import synth_stream

pytestmark = pytest.mark.gpu


def _conf(gnsscorr, **kw):
    c = gnsscorr.LoopConf()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


GPS = dict(fs_in=4e6, signal_carrier_freq_hz=1575.42e6, code_chip_rate_hz=1.023e6, code_period_s=0.001, carrier_lock_th=0.85,
    code_length_chips=1023, code_samples_per_chip=1, vector_length=4000, pull_in_time_s=2, veml=0, pll_filter_order=3, dll_filter_order=2,
    enable_fll_pull_in=0, enable_fll_steady_state=0, cn0_samples=20, cn0_min=25, max_lock_fail=50, pll_bw_hz=40.0, dll_bw_hz=2.0, fll_bw_hz=35.0,
    early_late_space_chips=0.5, very_early_late_space_chips=0.0)


def _signal(oracle, prn, fs, n, seed, doppler, delay_samples, cn0=46.0):
    code = oracle.gps_l1_ca_code(prn).astype(np.float32)
    rng = np.random.Generator(np.random.PCG64(seed))
    i = np.arange(n)
    rate = 1.023e6 * (1 + doppler / 1575.42e6) / fs
    tau0 = 1023.0 - delay_samples * 1.023e6 / fs
    chip = np.floor(tau0 + i * rate).astype(np.int64) % 1023
    amp = np.sqrt(10 ** (cn0 / 10) / fs)
    x = (amp * code[chip] * np.exp(1j * (2 * np.pi * doppler * i / fs + 0.4)) + (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(0.5)).astype(np.complex64)
    return code, x


def test_closed_loop_matches_cpu_restatement(gctx, oracle):
    import gnsscorr
    import torch
    from closed_loop_ref import run as ref_run
    fs, n_ep = 4e6, 150
    code, x = _signal(oracle, 5, fs, 4000 * (n_ep + 3), 101, 1680.0, 1234.0)
    conf = dict(GPS, acq_delay_samples=1234.0, acq_doppler_hz=1690.0, acq_samplestamp_samples=0, sample_counter=0)
    ref = ref_run(oracle, x, code, conf, n_ep)
    d = torch.from_numpy(x.view(np.float32)).cuda()
    loop = gnsscorr.TrackingLoop(gctx, 1, 1023)
    loop.set_input_dev(0, d.data_ptr(), x.size)
    loop.start(0, _conf(gnsscorr, **conf), code)
    rec = loop.run(n_ep)[0]
    loop.close()
    assert np.all(rec["valid"] == 1) and np.all(rec["state"] == 2)
    assert len(ref) == n_ep
    worst_d = worst_p = 0.0
    for k in range(n_ep):
        r, g = ref[k], rec[k]
        assert int(g["sample_counter"]) == r["sample_counter"], k      # same block boundaries every epoch
        assert int(g["current_prn_length_samples"]) == r["cur"]
        gp = g["corr"][2] + 1j * g["corr"][3]
        worst_p = max(worst_p, abs(gp - r["corr"][1]) / abs(r["corr"][1]))
        worst_d = max(worst_d, abs(float(g["carrier_doppler_hz"]) - r["doppler"]))
        assert abs(float(g["code_freq_chips"]) - r["code_freq"]) < 0.2   # float32 record of ~1.023e6
    assert worst_p < 2e-3 and worst_d < 0.05, (worst_p, worst_d)
    # the loop locked: Doppler at the truth, C/N0 and lock detector as expected
    assert abs(rec["carrier_doppler_hz"][-30:].mean() - 1680.0) < 3.0
    assert abs(rec["cn0_db_hz"][-1] - ref[-1]["cn0"]) < 0.05 and abs(rec["cn0_db_hz"][-1] - 46.0) < 7.0
    assert abs(rec["carrier_lock_test"][-1] - ref[-1]["lock_test"]) < 1e-3 and rec["carrier_lock_test"][-1] > 0.8


def test_closed_loop_taps_spread_wider_than_the_resident_pad(gctx, oracle):
    """Five taps with the very-early / very-late pair 40 chips out: the chips one period touches (L + 80) exceed L + 64, so the
    kernel cannot address a window inside its resident doubled image and takes the `index mod L` form on the image's first L
    entries (trk_device.hpp, `resident && !windowed`) -- a branch no other closed-loop case reaches.  Same checks as the plain case,
    against the CPU restatement; the outer taps sit far off the correlation peak and see noise plus the code's sidelobes."""
    import gnsscorr
    import torch
    from closed_loop_ref import run as ref_run
    fs, n_ep = 4e6, 60
    code, x = _signal(oracle, 9, fs, 4000 * (n_ep + 3), 414, -2210.0, 777.0)
    conf = dict(GPS, veml=1, very_early_late_space_chips=40.0, acq_delay_samples=777.0, acq_doppler_hz=-2200.0, acq_samplestamp_samples=0, sample_counter=0)
    ref = ref_run(oracle, x, code, conf, n_ep)
    d = torch.from_numpy(x.view(np.float32)).cuda()
    loop = gnsscorr.TrackingLoop(gctx, 1, 1023)
    loop.set_input_dev(0, d.data_ptr(), x.size)
    loop.start(0, _conf(gnsscorr, **conf), code)
    rec = loop.run(n_ep)[0]
    loop.close()
    assert np.all(rec["valid"] == 1) and len(ref) == n_ep
    for k in range(n_ep):
        r, g = ref[k], rec[k]
        assert int(g["sample_counter"]) == r["sample_counter"], k
        assert int(g["current_prn_length_samples"]) == r["cur"]
        got = g["corr"][0:10:2] + 1j * g["corr"][1:10:2]
        assert np.max(np.abs(got - np.asarray(r["corr"]))) <= 2e-3 * abs(r["corr"][2]), k   # all five taps, VE / VL included
        assert abs(float(g["carrier_doppler_hz"]) - r["doppler"]) < 0.05
    assert abs(rec["carrier_doppler_hz"][-20:].mean() + 2210.0) < 3.0
    m = np.abs(rec["corr"][-20:, 0:10:2] + 1j * rec["corr"][-20:, 1:10:2]).mean(axis=0)
    assert m[2] > m[1] > 2 * m[0] and m[2] > m[3] > 2 * m[4]  # prompt on the peak, E / L half a chip off, VE / VL off the peak altogether (noise + sidelobes)


def test_closed_loop_many_channels_and_restart(gctx, oracle):
    """32 channels in one launch on a shared stream; the state persists across launches (2 x 40 epochs ==
    1 x 80 epochs); an exhausted input yields invalid records."""
    import gnsscorr
    import torch
    fs, n_ch = 4e6, 32
    codes = [oracle.gps_l1_ca_code(p).astype(np.float32) for p in range(1, n_ch + 1)]
    x, truth = synth_stream(codes, int(fs), 4000 * 90, seed=77, cn0_db_hz=(44.0, 48.0))
    d = torch.from_numpy(x.view(np.float32)).cuda()

    def make():
        loop = gnsscorr.TrackingLoop(gctx, n_ch, 1023)
        for ch in range(n_ch):
            t = truth[ch]
            delay = ((1023.0 - t["tau0"]) % 1023.0) * fs / 1.023e6
            loop.set_input_dev(ch, d.data_ptr(), x.size)
            loop.start(ch, _conf(gnsscorr, **dict(GPS, acq_delay_samples=float(np.round(delay)), acq_doppler_hz=float(np.round(t["doppler"] / 10) * 10),
                acq_samplestamp_samples=0, sample_counter=0)), codes[ch])
        return loop
    a = make()
    one = a.run(80)
    a.close()
    b = make()
    two = np.concatenate([b.run(40), b.run(40)], axis=1)
    tail = b.run(20)
    b.close()
    assert np.array_equal(one["corr"], two["corr"]) and np.array_equal(one["sample_counter"], two["sample_counter"])
    assert np.all(one["valid"] == 1)
    assert np.all(tail["valid"][:, -5:] == 0)  # 90 ms of input: the last epochs find no samples
    for ch in range(n_ch):
        assert abs(one["carrier_doppler_hz"][ch, -20:].mean() - truth[ch]["doppler"]) < 6.0
        p = one["corr"][ch, -20:, 2] + 1j * one["corr"][ch, -20:, 3]
        assert np.mean(np.abs(p)) > 0.6 * truth[ch]["amp"] * 4000


def test_closed_loop_galileo_e1_veml(gctx, oracle):
    """Five-tap VE/E/P/L/VL loop of Galileo E1 (sinBOC(1,1) replica at 2 samples per chip, 4 ms code period, the VEML
    discriminator of dll_veml..., tracking_discriminators.cc:113-128) against the CPU restatement."""
    import os
    import gnsscorr
    import torch
    from closed_loop_ref import run as ref_run
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    e1b = np.load(os.path.join(G, "galileo_e1_codes.npz"))["e1b"]
    code = oracle.galileo_e1_sinboc11(e1b[4])  # 8184 half-chips
    fs, n, n_ep = 4e6, 16000, 40
    doppler, delay = -1234.0, 5000.0
    rng = np.random.Generator(np.random.PCG64(88))
    i = np.arange(n * (n_ep + 3))
    rate = 2.046e6 * (1 + doppler / 1575.42e6) / fs  # half-chips per sample
    tau0 = 8184.0 - delay * 2.046e6 / fs
    chip = np.floor(tau0 + i * rate).astype(np.int64) % 8184
    amp = np.sqrt(10 ** (45.0 / 10) / fs)
    x = (amp * code[chip] * np.exp(1j * (2 * np.pi * doppler * i / fs + 1.1)) + (rng.standard_normal(i.size) + 1j * rng.standard_normal(i.size)) * np.sqrt(0.5)).astype(np.complex64)
    conf = dict(fs_in=fs, signal_carrier_freq_hz=1575.42e6, code_chip_rate_hz=1.023e6, code_period_s=0.004, carrier_lock_th=0.85,
        code_length_chips=4092, code_samples_per_chip=2, vector_length=n, pull_in_time_s=2, veml=1, pll_filter_order=3, dll_filter_order=2,
        enable_fll_pull_in=0, enable_fll_steady_state=0, cn0_samples=10, cn0_min=25, max_lock_fail=50, pll_bw_hz=15.0, dll_bw_hz=0.75, fll_bw_hz=10.0,
        early_late_space_chips=0.15, very_early_late_space_chips=0.6, acq_delay_samples=delay, acq_doppler_hz=doppler + 4.0,
        acq_samplestamp_samples=0, sample_counter=0)
    ref = ref_run(oracle, x, code, conf, n_ep)
    d = torch.from_numpy(x.view(np.float32)).cuda()
    loop = gnsscorr.TrackingLoop(gctx, 1, 8184)
    loop.set_input_dev(0, d.data_ptr(), x.size)
    loop.start(0, _conf(gnsscorr, **conf), code)
    rec = loop.run(n_ep)[0]
    loop.close()
    assert len(ref) == n_ep and np.all(rec["valid"] == 1)
    for k in range(n_ep):
        r, g = ref[k], rec[k]
        assert int(g["sample_counter"]) == r["sample_counter"], k
        gp = g["corr"][4] + 1j * g["corr"][5]
        assert abs(gp - r["corr"][2]) <= 2e-3 * abs(r["corr"][2])
        assert abs(float(g["carrier_doppler_hz"]) - r["doppler"]) < 0.05
        assert abs(float(g["code_error_chips"]) - r["cerr"]) < 2e-3
    # prompt on the BOC main peak, E/L 0.15 chip off (~70 % of it), VE/VL 0.6 chip off (near the BOC side lobes' zero)
    m = np.abs(rec["corr"][-10:, 0::2] + 1j * rec["corr"][-10:, 1::2]).mean(axis=0)
    assert m[2] > m[1] > m[0] and m[2] > m[3] > m[4]
    assert abs(rec["carrier_doppler_hz"][-10:].mean() - doppler) < 3.0


def test_closed_loop_declares_loss_of_lock(gctx, oracle):
    """The satellite disappears mid-stream: cn0_and_tracking_lock_status (dll_pll_veml_tracking.cc:839-878) counts
    failed lock tests and, past max_lock_fail, the channel goes to standby (state 0) and stops producing valid
    records -- the device loop's image of the block's "loss of lock" message."""
    import gnsscorr
    import torch
    fs, n_ep = 4e6, 260
    code, x = _signal(oracle, 21, fs, 4000 * (n_ep + 3), 31, 440.0, 1000.0)
    rng = np.random.Generator(np.random.PCG64(32))
    cut = 4000 * 80
    x[cut:] = ((rng.standard_normal(x.size - cut) + 1j * rng.standard_normal(x.size - cut)) * np.sqrt(0.5)).astype(np.complex64)
    conf = dict(GPS, acq_delay_samples=1000.0, acq_doppler_hz=445.0, acq_samplestamp_samples=0, sample_counter=0)
    conf.update(cn0_samples=10, max_lock_fail=5, pull_in_time_s=0)  # the lock counter only runs after the pull-in transitory
    conf["acq_samplestamp_samples"] = 0
    d = torch.from_numpy(x.view(np.float32)).cuda()
    loop = gnsscorr.TrackingLoop(gctx, 1, 1023)
    loop.set_input_dev(0, d.data_ptr(), x.size)
    # pull_in_time_s = 0 ends the transitory after one second of stream: start the channel's counter one second in
    c = _conf(gnsscorr, **dict(conf, sample_counter=int(1.1 * fs), acq_samplestamp_samples=int(1.1 * fs) - int(1.05 * fs)))
    loop.start(0, c, code)
    rec = loop.run(n_ep)[0]
    loop.close()
    states = rec["state"]
    assert np.all(states[:75] == 2) and np.all(rec["valid"][:75] == 1)      # tracking while the signal is there
    lost = int(np.argmax(states == 0))
    assert states[lost] == 0 and 80 < lost <= 80 + 11 * (5 + 2) + 11        # ~max_lock_fail failed tests of 11 epochs each
    assert np.all(states[lost:] == 0) and np.all(rec["valid"][lost:] == 0)   # standby from then on


def test_closed_loop_on_the_real_glonass_capture(gctx, oracle):
    """GlonassL1CaDllPllTrackingTest.ValidationOfResults (glonass_l1_ca_dll_pll_tracking_test.cc:151-214) runs the tracking
    block over the 4 ms NT1065 capture from the hand-over (1343 samples, -2750 Hz, PRN 11) without asserting anything; here
    the device loop runs the same three code periods and the prompt must sit on the correlation peak in each of them."""
    import json
    import os
    import gnsscorr
    import torch
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    k = json.load(open(os.path.join(G, "kat_expected.json")))["glonass_l1_ca"]
    x = np.fromfile(os.path.join(G, k["file"]), np.complex64)
    g = k["reference_test"]
    conf = dict(fs_in=float(k["fs"]), signal_carrier_freq_hz=1602.0e6, code_chip_rate_hz=0.511e6, code_period_s=0.001, carrier_lock_th=0.85,
        code_length_chips=511, code_samples_per_chip=1, vector_length=6625, pull_in_time_s=2, veml=0, pll_filter_order=2, dll_filter_order=2,
        enable_fll_pull_in=0, enable_fll_steady_state=0, cn0_samples=20, cn0_min=25, max_lock_fail=50, pll_bw_hz=2.0, dll_bw_hz=0.5, fll_bw_hz=10.0,
        early_late_space_chips=0.5, very_early_late_space_chips=0.0, acq_delay_samples=float(g["expected_delay_samples"]),
        acq_doppler_hz=float(g["expected_doppler_hz"]), acq_samplestamp_samples=0, sample_counter=0)
    d = torch.from_numpy(x.view(np.float32).copy()).cuda()
    loop = gnsscorr.TrackingLoop(gctx, 1, 511)
    loop.set_input_dev(0, d.data_ptr(), x.size)
    loop.start(0, _conf(gnsscorr, **conf), gnsscorr.glonass_l1_ca_code_gen_float())
    rec = loop.run(4)[0]
    loop.close()
    # the pull-in of the block skips T_prn - fmod(-1343, T_prn) = 6625 + 1343 samples (dll_pll_veml_tracking.cc:1568-1600),
    # so two whole code periods of the 26499-sample capture are left
    assert list(rec["valid"]) == [1, 1, 0, 0]
    assert int(rec["sample_counter"][0]) == 6625 + 1343 + 6625 and abs(int(rec["sample_counter"][1]) - int(rec["sample_counter"][0]) - 6625) <= 1
    mag = np.abs(rec["corr"][:2, 0::2] + 1j * rec["corr"][:2, 1::2])[:, :3]
    assert np.all(mag[:, 1] > 1.25 * mag[:, 0]) and np.all(mag[:, 1] > 1.25 * mag[:, 2]) and np.all(mag[:, 1] > 700.0)
    assert np.all(np.abs(rec["carrier_doppler_hz"][:2] - g["expected_doppler_hz"]) < 50.0)


def test_closed_loop_workgroup_sizes_agree(gctx, oracle):
    """The device loop picks 1024 / 512 / 256 threads per channel from the channel count; the three variants track the
    same signal to the same loop state (block boundaries identical, correlator sums equal to float rounding)."""
    import gnsscorr
    import torch
    fs, n_ep = 4e6, 80
    code, x = _signal(oracle, 17, fs, 4000 * (n_ep + 3), 123, -777.0, 3100.0)
    conf = dict(GPS, acq_delay_samples=3100.0, acq_doppler_hz=-770.0, acq_samplestamp_samples=0, sample_counter=0)
    d = torch.from_numpy(x.view(np.float32)).cuda()
    recs = {}
    for threads in (1024, 512, 256):
        loop = gnsscorr.TrackingLoop(gctx, 2, 1023)
        loop.set_geometry(threads_per_workgroup=threads, slices_per_channel=1)
        for ch in range(2):
            loop.set_input_dev(ch, d.data_ptr(), x.size)
            loop.start(ch, _conf(gnsscorr, **conf), code)
        recs[threads] = loop.run(n_ep)
        loop.close()
    base = recs[1024]
    assert np.all(base["valid"] == 1) and np.array_equal(base[0], base[1])  # two channels on the same signal: identical
    for threads in (512, 256):
        r = recs[threads]
        assert np.array_equal(r["sample_counter"], base["sample_counter"])
        assert np.array_equal(r["current_prn_length_samples"], base["current_prn_length_samples"])
        assert np.max(np.abs(r["corr"] - base["corr"])) <= 2e-5 * np.max(np.abs(base["corr"]))
        assert np.max(np.abs(r["carrier_doppler_hz"] - base["carrier_doppler_hz"])) < 0.02


def test_closed_loop_high_dynamics(gctx, oracle):
    """Dll_Pll_Conf::high_dyn on the device loop: the high-dynamics resampler / rotator kernels plus the carrier and code rate
    smoothers of update_tracking_vars (dll_pll_veml_tracking.cc:1016-1033, :1047-1064), on a signal whose Doppler ramps at
    400 Hz/s, against the CPU restatement; the smoothed carrier rate converges to the ramp."""
    import gnsscorr
    import torch
    from closed_loop_ref import run as ref_run
    fs, n_ep, ramp = 4e6, 220, 400.0
    code = oracle.gps_l1_ca_code(11).astype(np.float32)
    rng = np.random.Generator(np.random.PCG64(55))
    n = 4000 * (n_ep + 3)
    t = np.arange(n) / fs
    f0 = 900.0
    phase = 2 * np.pi * (f0 * t + 0.5 * ramp * t * t)
    tau = (1023.0 - 321.0 * 1.023e6 / fs) + 1.023e6 * (t + (f0 * t + 0.5 * ramp * t * t) / 1575.42e6)
    chip = np.floor(tau).astype(np.int64) % 1023
    amp = np.sqrt(10 ** (50.0 / 10) / fs)
    x = (amp * code[chip] * np.exp(1j * (phase + 0.3)) + (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(0.5)).astype(np.complex64)
    conf = dict(GPS, acq_delay_samples=321.0, acq_doppler_hz=f0 + 3.0, acq_samplestamp_samples=0, sample_counter=0, high_dyn_smoother_length=8,
        pull_in_time_s=0)
    ref = ref_run(oracle, x, code, conf, n_ep)
    d = torch.from_numpy(x.view(np.float32)).cuda()
    loop = gnsscorr.TrackingLoop(gctx, 1, 1023)
    loop.set_input_dev(0, d.data_ptr(), x.size)
    loop.start(0, _conf(gnsscorr, **conf), code)
    rec = loop.run(n_ep)[0]
    loop.close()
    assert len(ref) == n_ep and np.all(rec["valid"] == 1)
    one = 2.2 * float(np.abs(x).max())
    for k in range(n_ep):
        r, g = ref[k], rec[k]
        assert int(g["sample_counter"]) == r["sample_counter"], k
        gp = g["corr"][2] + 1j * g["corr"][3]
        assert abs(gp - r["corr"][1]) <= max(2e-3 * abs(r["corr"][1]), one), k
        assert abs(float(g["carrier_doppler_hz"]) - r["doppler"]) < 0.3, k
    # the loop follows the ramp: Doppler at the end is f0 + ramp * t
    t_end = float(rec["sample_counter"][-1]) / fs
    assert abs(float(rec["carrier_doppler_hz"][-1]) - (f0 + ramp * t_end)) < 8.0
    # one engine, one mode
    loop = gnsscorr.TrackingLoop(gctx, 2, 1023)
    for ch in range(2):
        loop.set_input_dev(ch, d.data_ptr(), x.size)
    loop.start(0, _conf(gnsscorr, **conf), code)
    with pytest.raises(gnsscorr.GnsscorrError, match="high_dyn mode"):
        loop.start(1, _conf(gnsscorr, **dict(conf, high_dyn_smoother_length=0)), code)
    loop.close()


def test_closed_loop_standby_slots_and_stop(gctx, oracle):
    """An engine sized for the receiver's channel count: slots that were never started, were stopped, or lost lock produce all-zero
    standby records and do not disturb the running ones (gc_trk_loop_stop)."""
    import gnsscorr
    import torch
    fs, n_ep = 4e6, 40
    code, x = _signal(oracle, 5, fs, 4000 * (n_ep + 3), 101, 1680.0, 1234.0)
    conf = _conf(gnsscorr, **dict(GPS, acq_delay_samples=1234.0, acq_doppler_hz=1690.0, acq_samplestamp_samples=0, sample_counter=0))
    d = torch.from_numpy(x.view(np.float32)).cuda()
    solo = gnsscorr.TrackingLoop(gctx, 1, 1023)
    solo.set_input_dev(0, d.data_ptr(), x.size)
    solo.start(0, conf, code)
    want = solo.run(n_ep)[0]
    solo.close()
    loop = gnsscorr.TrackingLoop(gctx, 3, 1023)
    for ch in range(3):
        loop.set_input_dev(ch, d.data_ptr(), x.size)
    with pytest.raises(gnsscorr.GnsscorrError, match="no channel has been started"):
        loop.run(1)
    loop.start(1, conf, code)
    rec = loop.run(n_ep // 2)
    assert np.all(rec[0].view(np.uint8) == 0) and np.all(rec[2].view(np.uint8) == 0)
    loop.start(2, conf, code)  # a later hand-over into a free slot
    rec2 = loop.run(n_ep - n_ep // 2)
    got = np.concatenate([rec[1], rec2[1]])
    for f in ("corr", "sample_counter", "carrier_doppler_hz", "state", "valid"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(rec2[2]["corr"], want["corr"][:n_ep - n_ep // 2])  # slot 2 started from the beginning of its block
    loop.stop(1)
    rec3 = loop.run(2)
    assert np.all(rec3[1].view(np.uint8) == 0) and np.all(rec3[0].view(np.uint8) == 0) and np.all(rec3[2]["valid"] == 1)
    loop.stop(2)
    with pytest.raises(gnsscorr.GnsscorrError, match="no channel has been started"):
        loop.run(1)
    loop.close()


@pytest.mark.parametrize("order,fll_pull_in,fll_steady", [(3, 0, 0), (2, 0, 0), (3, 0, 1), (2, 1, 0), (3, 1, 0)])
def test_device_carrier_filter_against_the_pinned_filter(gctx, oracle, order, fll_pull_in, fll_steady):
    """The device loop's copy of Tracking_FLL_PLL_filter (trk_closed_loop.hip::pll_get_carrier_error) against
    tests/closed_loop_ref.py::Pll, which tests/test_loop_filter_pin.py holds BIT FOR BIT to the reference's filter compiled in
    oracle/_ref: the pinned filter is fed with the discriminator outputs the DEVICE recorded (carr_phase_error_hz is the
    filter's own float input; the FLL discriminator is re-derived from the recorded prompt accumulators with the reference's
    formula, tracking_discriminators.cc:41-56) and must reproduce the recorded carrier_doppler_hz -- exactly where only the PLL
    input is used, within 2 float32 ulp where the re-derived FLL input (a double atan2 on another libm) enters."""
    import gnsscorr
    import torch
    from closed_loop_ref import Pll
    fs, n_ep = 4e6, 120
    code, x = _signal(oracle, 9, fs, 4000 * (n_ep + 3), 303, -2210.0, 2345.0)
    conf = dict(GPS, acq_delay_samples=2345.0, acq_doppler_hz=-2200.0, acq_samplestamp_samples=0, sample_counter=0,
        pll_filter_order=order, enable_fll_pull_in=fll_pull_in, enable_fll_steady_state=fll_steady)
    d = torch.from_numpy(x.view(np.float32)).cuda()
    loop = gnsscorr.TrackingLoop(gctx, 1, 1023)
    loop.set_input_dev(0, d.data_ptr(), x.size)
    loop.start(0, _conf(gnsscorr, **conf), code)
    rec = loop.run(n_ep)[0]
    loop.close()
    assert np.all(rec["valid"] == 1) and np.all(rec["state"] == 2)
    f = Pll(conf["fll_bw_hz"], conf["pll_bw_hz"], order)
    f.initialize(np.float32(conf["acq_doppler_hz"]))
    T = np.float32(conf["code_period_s"])
    p_old = np.zeros(2, np.float32)
    worst = 0
    for k in range(n_ep):
        g = rec[k]
        P = np.array([g["accu"][4], g["accu"][5]], np.float32)  # the prompt accumulator the discriminators saw
        pll_in = np.float32(g["carr_phase_error_hz"])
        if fll_pull_in or fll_steady:
            dot = np.float32(np.float32(p_old[0] * P[0]) + np.float32(p_old[1] * P[1]))
            cross = np.float32(np.float32(p_old[0] * P[1]) - np.float32(P[0] * p_old[1]))
            fll_in = np.float32(np.arctan2(float(cross), float(dot)) / conf["code_period_s"] / (2.0 * np.pi))  # double, like the device
            p_old = P
            want = f.get_carrier_error(fll_in, np.float32(0.0) if fll_pull_in else pll_in, T)
        else:
            want = f.get_carrier_error(np.float32(0.0), pll_in, T)
        got = np.float32(g["carrier_doppler_hz"])
        ulp = abs(int(np.float32(want).view(np.int32)) - int(got.view(np.int32)))
        worst = max(worst, ulp)
        if not (fll_pull_in or fll_steady):
            assert ulp == 0, (k, float(want), float(got))
        # keep the pinned filter on the device's trajectory: its state is a function of its inputs only, so nothing to re-seed
    assert worst <= (2 if (fll_pull_in or fll_steady) else 0), worst


def test_product_library_refuses_sliced_periods(gctx):
    """gc_trk_loop_set_geometry: thread counts are a product setting; slices > 1 (one launch per code period) measured slower and
    exist in the experiments build only -- the product library says so instead of silently ignoring the request."""
    import gnsscorr
    if gnsscorr.load_library().gc_build_has_experiments():
        pytest.skip("experiments build")
    loop = gnsscorr.TrackingLoop(gctx, 2, 1023)
    loop.set_geometry(threads_per_workgroup=512, slices_per_channel=1)
    loop.set_geometry(0, 0)
    with pytest.raises(Exception) as e:
        loop.set_geometry(slices_per_channel=4)
    assert "experiments" in str(e.value)
    with pytest.raises(Exception):
        loop.set_geometry(threads_per_workgroup=300)
    loop.close()
