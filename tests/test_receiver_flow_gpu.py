"""End-to-end flow on the GPU through the C ABI, the way a receiver would drive it: one RF stream pushed
block by block into the HBM ring; PCPS acquisition of a PRN list on the first block (present and absent
satellites); acquisition -> tracking hand-over (delay, Doppler, sample stamp, as ChannelFsm / start_tracking do,
channel_fsm.cc:192-219, dll_pll_veml_tracking.cc:549-557); closed-loop DLL/PLL tracking of every detected
satellite on the same ring, a few code periods per launch as the blocks arrive."""
import numpy as np
import pytest

from helpers import synth_stream

pytestmark = pytest.mark.gpu


def test_acquire_then_track_on_one_ring(gctx, oracle):
    import gnsscorr
    from test_closed_loop_gpu import GPS, _conf
    fs, n = 4_000_000, 4000
    present, absent = [3, 8, 14, 22], [5, 11, 19, 30]
    codes = {p: oracle.gps_l1_ca_code(p).astype(np.float32) for p in present + absent}
    n_ms = 260
    x, truth = synth_stream([codes[p] for p in present], fs, n_ms * n, seed=404, cn0_db_hz=(46.0, 50.0), doppler_max=4000.0)
    ring = gnsscorr.IqStream(gctx, capacity_samples=40 * n, max_window_samples=4 * n)  # the 4 ms acquisition block is the longest window
    ring.push(x[:4 * n])

    # ---- acquisition: 8 PRNs searched at once on the first 4 ms (coherent), straight from the ring; 4 ms and
    # 50 Hz bins put the Doppler estimate inside the pull-in range of the 40 Hz PLL ----
    prns = present + absent
    acq = gnsscorr.PcpsAcquisition(gctx, len(prns), fs, 4, 1, np.float32(fs) * np.float32(0.001), 4000.0, 4, 5000, 50)
    assert (acq.consumed_samples, acq.fft_size) == (4 * n, 8 * n)
    for s, p in enumerate(prns):
        acq.set_local_code(s, np.tile(oracle.gps_l1_ca_code_sampled(p, fs), 4))  # gps_l1_ca_pcps_acquisition.cc:239-243
    res = acq.dwell_stream(ring, 0)
    acq.close()
    stats = np.array([r.test_statistics for r in res])
    detected = [s for s in range(len(prns)) if stats[s] > 2.0 * np.median(stats[len(present):])]
    assert [prns[s] for s in detected] == present, stats

    # ---- hand-over + closed-loop tracking on the same ring ----
    loop = gnsscorr.TrackingLoop(gctx, len(detected), 1023)
    for ch, s in enumerate(detected):
        r = res[s]
        t = truth[ch]
        assert abs(r.acq_doppler_hz - t["doppler"]) <= 50.0
        conf = dict(GPS, acq_delay_samples=float(r.acq_delay_samples), acq_doppler_hz=float(r.acq_doppler_hz),
            acq_samplestamp_samples=0, sample_counter=0)
        loop.set_input_stream(ch, ring)
        loop.start(ch, _conf(gnsscorr, **conf), codes[prns[s]])
    recs = [[] for _ in detected]
    for ms in range(4, n_ms, 5):
        ring.push(x[ms * n:(ms + 5) * n])  # 5 ms blocks
        out = loop.run(6)
        for ch in range(len(detected)):
            recs[ch].extend(r.copy() for r in out[ch] if r["valid"])
    loop.close()
    ring.close()
    for ch in range(len(detected)):
        rr = np.array(recs[ch])
        t = truth[ch]
        assert len(rr) >= n_ms - 3                       # every complete code period was tracked
        assert np.all(np.diff(rr["sample_counter"].astype(np.int64)) >= n - 1)
        assert abs(rr["carrier_doppler_hz"][-50:].mean() - t["doppler"]) < 3.0   # PLL locked on the true Doppler
        # the SNV estimate over 20 periods reads low at 46-50 dB-Hz (loop phase jitter counts as noise), as it does in the reference
        assert rr["carrier_lock_test"][-1] > 0.8 and 36.0 < rr["cn0_db_hz"][-1] < t["cn0"] + 3.0
        p = rr["corr"][-50:, 2] + 1j * rr["corr"][-50:, 3]
        e = rr["corr"][-50:, 0] + 1j * rr["corr"][-50:, 1]
        assert np.abs(p).mean() > 1.5 * np.abs(e).mean()  # prompt on the correlation peak, early half a chip off
