"""GPU tests of the RF stream ring (gc_stream_*, SURVEY.md section 8b/8e "shared IQ ring per RF stream"):
blocks pushed from the host in ragged sizes, windows addressed by absolute sample number across the ring's wrap,
results identical to the same kernels reading one linear buffer."""
import numpy as np
import pytest

from helpers import open_loop_params, synth_stream

pytestmark = pytest.mark.gpu


def test_tracking_from_ring_equals_linear_buffer(gctx, oracle):
    import gnsscorr
    import torch
    fs, n, n_epochs = 4_000_000, 4000, 12
    codes = [oracle.gps_l1_ca_code(p).astype(np.float32) for p in (2, 5)]
    sig, truth = synth_stream(codes, fs, n_epochs * n, seed=31, cn0_db_hz=(45.0, 48.0))
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    recs = []
    for ch in range(2):
        recs.append([gnsscorr.epoch_params(p["sample_offset"], float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n)
            for p in open_loop_params(truth[ch], fs, 1023, n, n_epochs)])
    # linear reference run
    d_sig = torch.from_numpy(sig.view(np.float32).copy()).cuda()
    lin = gnsscorr.TrackingBatch(gctx, 2, 3, 1023)
    lin.set_slices(2)
    for ch in range(2):
        lin.set_code(ch, codes[ch], shifts)
        lin.set_input_dev(ch, d_sig.data_ptr(), sig.size)
    want = lin.run(n_epochs, gnsscorr.epoch_params_array(recs))
    lin.close()
    # ring of 2.5 epochs' capacity: the stream wraps several times; blocks of ragged sizes
    ring = gnsscorr.IqStream(gctx, capacity_samples=10000, max_window_samples=n)
    b = gnsscorr.TrackingBatch(gctx, 2, 3, 1023)
    b.set_slices(2)
    for ch in range(2):
        b.set_code(ch, codes[ch], shifts)
        b.set_input_stream(ch, ring)
    got = np.zeros_like(want)
    pushed, done = 0, 0
    sizes = [1777, 2500, 3999, 811, 6000, 4000, 123]
    k = 0
    while done < n_epochs:
        if pushed < sig.size:
            m = min(sizes[k % len(sizes)], sig.size - pushed)
            k += 1
            assert ring.push(sig[pushed:pushed + m]) == pushed
            pushed += m
        while done < n_epochs and (done + 1) * n <= pushed:
            # one epoch of both channels as soon as its samples are resident
            out = b.run(1, gnsscorr.epoch_params_array([[recs[0][done]], [recs[1][done]]]))
            got[:, done] = out[:, 0]
            done += 1
    oldest, head, cap = ring.info()
    assert (head, cap) == (sig.size, 10000) and oldest == sig.size - 10000
    assert np.array_equal(got, want)  # same kernel, same samples: bit-identical
    # a window that has been evicted, one beyond the head, and one longer than max_window are refused
    for bad in (gnsscorr.epoch_params(0, 0.0, 0.0, 0.0, 0.25, n), gnsscorr.epoch_params(sig.size - 10, 0.0, 0.0, 0.0, 0.25, n),
            gnsscorr.epoch_params(sig.size - 9000, 0.0, 0.0, 0.0, 0.25, 2 * n)):
        with pytest.raises(gnsscorr.GnsscorrError):
            b.run(1, gnsscorr.epoch_params_array([[bad], [bad]]))
    b.close()
    ring.close()


def test_int16_ring_and_run_dev_with_read_floor(gctx, oracle):
    """cshort ring, device-resident parameters, several pushes in flight behind the compute launches."""
    import gnsscorr
    import torch
    fs, n, n_epochs = 25_000_000, 25000, 8
    code = oracle.gps_l1_ca_code(7).astype(np.float32)
    sig, truth = synth_stream([code], fs, n_epochs * n, seed=32, cn0_db_hz=(48.0, 48.0))
    q = np.round(sig.view(np.float32).reshape(-1, 2) * 32.0).astype(np.int16)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    ps = open_loop_params(truth[0], fs, 1023, n, n_epochs)
    recs = [gnsscorr.epoch_params(p["sample_offset"], float(p["rem_carr"]), float(p["phase_step"]), float(p["rem_code"]), float(p["code_step"]), n) for p in ps]
    d_q = torch.from_numpy(q).cuda()
    lin = gnsscorr.TrackingBatch(gctx, 1, 3, 1023)
    lin.set_input_format(gnsscorr.GC_IQ_I16)
    lin.set_code(0, code, shifts)
    lin.set_input_dev(0, d_q.data_ptr(), q.shape[0])
    lin.set_slices(4)  # same cut of every epoch in both runs: the partial sums are added in the same order
    want = lin.run(n_epochs, gnsscorr.epoch_params_array(recs))[0]
    lin.close()
    ring = gnsscorr.IqStream(gctx, capacity_samples=4 * n, max_window_samples=n, iq_format=gnsscorr.GC_IQ_I16)
    b = gnsscorr.TrackingBatch(gctx, 1, 3, 1023)
    b.set_input_format(gnsscorr.GC_IQ_I16)
    b.set_code(0, code, shifts)
    b.set_input_stream(0, ring)
    b.set_slices(4)
    d_params = torch.from_numpy(gnsscorr.epoch_params_array(recs).view(np.uint8)).cuda()
    d_out = torch.zeros(n_epochs * 3, 2, device="cuda", dtype=torch.float32)
    stream = torch.cuda.Stream()
    for k in range(n_epochs):
        ring.push(q[k * n:(k + 1) * n])
        b.set_read_floor(k * n)  # this launch reads epoch k only: later pushes may evict everything older
        b.run_dev(1, d_params.data_ptr() + 48 * k, d_out.data_ptr() + 24 * k, stream.cuda_stream)
    stream.synchronize()
    got = d_out.cpu().numpy().view(np.complex64).reshape(n_epochs, 3)
    assert np.array_equal(got, want)
    b.close()
    ring.close()


def test_acquisition_from_ring(gctx, oracle):
    import json
    import os
    import gnsscorr
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    k = json.load(open(os.path.join(G, "kat_expected.json")))["gps_l1_ca"]
    x = np.fromfile(os.path.join(G, k["file"]), np.complex64)
    fs = k["fs"]
    acq = gnsscorr.PcpsAcquisition(gctx, 1, fs, 1, 1, np.float32(fs) * np.float32(0.001), 4000.0, 4, k["doppler_max"], k["doppler_step"])
    acq.set_local_code(0, oracle.gps_l1_ca_code_sampled(k["prn"], fs))
    want = acq.dwell(x[3000:7000])[0]
    acq.reset()
    ring = gnsscorr.IqStream(gctx, capacity_samples=9000, max_window_samples=4000)
    ring.push(np.zeros(7500, np.complex64))  # so that the block of interest straddles the wrap: it starts at 8000 % 9000... below
    ring.push(x[:1500])
    ring.push(x[1500:8000])
    got = acq.dwell_stream(ring, 7500 + 3000)[0]  # samples x[3000:7000] live at ring positions 1500 .. 5500 after wrapping
    assert (got.indext, got.doppler_hz, got.mag, got.test_statistics) == (want.indext, want.doppler_hz, want.mag, want.test_statistics)
    with pytest.raises(gnsscorr.GnsscorrError):
        acq.dwell_stream(ring, 100)  # evicted
    acq.close()
    ring.close()


def test_handles_outlive_their_context_and_stream():
    """Destroy order must not matter: the context (and a ring) stay alive while handles created on them exist."""
    import gnsscorr
    ctx = gnsscorr.Context(0)
    ring = gnsscorr.IqStream(ctx, 8000, 4000)
    b = gnsscorr.TrackingBatch(ctx, 1, 3, 1023)
    b.set_code(0, np.ones(1023, np.float32), np.zeros(3, np.float32))
    b.set_input_stream(0, ring)
    ring.push(np.ones(4000, np.complex64))
    corr = gnsscorr.HipMulticorrelatorRealCodes(ctx)
    corr.init(100, 3)
    ctx.close()   # creator's reference only
    ring.close()  # the batch still reads it
    out = b.run(1, gnsscorr.epoch_params_array([[gnsscorr.epoch_params(0, 0.0, 0.0, 0.0, 0.25575, 4000)]]))
    assert out[0, 0, 1] == np.complex64(4000)
    b.close()
    corr.close()


def test_closed_loop_on_the_ring_equals_block_input(gctx, oracle):
    """Level 3 fed by the ring: push a block, run, repeat -- the per-epoch records equal those of the same
    loop run over the whole capture held as one block (same windows, same loop maths, same order)."""
    import gnsscorr
    import torch
    from test_closed_loop_gpu import GPS, _conf, _signal
    fs, n_ep = 4e6, 60
    code, x = _signal(oracle, 9, fs, 4000 * (n_ep + 3), 55, -2210.0, 777.0)
    conf = dict(GPS, acq_delay_samples=777.0, acq_doppler_hz=-2200.0, acq_samplestamp_samples=0, sample_counter=0)
    d = torch.from_numpy(x.view(np.float32)).cuda()
    lin = gnsscorr.TrackingLoop(gctx, 1, 1023)
    lin.set_input_dev(0, d.data_ptr(), x.size)
    lin.start(0, _conf(gnsscorr, **conf), code)
    want = lin.run(n_ep)[0]
    lin.close()
    assert np.all(want["valid"] == 1)

    ring = gnsscorr.IqStream(gctx, capacity_samples=24000, max_window_samples=4000)
    loop = gnsscorr.TrackingLoop(gctx, 1, 1023)
    loop.set_input_stream(0, ring)
    loop.start(0, _conf(gnsscorr, **conf), code)
    got = []
    pushed = 0
    while len(got) < n_ep and pushed < x.size:
        m = min(6500, x.size - pushed)  # 1.6 code periods per push: launches see 1 or 2 complete periods
        ring.push(x[pushed:pushed + m])
        pushed += m
        rec = loop.run(3)[0]
        for r in rec:
            if r["valid"]:
                got.append(r.copy())
    got = np.array(got[:n_ep])
    assert len(got) == n_ep
    for name in want.dtype.names:
        assert np.array_equal(got[name], want[name]), name
    # a channel that is not serviced while the stream moves on falls out of the ring: reported, not silently wrong
    for _ in range(6):
        ring.push(np.zeros(6000, np.complex64))
    with pytest.raises(gnsscorr.GnsscorrError):
        loop.run(1)
    loop.close()
    ring.close()


def test_closed_loop_on_a_cshort_ring(gctx, oracle):
    """The closed-loop engine fed with lv_16sc_t samples (as an SDR front-end delivers them) through a cshort ring:
    same records as the float engine run on the converted samples."""
    import gnsscorr
    import torch
    from test_closed_loop_gpu import GPS, _conf, _signal
    fs, n_ep = 4e6, 40
    code, x = _signal(oracle, 12, fs, 4000 * (n_ep + 3), 66, 905.0, 2100.0)
    q = np.round(x.view(np.float32).reshape(-1, 2) * 64.0).astype(np.int16)
    xf = q.astype(np.float32).reshape(-1).view(np.complex64)  # what the cast on load produces
    conf = dict(GPS, acq_delay_samples=2100.0, acq_doppler_hz=900.0, acq_samplestamp_samples=0, sample_counter=0)
    d = torch.from_numpy(xf.view(np.float32).copy()).cuda()
    lin = gnsscorr.TrackingLoop(gctx, 1, 1023)
    lin.set_input_dev(0, d.data_ptr(), xf.size)
    lin.start(0, _conf(gnsscorr, **conf), code)
    want = lin.run(n_ep)[0]
    lin.close()
    ring = gnsscorr.IqStream(gctx, capacity_samples=20000, max_window_samples=4000, iq_format=gnsscorr.GC_IQ_I16)
    loop = gnsscorr.TrackingLoop(gctx, 1, 1023)
    loop.set_input_format(gnsscorr.GC_IQ_I16)
    loop.set_input_stream(0, ring)
    loop.start(0, _conf(gnsscorr, **conf), code)
    got, pushed = [], 0
    while len(got) < n_ep and pushed < q.shape[0]:
        m = min(8000, q.shape[0] - pushed)
        ring.push(q[pushed:pushed + m])
        pushed += m
        got.extend(r.copy() for r in loop.run(3)[0] if r["valid"])
    got = np.array(got[:n_ep])
    assert len(got) == n_ep and np.all(want["valid"] == 1)
    for name in want.dtype.names:
        assert np.array_equal(got[name], want[name]), name
    with pytest.raises(gnsscorr.GnsscorrError):
        loop.set_input_format(gnsscorr.GC_IQ_F32)  # not while channels are bound
    loop.close()
    ring.close()


def test_broadcast_of_one_pinned_block_into_several_rings(gctx, oracle):
    """gc_stream_broadcast_pinned: the RF stream's copy for every GPU of a node (two rings on this GPU stand in for two GPUs):
    both rings hold the block, channels reading either ring produce identical results."""
    import gnsscorr
    import torch
    N, n_ep = 4000, 6
    rng = np.random.Generator(np.random.PCG64(77))
    x = (rng.standard_normal(N * (n_ep + 1)) + 1j * rng.standard_normal(N * (n_ep + 1))).astype(np.complex64)
    host = torch.from_numpy(x.view(np.float32)).pin_memory()
    rings = [gnsscorr.IqStream(gctx, N * 16, 2 * N) for _ in range(2)]
    gnsscorr.stream_broadcast_pinned(rings, host.data_ptr(), x.size)
    code = oracle.gps_l1_ca_code(7).astype(np.float32)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    outs = []
    for r in rings:
        r.synchronize()
        assert r.info()[1] == x.size
        b = gnsscorr.TrackingBatch(gctx, 1, 3, 1023)
        b.set_code(0, code, shifts)
        b.set_input_stream(0, r)
        recs = [[gnsscorr.epoch_params(e * N, 0.2, 2e-3, 0.1, 0.25575, N) for e in range(n_ep)]]
        outs.append(b.run(n_ep, gnsscorr.epoch_params_array(recs)))
        b.close()
    assert np.array_equal(outs[0], outs[1]) and np.abs(outs[0]).max() > 0
    with pytest.raises(gnsscorr.GnsscorrError, match="another sample format"):
        other = gnsscorr.IqStream(gctx, N * 16, 2 * N, gnsscorr.GC_IQ_I16)
        gnsscorr.stream_broadcast_pinned([rings[0], other], host.data_ptr(), 16)
    for r in rings:
        r.close()


def test_closed_loop_push_run_dev_back_to_back_without_sync(gctx, oracle):
    """push -> run_dev -> push -> run_dev ... with NO host synchronisation in between, the launches queued behind a slow kernel on
    the caller's stream so that several are pending at once: every launch must see the ring head of ITS OWN push (the limits
    travel through a ring of slots, one per pending launch), not the head of a later one whose DMA may not have landed.  The valid
    records, in order, equal those of the same capture tracked as one block with run()."""
    import gnsscorr
    import torch
    from test_closed_loop_gpu import GPS, _conf, _signal
    fs, n_ep = 4e6, 48
    code, x = _signal(oracle, 9, fs, 4000 * (n_ep + 3), 55, -2210.0, 777.0)
    conf = dict(GPS, acq_delay_samples=777.0, acq_doppler_hz=-2200.0, acq_samplestamp_samples=0, sample_counter=0)
    d = torch.from_numpy(x.view(np.float32)).cuda()
    lin = gnsscorr.TrackingLoop(gctx, 1, 1023)
    lin.set_input_dev(0, d.data_ptr(), x.size)
    lin.start(0, _conf(gnsscorr, **conf), code)
    want = lin.run(n_ep)[0]
    lin.close()

    # a ring large enough for the whole capture: nothing is evicted, so the only thing at stake is which head a launch sees
    ring = gnsscorr.IqStream(gctx, capacity_samples=4000 * (n_ep + 8), max_window_samples=4000)
    loop = gnsscorr.TrackingLoop(gctx, 1, 1023)
    loop.set_input_stream(0, ring)
    loop.start(0, _conf(gnsscorr, **conf), code)
    st = torch.cuda.Stream()
    per_launch = 3
    blocks = [(p, min(6500, x.size - p)) for p in range(0, x.size, 6500)]
    item = gnsscorr.LOOP_RECORD_DTYPE.itemsize
    d_rec = torch.zeros(len(blocks) * per_launch * item, dtype=torch.uint8, device="cuda")
    pinned = torch.from_numpy(x.view(np.float32).copy()).pin_memory()
    busy = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for _ in range(6):
            busy = busy @ busy * 1e-3   # ~ms of queued work in front of the loop launches: they pile up behind it
    for k, (p, m) in enumerate(blocks):
        ring.push_pinned(pinned.data_ptr() + 8 * p, m)
        loop.run_dev(per_launch, d_rec.data_ptr() + k * per_launch * item, st.cuda_stream)
    torch.cuda.synchronize()
    rec = np.frombuffer(d_rec.cpu().numpy().tobytes(), gnsscorr.LOOP_RECORD_DTYPE)
    got = rec[rec["valid"] == 1][:n_ep]
    assert len(got) == n_ep
    for name in want.dtype.names:
        assert np.array_equal(got[name], want[name]), name
    loop.close()
    ring.close()
