"""Randomised differential test of the tracking multicorrelator against the oracle: many (window, NCO, tap, code, format,
mode) combinations per launch.  The default run is a few thousand channel-epochs; GNSSCORR_FUZZ_BATCHES=N widens it."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _one_batch(gctx, oracle, rng, fmt_name, high_dyn):
    import gnsscorr
    import torch
    n_taps = int(rng.integers(1, 9))
    L = int(rng.choice([1, 2, 31, 511, 1023, 2046, 4092, 8184, int(rng.integers(3, 12000))]))
    code = np.sign(rng.standard_normal(L)).astype(np.float32)
    code[code == 0] = 1
    shifts = np.sort(rng.uniform(-3.0, 3.0, n_taps)).astype(np.float32)
    if high_dyn:
        # the reference's high-dynamics resampler needs ordered taps whose sample delays stay inside the window
        shifts = np.sort(rng.uniform(-1.0, 1.0, n_taps)).astype(np.float32)
    n_sig = 16384
    raw = rng.standard_normal((n_sig, 2))
    fmt = getattr(gnsscorr, fmt_name)
    if fmt_name == "GC_IQ_F32":
        q = raw.astype(np.float32)
        sig = q.reshape(-1).view(np.complex64)
    elif fmt_name == "GC_IQ_I16":
        q = np.round(raw * 500).astype(np.int16)
        sig = q.astype(np.float32).reshape(-1).view(np.complex64)
    else:
        q = np.clip(np.round(raw * 30), -128, 127).astype(np.int8)
        sig = q.astype(np.float32).reshape(-1).view(np.complex64)
    d = torch.from_numpy(q).cuda()
    n_epochs = 48
    recs, refs = [], []
    for _ in range(n_epochs):
        n = int(rng.choice([0, 1, 2, 17, 255, 256, 257, 511, 512, 513, 1000, 2047, 4000, int(rng.integers(16 if high_dyn else 0, 6000))]))
        if high_dyn and n < 64:
            n = 64 + n
        off = int(rng.integers(0, n_sig - n + 1))
        rem_carr = float(np.float32(rng.uniform(-7, 7)))
        pstep = float(np.float32(rng.uniform(-3.1, 3.1) if rng.random() < 0.3 else rng.uniform(-0.02, 0.02)))
        rem_code = float(np.float32(rng.uniform(-3 * L, 3 * L) if rng.random() < 0.3 else rng.uniform(-2, 2)))
        cstep = float(np.float32(rng.uniform(0.0, 3.0) if rng.random() < 0.3 else rng.uniform(0.01, 0.6)))
        prate = float(np.float32(rng.uniform(-1e-7, 1e-7))) if high_dyn else 0.0
        crate = float(np.float32(rng.uniform(0.0, 1e-9))) if high_dyn else 0.0
        if high_dyn:
            # tap delays in samples must be < n (the reference memcpy()s n - delay floats)
            while n > 0 and (shifts[-1] - shifts[0]) / max(cstep, 1e-6) + n_taps >= n:
                cstep = float(np.float32(cstep * 2 + 0.05))
        recs.append(gnsscorr.epoch_params(off, rem_carr, pstep, rem_code, cstep, n, carr_phase_rate_step_rad=prate, code_phase_rate_step_chips=crate))
        refs.append(oracle.multicorrelator(sig[off:], code, shifts, np.float32(rem_carr), np.float32(pstep), np.float32(rem_code), np.float32(cstep), n,
            phase_rate_step=np.float32(prate), code_rate_step=np.float32(crate), high_dyn=high_dyn))
    b = gnsscorr.TrackingBatch(gctx, 1, n_taps, L, high_dyn=high_dyn)
    if fmt != gnsscorr.GC_IQ_F32:
        b.set_input_format(fmt)
    b.set_code(0, code, shifts)
    b.set_input_dev(0, d.data_ptr(), n_sig)
    if rng.random() < 0.3:
        b.set_slices(int(rng.integers(2, 9)))
    out = b.run(n_epochs, gnsscorr.epoch_params_array(recs))[0]
    b.close()
    scale = {"GC_IQ_F32": 1.0, "GC_IQ_I16": 500.0, "GC_IQ_I8": 30.0}[fmt_name]
    worst = 0.0
    for k in range(n_epochs):
        n = recs[k].n_samples
        tol = 3e-5 * scale * np.sqrt(max(n, 1)) + 1e-4 * float(np.max(np.abs(refs[k]))) if n else 0.0
        err = float(np.max(np.abs(out[k] - refs[k])))
        assert err <= tol, (fmt_name, high_dyn, n_taps, L, k, n, recs[k].sample_offset, shifts, out[k], refs[k])
        worst = max(worst, err / tol if tol else 0.0)
    return worst


def test_randomised_open_loop_parity(gctx, oracle):
    n_batches = int(os.environ.get("GNSSCORR_FUZZ_BATCHES", "60"))
    rng = np.random.Generator(np.random.PCG64(20261004))
    worst = 0.0
    for i in range(n_batches):
        fmt = ["GC_IQ_F32", "GC_IQ_F32", "GC_IQ_I16", "GC_IQ_I8"][i % 4]
        worst = max(worst, _one_batch(gctx, oracle, rng, fmt, high_dyn=(i % 5 == 4)))
    print("worst error / tolerance over %d channel-epochs: %.3f" % (n_batches * 48, worst))
