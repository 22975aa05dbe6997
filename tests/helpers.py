"""Shared test helpers: seeded synthetic IQ and error metrics."""
import numpy as np

GPS_L1_FREQ_HZ = 1575.42e6


def synth_stream(codes, fs, n_samples, seed, cn0_db_hz=(38.0, 48.0), chip_rate=1.023e6, doppler_max=5000.0,
        noise=True, carrier_freq=GPS_L1_FREQ_HZ):
    """x[n] = sum_s A_s c_s(tau_s(n)) exp(j(2 pi f_s n/fs + phi_s)) + w[n], w ~ CN(0,1).
    codes: list of +-1 arrays (one per satellite, any length L, chips at `chip_rate`*len/1023...).
    Returns (x complex64, truth list of dicts)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = np.arange(n_samples, dtype=np.float64)
    x = np.zeros(n_samples, np.complex128)
    truth = []
    for code in codes:
        L = len(code)
        cn0 = rng.uniform(*cn0_db_hz)
        amp = np.sqrt(10.0 ** (cn0 / 10.0) / fs)
        fd = rng.uniform(-doppler_max, doppler_max)
        tau0 = rng.uniform(0, L)
        phi = rng.uniform(0, 2 * np.pi)
        rate = chip_rate * (1.0 + fd / carrier_freq)  # chips (code samples) per second
        chip = np.floor(tau0 + n * (rate / fs)).astype(np.int64) % L
        x += amp * np.asarray(code, np.float64)[chip] * np.exp(1j * (2 * np.pi * fd * n / fs + phi))
        truth.append(dict(cn0=cn0, amp=amp, doppler=fd, tau0=tau0, phi=phi, code_rate=rate))
    if noise:
        x += (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples)) * np.sqrt(0.5)
    return x.astype(np.complex64), truth


def open_loop_params(truth, fs, L, n_samples_epoch, n_epochs, carr_offset_hz=0.0):
    """Per-epoch scalar arguments as dll_pll_veml_tracking would hand them to the correlator
    (do_correlation_step, dll_pll_veml_tracking.cc:886-897), propagated open-loop from the truth.
    Returns list of dicts with float32 values and the window start."""
    out = []
    step_chips = truth["code_rate"] / fs
    fd = truth["doppler"] + carr_offset_hz
    for k in range(n_epochs):
        start = k * n_samples_epoch
        code_phase = (truth["tau0"] + start * step_chips) % L  # chip position of the first sample
        # the correlator computes index = floor(step*n + shift - rem): rem = -code_phase (mod L)
        rem = -code_phase
        if rem < -L / 2:
            rem += L
        carr_phase = (truth["phi"] + 2 * np.pi * fd * start / fs) % (2 * np.pi)
        out.append(dict(sample_offset=start,
            rem_carr=np.float32(carr_phase), phase_step=np.float32(2 * np.pi * fd / fs),
            rem_code=np.float32(rem), code_step=np.float32(step_chips), n=n_samples_epoch))
    return out


def rel_err(got, ref, prompt_index):
    """max_t |got[t] - ref[t]| / |P_ref| (SURVEY.md section 8d parity metric)."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    return float(np.max(np.abs(got - ref)) / np.abs(ref[prompt_index]))


def gsoc_grid_statistics(grid_mag2, fs, doppler_step):
    """The figures the reference's plot_acq_grid_gsoc.m prints for an acquisition grid (src/utils/matlab/plot_acq_grid_gsoc.m:
    acq_grid = abs(complex correlation output); peak; noise floor = mean of the grid outside +-floor(3 fs / 1.023e6) samples and
    +-floor(500 / step) bins around the peak; gain = 10 log10(peak / floor)).  grid_mag2: [bins][N] of |.|^2 as the engine and the
    block hold it.  Returns (peak magnitude, noise floor, gain in dB, row, column)."""
    g = np.sqrt(np.asarray(grid_mag2, np.float64))
    row, col = np.unravel_index(int(np.argmax(g)), g.shape)
    ds, dp = int(np.floor(3 * fs / 1.023e6)), int(np.floor(500 / doppler_step))
    ng = g.copy()
    ng[max(row - dp, 0):row + dp + 1, max(col - ds, 0):col + ds + 1] = 0.0
    n = ng.size - (2 * ds + 1) * (2 * dp + 1)
    floor_ = ng.sum() / n
    return float(g[row, col]), float(floor_), float(10 * np.log10(g[row, col] / floor_)), int(row), int(col)
