import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "gnss-sdr-1_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, gnsscorr, bench
dev = torch.device("cuda", 0); ctx = gnsscorr.Context(0)
FS, N = 25_000_000, 25000
code = bench.gps_ca_code(1)
x, truth = bench.make_channel_stream(torch, dev, code, 70 * N, seed=5)
def eng(n_c, L, spc, veml, chip_rate, vlen, period, el, vel):
    e = gnsscorr.TrackingLoop(ctx, n_c, L)
    c = gnsscorr.LoopConf()
    for k, v in dict(fs_in=float(FS), signal_carrier_freq_hz=1575.42e6, code_chip_rate_hz=chip_rate, code_period_s=period, carrier_lock_th=0.85, code_length_chips=L // spc,
            code_samples_per_chip=spc, vector_length=vlen, pull_in_time_s=2, veml=veml, pll_filter_order=3, dll_filter_order=2, cn0_samples=20, cn0_min=25, max_lock_fail=50,
            pll_bw_hz=40.0, dll_bw_hz=2.0, fll_bw_hz=35.0, early_late_space_chips=el, very_early_late_space_chips=vel, acq_delay_samples=0.0, acq_doppler_hz=1000.0).items():
        setattr(c, k, v)
    cd = code if L == 1023 else np.sign(np.random.default_rng(1).standard_normal(L)).astype(np.float32)
    return e, c, cd, n_c
specs = [(16, 1023, 1, 0, 1.023e6, N, 0.001, 0.5, 0.0, 64), (8, 8184, 2, 1, 1.023e6, 4 * N, 0.004, 0.15, 0.6, 16), (8, 2046, 1, 0, 2.046e6, N, 0.001, 0.5, 0.0, 64)]
engs = [eng(*s[:9]) + (s[9],) for s in specs]
recs = [torch.zeros(e[3] * e[4] * gnsscorr.LOOP_RECORD_DTYPE.itemsize, dtype=torch.uint8, device=dev) for e in engs]
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
def start_all():
    for e, c, cd, n_c, n_per in engs:
        for ch in range(n_c):
            e.set_input_dev(ch, x.data_ptr(), 70 * N); e.start(ch, c, cd)
    torch.cuda.synchronize()
for which in ([0], [1], [2], [0, 1, 2]):
    for rep in range(3):
        start_all()
        evs = []
        t0 = time.perf_counter()
        for k in which:
            with torch.cuda.stream(streams[k]):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); engs[k][0].run_dev(engs[k][4], recs[k].data_ptr(), streams[k].cuda_stream); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
    print(which, "wall %.3f ms" % wall, ["%.3f" % a.elapsed_time(b) for a, b in evs])
