"""GPU-side (HIP events) and host wall time of one closed-loop engine per geometry: 16 GPS L1 C/A channels x 25 Msps x 64 code periods.
Usage (GPU box): python profiles/tools/loop_slices_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnss-sdr-1_amd"))
import numpy as np, torch, gnsscorr
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
ctx = gnsscorr.Context(0)
FS, N = 25_000_000, 25000
code = bench.gps_ca_code(1)
x, truth = bench.make_channel_stream(torch, dev, code, 70 * N, seed=5)
n_ch, n_per = int(os.environ.get("N_CH", "16")), 64
delay = ((1023 - truth["tau0"]) % 1023) * FS / 1.023e6
st = torch.cuda.Stream(device=dev)
for slices in (1, 2, 4, 8, 16):
    eng = gnsscorr.TrackingLoop(ctx, n_ch, 1023)
    eng.set_geometry(slices_per_channel=slices)
    lc = gnsscorr.LoopConf()
    for k, v in dict(fs_in=float(FS), signal_carrier_freq_hz=1575.42e6, code_chip_rate_hz=1.023e6, code_period_s=0.001, carrier_lock_th=0.85, code_length_chips=1023,
            code_samples_per_chip=1, vector_length=N, pull_in_time_s=2, veml=0, pll_filter_order=3, dll_filter_order=2, cn0_samples=20, cn0_min=25, max_lock_fail=50,
            pll_bw_hz=40.0, dll_bw_hz=2.0, fll_bw_hz=35.0, early_late_space_chips=0.5, acq_delay_samples=float(np.round(delay)),
            acq_doppler_hz=float(np.round(truth["doppler"] / 10) * 10)).items():
        setattr(lc, k, v)
    recs = torch.zeros(n_ch * n_per * gnsscorr.LOOP_RECORD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    best_gpu, best_wall = 1e9, 1e9
    for rep in range(4):
        for ch in range(n_ch):
            eng.set_input_dev(ch, x.data_ptr(), 70 * N)
            eng.start(ch, lc, code)
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            eng.run_dev(n_per, recs.data_ptr(), st.cuda_stream)
            e1.record()
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        best_gpu, best_wall = min(best_gpu, e0.elapsed_time(e1)), min(best_wall, wall * 1e3)
    r = np.frombuffer(recs.cpu().numpy().tobytes(), gnsscorr.LOOP_RECORD_DTYPE).reshape(n_ch, n_per)
    print("slices %2d: GPU %.3f ms = %.2f us per period, wall %.3f ms, host enqueue %.3f ms, valid %d, doppler %.1f (truth %.1f)" % (slices, best_gpu, best_gpu * 1e3 / n_per,
        best_wall, t_host * 1e3, int(r["valid"].sum()), r["carrier_doppler_hz"][0, -1], truth["doppler"]))
    eng.close()
