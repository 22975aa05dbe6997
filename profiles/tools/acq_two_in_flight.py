import os, sys, json, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "gnss-sdr-1_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, gnsscorr, bench
dev = torch.device("cuda", 0); ctx = gnsscorr.Context(0)
FS, N = bench.FS, bench.N_EPOCH
def make():
    a = gnsscorr.PcpsAcquisition(ctx, 32, FS, 1, 1, np.float32(FS) * np.float32(0.001), 25000.0, 25, 5000, 250, max_dwells=2, use_cfar=False, num_doppler_bins_override=41)
    idx = np.minimum((np.arange(N) * (1.023e6 / FS)).astype(np.int64), 1022)
    for s in range(32): a.set_local_code(s, bench.gps_ca_code(s + 1)[idx].astype(np.complex64))
    return a
engs = [make(), make()]
x, truth = bench.make_channel_stream(torch, dev, bench.gps_ca_code(1), 2 * N + 64, seed=5003, cn0_db_hz=47.0)
sts = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
torch.cuda.synchronize()
def search(a, st):
    a.reset(); a.dwell_enqueue(x.data_ptr(), st.cuda_stream); a.dwell_enqueue(x.data_ptr() + 8 * N, st.cuda_stream); a.flush(st.cuda_stream)
def run(n_eng, reps):
    for k in range(6): search(engs[k % n_eng], sts[k % n_eng])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps): search(engs[k % n_eng], sts[k % n_eng])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for _ in range(2):
    print("one engine  : %.4f ms per search" % run(1, 40))
    print("two engines : %.4f ms per search" % run(2, 40))
for k in range(2):
    r = engs[k].fetch_results(sts[k].cuda_stream)
    print("engine", k, "found", int(np.argmax([q.test_statistics for q in r])) == 0)
