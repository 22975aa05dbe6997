#!/usr/bin/env python3
"""Times the BASELINE configs[3] acquisition search (32 PRNs x 41 bins x 2 dwells, N = 25000) alone: the shape of bench.py's
acquisition leg without the rest of the line, for A/B runs of library builds ($GNSSCORR_LIB) and for rocprofv3 kernel traces.

    python3 profiles/tools/acq_time.py [--sats 32] [--warm 20] [--reps 20] [--label x]

Prints one JSON line: ms per search behind 1 warm-up search ("cold", bench.py's default) and behind --warm searches ("steady")."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnss-sdr-1_amd"))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sats", type=int, default=32)
    ap.add_argument("--bins", type=int, default=41)
    ap.add_argument("--warm", type=int, default=20)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--label", default="")
    ap.add_argument("--idle-s", type=float, default=0.25)
    args = ap.parse_args()
    import time
    import torch
    import gnsscorr
    import bench
    dev = torch.device("cuda", 0)
    ctx = gnsscorr.Context(0)
    FS, N = bench.FS, bench.N_EPOCH
    acq = gnsscorr.PcpsAcquisition(ctx, args.sats, FS, 1, 1, np.float32(FS) * np.float32(0.001), 25000.0, 25, 5000, 250, max_dwells=2, use_cfar=False,
        num_doppler_bins_override=args.bins)
    idx = np.minimum((np.arange(N) * (1.023e6 / FS)).astype(np.int64), 1022)
    for s in range(args.sats):
        acq.set_local_code(s, bench.gps_ca_code(s % 32 + 1)[idx].astype(np.complex64))
    x, truth = bench.make_channel_stream(torch, dev, bench.gps_ca_code(1), 2 * N + 64, seed=5003, cn0_db_hz=47.0)
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    torch.cuda.synchronize()

    def search():
        acq.reset()
        acq.dwell_enqueue(x.data_ptr(), stream)
        acq.dwell_enqueue(x.data_ptr() + 8 * N, stream)
        acq.flush(stream)

    def timed(n_warm, reps):
        for _ in range(n_warm):
            search()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(reps):
            search()
        a1.record()
        torch.cuda.synchronize()
        return float(a0.elapsed_time(a1)) / reps

    search()
    torch.cuda.synchronize()
    time.sleep(args.idle_s)
    cold = timed(1, 5)
    steady = timed(args.warm, args.reps)
    res = acq.fetch_results(stream)
    want = ((1023 - truth["tau0"]) % 1023) * FS / 1.023e6
    d = abs(res[0].indext - want)
    ok = int(np.argmax([r.test_statistics for r in res])) == 0 and min(d, N - d) <= 26 and abs(res[0].doppler_hz - truth["doppler"]) <= 250
    alg = args.sats * args.bins * N * 44.0
    print(json.dumps({"label": args.label, "lib": os.environ.get("GNSSCORR_LIB", "libgnsscorr.so"), "sats": args.sats, "cold_ms": cold, "steady_ms": steady,
        "cold_frac": alg / (cold * 1e-3) / 8e12, "steady_frac": alg / (steady * 1e-3) / 8e12, "found": bool(ok)}))
    acq.close()


if __name__ == "__main__":
    main()
