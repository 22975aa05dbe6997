#!/usr/bin/env python3
"""One RF stream through the whole front end on one MI355X: ring push -> PCPS search over all 32 GPS PRNs -> hand-over of the
detected satellites into free slots of one closed-loop tracking engine -> tracking, one launch per pushed block.

    python examples/receiver_bench.py [--fs 25000000] [--seconds 1.0] [--sats 8] [--block-ms 16]

Prints one JSON line: wall time, real-time factor for the stream, what was found and how well it was tracked.  Synthetic input
(random-init: GPS L1 C/A codes from the library's own generator, random Doppler / delay / data bits, unit-variance noise).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnss-sdr-1_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fs", type=int, default=25000000)
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--sats", type=int, default=8)
    ap.add_argument("--block-ms", type=int, default=16)
    ap.add_argument("--slots", type=int, default=12)
    args = ap.parse_args()
    import torch
    import gnsscorr
    fs, n_ms = args.fs, int(round(args.seconds * 1000))
    N = fs // 1000
    dev = torch.device("cuda:0")
    rng = np.random.Generator(np.random.PCG64(2024))
    prns = sorted(rng.choice(np.arange(1, 33), args.sats, replace=False).tolist())
    truth = {p: dict(doppler=float(rng.uniform(-4500, 4500)), delay=int(rng.integers(0, N)), cn0=float(rng.uniform(47, 51))) for p in prns}

    # ---- synthetic stream, built on the GPU one second at a time (float32 phase would not do: float64 for the carriers) ----
    n = N * n_ms
    t = torch.arange(n, device=dev, dtype=torch.float64)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    x = torch.complex(torch.randn(n, device=dev, generator=gen), torch.randn(n, device=dev, generator=gen)) * (0.5 ** 0.5)
    for p in prns:
        tr = truth[p]
        code = torch.from_numpy(gnsscorr.gps_l1_ca_code_gen_float(p)).to(dev)
        rate = 1.023e6 * (1 + tr["doppler"] / 1575.42e6) / fs
        ph = (1023.0 - tr["delay"] * 1.023e6 / fs) + t * rate
        chip = torch.remainder(torch.floor(ph), 1023).long()
        bits = torch.from_numpy(rng.integers(0, 2, n_ms // 20 + 2) * 2.0 - 1.0).to(dev)
        sym = bits[torch.clamp((torch.floor(ph / 1023.0) // 20).long(), 0, bits.numel() - 1)]
        amp = (10 ** (tr["cn0"] / 10) / fs) ** 0.5
        phase = torch.remainder(tr["doppler"] * t / fs, 1.0) * (2 * np.pi)
        x += (amp * code[chip] * sym) * torch.complex(torch.cos(phase), torch.sin(phase)).to(torch.complex64)
    host = torch.view_as_real(x.to(torch.complex64)).cpu().pin_memory()
    del x, t
    torch.cuda.synchronize()

    ctx = gnsscorr.Context(0)
    block = N * args.block_ms
    ring = gnsscorr.IqStream(ctx, block * 8, 2 * N)
    acq = gnsscorr.PcpsAcquisition(ctx, 32, fs, 1, 1, np.float32(fs) * np.float32(0.001), float(N), max(1, int(np.ceil(fs / 1.023e6))), 5000, 100,
        max_dwells=1, use_cfar=True)  # 100 Hz bins: the hand-over error stays inside the PLL's pull-in range
    for s in range(32):
        acq.set_local_code(s, gnsscorr.gps_l1_ca_code_gen_complex_sampled(s + 1, fs))
    loop = gnsscorr.TrackingLoop(ctx, args.slots, 1023)
    for ch in range(args.slots):
        loop.set_input_stream(ch, ring)
    conf = dict(fs_in=float(fs), signal_carrier_freq_hz=1575.42e6, code_chip_rate_hz=1.023e6, code_period_s=0.001, carrier_lock_th=0.85, code_length_chips=1023,
        code_samples_per_chip=1, vector_length=N, pull_in_time_s=1, veml=0, pll_filter_order=3, dll_filter_order=2, enable_fll_pull_in=0, enable_fll_steady_state=0,
        cn0_samples=20, cn0_min=25, max_lock_fail=50, pll_bw_hz=50.0, dll_bw_hz=2.0, fll_bw_hz=10.0, early_late_space_chips=0.5, very_early_late_space_chips=0.0)
    sync = gnsscorr.loop_sync_for_signal("G", "1C", 1)

    slot_of, records, launches, searches = {}, {}, 0, 0
    hbase = host.data_ptr()
    t0 = time.perf_counter()
    for b in range(n // block):
        ring.push_pinned(hbase + b * block * 8, block)
        if b == 0:
            # cold start: search every PRN on the first milliseconds of the stream (2 dwells would be the BASELINE search; one is enough here)
            res = acq.dwell_stream(ring, 0)
            searches += 1
            # max_to_input_power statistic: a noise cell averages 1 / N, the largest of 32 x bins x N cells about 18 / N; a 47 dB-Hz signal reaches 50 / N in 1 ms
            found = [(s + 1, r) for s, r in enumerate(res) if r.test_statistics > 30.0 / N]
            found.sort(key=lambda pr: -pr[1].test_statistics)
            for prn, r in found[:args.slots]:
                ch = len(slot_of)
                c = gnsscorr.LoopConf()
                for k_, v_ in dict(conf, acq_delay_samples=float(r.acq_delay_samples), acq_doppler_hz=float(r.acq_doppler_hz), acq_samplestamp_samples=0,
                        sample_counter=0).items():
                    setattr(c, k_, v_)
                loop.set_sync(ch, sync)
                loop.start(ch, c, gnsscorr.gps_l1_ca_code_gen_float(prn))
                slot_of[prn] = ch
                records[prn] = []
        if slot_of:
            rec = loop.run(args.block_ms + 1)
            launches += 1
            for prn, ch in slot_of.items():
                records[prn].append(rec[ch][rec[ch]["valid"] == 1])
    ring.synchronize()
    wall = time.perf_counter() - t0
    report = {}
    for prn in sorted(slot_of):
        r = np.concatenate(records[prn])
        tr = truth.get(prn)
        tail = r[-200:]
        report[prn] = dict(periods=int(r.size), doppler_hz=float(tail["carrier_doppler_hz"].mean()), cn0_db_hz=float(tail["cn0_db_hz"][-1]),
            truth_doppler_hz=None if tr is None else tr["doppler"], truth_cn0_db_hz=None if tr is None else tr["cn0"])
    ok = sorted(slot_of) == prns and all(abs(v["doppler_hz"] - v["truth_doppler_hz"]) < 5.0 and v["periods"] > n_ms - 40 for v in report.values())
    print(json.dumps(dict(stream_msps=fs / 1e6, seconds=args.seconds, satellites_in_stream=prns, detected=sorted(slot_of), wall_s=wall,
        realtime_factor=args.seconds / wall, launches=launches, searches=searches, all_found_and_tracked=bool(ok), channels=report)))
    loop.close()
    acq.close()
    ring.close()
    ctx.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
