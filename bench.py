#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X correlator engine.

Metric (BASELINE.json): Msamples/s tracked by the N-channel multicorrelator
(+ acquisition dwells/s, + achieved HBM-bandwidth fraction).

Workload at N=1 (BASELINE.json configs[1]): GPS L1 C/A, 32 channels, 25 Msps,
3-tap E/P/L multicorrelator, one code period (25000 samples) per channel-epoch.
A "step" = one pass of the tracking hot path over one batch: 32 channels x
EPOCHS epochs, i.e. ONE launch of the batched kernel, with IQ, per-epoch
parameters and code tables already resident in HBM.  Every channel reads its
OWN IQ buffer ("distinct-input" mode) so that the
algorithmic bytes (8 B per channel-sample) are real HBM traffic and the HBM
roofline is meaningful; the shared-stream figure (all 32 channels on one RF
stream, served from L2 / Infinity Cache) is reported beside it, not as `value`.

Multi-GPU (driver: python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...): channels shard over ranks (32 per GPU, BASELINE configs[4]), no
data-path collective; RCCL is used only for the barrier and the max-over-ranks
of the elapsed time.  scaling = "weak".  Behind the headline region every rank
also times ITS share of BASELINE configs[4] (16 GPS L1 C/A + 8 Galileo E1 5-tap
+ 8 BeiDou B1I channels, open loop) and its PRNs of the configs[3] search
(PRN i on GPU (i - 1) mod G); both travel in `per_gpu[]`, their sums in
`hybrid_all_gpus` / `acquisition_all_gpus`.

The headline is the contract's region: W warm-up steps, then K timed steps
(no pre-roll by default).  On this chip that region lies inside a start-of-load
power transient; the figure after ~10 ms of load is reported beside it as
`roofline.steady_state_frac` (and `--preroll-ms 15` moves the timed region there).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "gnss-sdr-1_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

# The HIP runtime maps streams onto 4 hardware queues by default; the three closed-loop engines of the cfg5 share then land two on one
# queue and run one after the other (1.37 ms instead of 0.89 ms for 64 ms of signal, profiles/tools/loop_share_overlap.py).  Read by the
# runtime when it initialises; named in the JSON line (`runtime_env`).  The single-stream measurements do not depend on it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FS = 25_000_000
N_EPOCH = 25000          # samples per code period at 25 Msps
N_CHANNELS = 32          # per GPU
N_TAPS = 3
CODE_LEN = 1023
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# <NTAPS, HDR, HDC, FMT = GC_IQ_F32, CC, SC16>: the name rocprofv3 prints for the headline kernel
TRK_KERNEL_NAME = "trk_multicorrelator_kernel<3, false, false, 0, false, false, false>"


def gps_ca_code(prn):
    """GPS L1 C/A code (IS-GPS-200 G1/G2 generators), +-1 floats.  Bench-local
    generator so that the timed path does not depend on the oracle."""
    taps = {1: (2, 6), 2: (3, 7), 3: (4, 8), 4: (5, 9), 5: (1, 9), 6: (2, 10), 7: (1, 8), 8: (2, 9), 9: (3, 10),
        10: (2, 3), 11: (3, 4), 12: (5, 6), 13: (6, 7), 14: (7, 8), 15: (8, 9), 16: (9, 10), 17: (1, 4), 18: (2, 5),
        19: (3, 6), 20: (4, 7), 21: (5, 8), 22: (6, 9), 23: (1, 3), 24: (4, 6), 25: (5, 7), 26: (6, 8), 27: (7, 9),
        28: (8, 10), 29: (1, 6), 30: (2, 7), 31: (3, 8), 32: (4, 9)}
    g1 = [1] * 10
    g2 = [1] * 10
    a, b = taps[prn]
    out = np.empty(1023, np.float32)
    for i in range(1023):
        chip = g1[9] ^ g2[a - 1] ^ g2[b - 1]
        out[i] = 1.0 if chip else -1.0
        f1 = g1[2] ^ g1[9]
        f2 = g2[1] ^ g2[2] ^ g2[5] ^ g2[7] ^ g2[8] ^ g2[9]
        g1 = [f1] + g1[:9]
        g2 = [f2] + g2[:9]
    return out


def make_channel_stream(torch, dev, code, n_samples, seed, cn0_db_hz=None):
    """Seeded synthetic IQ for one channel: unit-variance complex noise + one
    PRN at C/N0 in [38, 48] dB-Hz, Doppler in +-5 kHz."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cn0 = rng.uniform(38.0, 48.0)
    if cn0_db_hz is not None:
        cn0 = cn0_db_hz
    amp = float(np.sqrt(10.0 ** (cn0 / 10.0) / FS))
    fd = float(rng.uniform(-5000.0, 5000.0))
    tau0 = float(rng.uniform(0, CODE_LEN))
    phi = float(rng.uniform(0, 2 * np.pi))
    rate = 1.023e6 * (1.0 + fd / 1575.42e6)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.randn(n_samples, 2, device=dev, generator=g, dtype=torch.float32) * float(np.sqrt(0.5))
    n = torch.arange(n_samples, device=dev, dtype=torch.float64)
    chip = torch.floor(tau0 + n * (rate / FS)).to(torch.int64) % CODE_LEN
    c = torch.from_numpy(code).to(dev)[chip]
    ph = (2 * np.pi * fd / FS) * n + phi
    x[:, 0] += (amp * c * torch.cos(ph).to(torch.float32))
    x[:, 1] += (amp * c * torch.sin(ph).to(torch.float32))
    truth = dict(amp=amp, doppler=fd, tau0=tau0, phi=phi, code_rate=rate)
    return x.contiguous(), truth


def epoch_records(gnsscorr, truth, n_epochs):
    """Open-loop per-epoch arguments as do_correlation_step would pass them
    (dll_pll_veml_tracking.cc:886-897), one window per code period."""
    recs = []
    step = truth["code_rate"] / FS
    for k in range(n_epochs):
        start = k * N_EPOCH
        code_phase = (truth["tau0"] + start * step) % CODE_LEN
        rem = -code_phase
        if rem < -CODE_LEN / 2:
            rem += CODE_LEN
        carr = (truth["phi"] + 2 * np.pi * truth["doppler"] * start / FS) % (2 * np.pi)
        recs.append(gnsscorr.epoch_params(start, float(np.float32(carr)), float(np.float32(2 * np.pi * truth["doppler"] / FS)),
            float(np.float32(rem)), float(np.float32(step)), N_EPOCH))
    return recs


def cpu_baseline(codes_np, shifts, sample_sig, sample_recs, budget_s):
    """The oracle (validated CPU restatement of the volk_gnsssdr generic path,
    -O3 -march=native, 1 thread) timed on a bounded sample of the same workload."""
    from oracle import Oracle
    orc = Oracle(native=True)
    sig = sample_sig
    n_done = 0
    t0 = time.perf_counter()
    while True:
        for rec in sample_recs:
            # same scalars the GPU epoch was built from
            orc.multicorrelator(sig[rec[0]:], codes_np, shifts, rec[1], rec[2], rec[3], rec[4], N_EPOCH)
            n_done += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    return n_done * N_EPOCH / dt / 1e6, n_done, dt


def cpu_baseline_threads(codes_np, shifts, sample_sig, sample_recs, budget_s, n_threads, rate_1t):
    """One thread per channel like the reference's own timing test
    (cpu_multicorrelator_real_codes_test.cc:126-146); every thread spends its whole budget inside one C call."""
    import threading
    from oracle import Oracle
    orc = Oracle(native=True)
    n_iter = max(1, int(rate_1t * 1e6 / N_EPOCH * budget_s))  # channel-epochs one thread finishes in budget_s
    rec = sample_recs[0]

    def work():
        orc.multicorrelator_repeat(n_iter, sample_sig[rec[0]:], codes_np, shifts, rec[1], rec[2], rec[3], rec[4], N_EPOCH)

    threads = [threading.Thread(target=work) for _ in range(n_threads)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    return n_threads * n_iter * N_EPOCH / dt / 1e6, n_threads * n_iter, dt


def load_valu_ceilings():
    """profiles/r04_valu_ceilings.json (profiles/tools/valu_ceiling.py): VALU-issue ceiling of each kernel's interior loop from its
    disassembled instruction mix x the issue rates measured on the chip x 1024 SIMDs at the nominal clock."""
    for name in ("r04_valu_ceilings.json", "r03_valu_ceilings.json"):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
        except Exception:
            continue
    return {}


def mode_roofline(ceil, parts, ms, serial_us_per_unit=None):
    """Roofline of one bench entry that is not simply HBM-bound.  parts: [(samples, bytes_per_sample_from_hbm or 0, kernel key,
    share of the chip's SIMDs the launch can use)]; the launches of an entry run back to back, so the entry's ceiling time is the sum
    over parts of max(HBM time, VALU time) (+ a serial term for the closed loop).  Everything in Msamples/s so that the two bounds
    are comparable; `bound` names the resource whose ceiling is lower."""
    t_hbm = t_valu = t_bound = 0.0
    n = 0
    missing = False
    for samples, bps, key, share in parts:
        n += samples
        th = samples * bps / (HBM_PEAK_GBPS * 1e9)
        k = ceil.get(key)
        if k is None:
            missing = True
            tv = 0.0
        else:
            tv = samples / (k["ceiling_msamples_s"] * 1e6 * share)
        t_hbm += th
        t_valu += tv
        t_bound += max(th, tv)
    if serial_us_per_unit:
        t_bound += serial_us_per_unit * 1e-6
    achieved = n / (ms * 1e-3) / 1e6
    out = {"bound": "hbm" if t_hbm >= t_valu else "valu", "achieved": achieved, "unit": "Msamples/s",
        "ceiling": n / t_bound / 1e6 if t_bound > 0 else None, "frac": t_bound / (ms * 1e-3),
        "hbm_frac": t_hbm / (ms * 1e-3), "valu_frac": None if missing else t_valu / (ms * 1e-3),
        "ceiling_source": "max(HBM: bytes / 8 TB/s, VALU: profiles/r04_valu_ceilings.json = interior-loop instruction mix (hipcc -S) x "
                          "profiles/r02_valu_issue_rates.txt x 1024 SIMDs x 2.4 GHz) per launch, summed; counters: profiles/r04_trk_sq_counters.json"}
    if serial_us_per_unit:
        out["bound"] = out["bound"] + " + one-lane loop maths"
        out["serial_us"] = serial_us_per_unit
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--epochs", type=int, default=256, help="code periods per channel per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-acq", action="store_true")
    ap.add_argument("--acq-reps", type=int, default=5, help="timed acquisition searches (profiles/collect.sh counts on this)")
    ap.add_argument("--acq-warmup", type=int, default=1, help="untimed acquisition searches in front of the timed ones (named in the JSON; "
        "the steady-state figure behind 20 of them is reported beside it as `steady_state`)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-shared", action="store_true", help="skip the shared-stream extra (clean rocprof runs)")
    ap.add_argument("--segments", type=int, default=24, help="untimed diagnostic pass behind the timed region: this many segments of "
        "5 back-to-back launches, one HIP event between segments (0: skip)")
    ap.add_argument("--preroll-ms", type=float, default=0.0, help="untimed launches of the same step IN FRONT of the W warm-up steps, "
        "about this many ms of GPU time (named in the JSON line as `preroll`).  Default 0: the headline is the contract's region, W "
        "warm-up steps then K timed ones; the steady-state figure is reported beside it as roofline.steady_state_frac")
    args = ap.parse_args()

    import torch
    import gnsscorr

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only path")
    # rehearsal knobs (not used by the driver): several ranks on ONE GPU with gloo, to exercise the N > 1 code path
    force_dev = os.environ.get("BENCH_FORCE_DEVICE")
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = int(force_dev) if force_dev is not None else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # BENCH_FORCE_DIST=1 (rehearsal, tests/test_bench_dist_gpu.py): initialise the process group for ONE rank too, so that the
    # RCCL branch -- init with device_id, barrier, MAX all-reduce of the elapsed time -- runs on a single-GPU box
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else None

    ctx = gnsscorr.Context(dev_index)
    E = args.epochs
    n_stream = E * N_EPOCH + 64
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    # global channel g runs on GPU g mod G; it tracks PRN (g % 32) + 1 on its own IQ buffer
    from gnsscorr import sharding
    my_channels = sharding.weak_shard(N_CHANNELS, world, rank)
    streams, params, truths, codes = [], [], [], []
    batch = gnsscorr.TrackingBatch(ctx, N_CHANNELS, N_TAPS, CODE_LEN)
    if os.environ.get("BENCH_SLICES"):  # tuning knob: workgroups per channel-epoch (default: the engine decides, 1 here)
        batch.set_slices(int(os.environ["BENCH_SLICES"]))
    for ch, gid in enumerate(my_channels):
        prn = gid % 32 + 1
        code = gps_ca_code(prn)
        x, truth = make_channel_stream(torch, dev, code, n_stream, seed=1002 + 7 * gid)
        stagger = int(os.environ.get("BENCH_STAGGER_SAMPLES", "0"))  # experiment: start channel ch's buffer ch * stagger samples into its allocation
        if stagger:
            hold = torch.empty(n_stream + ch * stagger + 16, 2, device=dev, dtype=torch.float32)
            hold[ch * stagger:ch * stagger + n_stream] = x
            x = hold[ch * stagger:ch * stagger + n_stream]
        streams.append(x)
        truths.append(truth)
        codes.append(code)
        batch.set_code(ch, code, shifts)
        batch.set_input_dev(ch, x.data_ptr(), n_stream)
        params.append(epoch_records(gnsscorr, truth, E))
    batch.set_nominal_length(N_EPOCH)
    h_params = gnsscorr.epoch_params_array(params)
    d_params = torch.from_numpy(h_params.view(np.uint8)).to(dev)
    d_out = torch.zeros(N_CHANNELS * E * N_TAPS, 2, device=dev, dtype=torch.float32)
    # a real (non-null) HIP stream shared by the launches and the timing events
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    def step():
        batch.run_dev(E, d_params.data_ptr(), d_out.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()  # inputs were generated on the default stream
    # Pre-roll (NAMED in the JSON line: `preroll`): a receiver tracks continuously, a 20-step timed region is 5 ms.  On this chip the
    # first ~8 ms of load after an idle period are a power-management transient (launches at 240-245 us for 3 ms, then 260-300 us
    # for 5 ms, then the steady 245-250 us; profiles/r03_tracking_kernel_trace.csv), and W = 5 warm-up steps end 1.3 ms into it.
    # The pre-roll runs the SAME step untimed until the transient is over, then the W warm-up steps and the barrier follow as the
    # contract says; the timed region itself is untouched (K full steps, nothing skipped).  --preroll-ms 0 gives the cold-start
    # figure, which the diagnostic pass below reports as well (`cold_start_kernel_ms`).
    n_preroll = int(round(args.preroll_ms / 0.25)) if args.preroll_ms > 0 else 0
    for _ in range(n_preroll):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events on the launch stream around the WHOLE timed region: the kernel's average launch duration is their distance / K.  (A
    # pair of events per step -- rounds 1 and 2a -- puts two event packets between consecutive launches: ~12 us of idle GPU per
    # step, 0.262 instead of 0.250 ms, inside the wall-clock region as well.)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed_local, dist, red_dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = float(ev0.elapsed_time(ev1)) / args.steps

    # ---- diagnostic pass, OUTSIDE the timed region and after it (it changes neither `value` nor `ms_per_step` nor `kernel_ms`):
    # the same lead-in as the timed region (W warm-up launches from an idle GPU, synchronise), then SEG_LEN x n_seg back-to-back
    # launches with ONE event between segments.  The first ceil(K / SEG_LEN) segments re-enact the timed region; the later ones
    # show where the launch time settles (`steady_state_kernel_ms` = median of the second half) -- so the line itself says
    # whether the timed region sat in the start-of-load transient of the box it ran on.
    SEG_LEN = 5
    COLD_IDLE_S = 0.25
    segments_ms, steady_ms, cold_ms = None, None, None
    if args.segments > 0:
        torch.cuda.synchronize()
        time.sleep(COLD_IDLE_S)  # let the chip fall back to its idle power state: this pass starts cold, with NO pre-roll
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.segments + 1)]
        evs[0].record()
        for sgi in range(args.segments):
            for _ in range(SEG_LEN):
                step()
            evs[sgi + 1].record()
        torch.cuda.synchronize()
        segments_ms = [float(evs[i].elapsed_time(evs[i + 1])) / SEG_LEN for i in range(args.segments)]
        tail = sorted(segments_ms[args.segments // 2:])
        steady_ms = tail[len(tail) // 2]
        # what the timed region would have measured without the pre-roll: the first K launches of this cold pass
        n_cold = min(args.segments, max(1, -(-args.steps // SEG_LEN)))
        cold_ms = sum(segments_ms[:n_cold]) / n_cold

    samples_per_step = N_CHANNELS * E * N_EPOCH  # per GPU
    value = sharding.aggregate_throughput(samples_per_step, args.steps, world, elapsed) / 1e6
    alg_bytes = 8.0 * samples_per_step
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    # per-GPU figures (north_star: "per-GPU Msamples/s and achieved HBM-bandwidth fraction at 1/2/4/8 GPUs"): every rank's own
    # wall time over the timed steps and its own kernel time, gathered on all ranks; rank 0 prints them
    mine = {"rank": rank, "device": dev_index, "elapsed_s": elapsed_local, "kernel_ms": kernel_ms,
        "msamples_s": samples_per_step * args.steps / elapsed_local / 1e6,
        "hbm_frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "steady_state_kernel_ms": steady_ms,
        "steady_state_hbm_frac": (alg_bytes / (steady_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if steady_ms else None,
        "cold_start_kernel_ms": cold_ms}

    vceil = load_valu_ceilings()

    # ---- BASELINE configs[4], one GPU's share: 32 channels = 16 GPS L1 C/A + 8 Galileo E1 (5 taps, 4 ms) + 8 BeiDou B1I, open loop.
    # EVERY rank times its own share (gnss_flowgraph.cc:496-499: every channel hangs on the one conditioner output; here channel g of
    # the 256 runs on GPU g mod G with 32 per GPU, and each GPU holds the same mix) ----
    def run_hybrid():
        rng_h = np.random.Generator(np.random.PCG64(1005 + rank))
        hb = []  # (batch, n_epochs, params, out, samples)
        def add_group(first_ch, n_ch, n_taps, L, n_len, step_chips, hshifts):
            b = gnsscorr.TrackingBatch(ctx, n_ch, n_taps, L)
            n_ep = max(1, (E * N_EPOCH) // n_len)
            recs = []
            for k in range(n_ch):
                b.set_code(k, np.sign(rng_h.standard_normal(L)).astype(np.float32), hshifts)
                b.set_input_dev(k, streams[first_ch + k].data_ptr(), n_stream)
                recs.append([gnsscorr.epoch_params(e * n_len, 0.1, 1e-3, 0.3, float(np.float32(step_chips)), n_len) for e in range(n_ep)])
            b.set_nominal_length(n_len)
            if os.environ.get("BENCH_SLICES"):
                b.set_slices(int(os.environ["BENCH_SLICES"]))
            d_p = torch.from_numpy(gnsscorr.epoch_params_array(recs).view(np.uint8)).to(dev)
            d_o = torch.zeros(n_ch * n_ep * n_taps, 2, device=dev, dtype=torch.float32)
            hb.append((b, n_ep, d_p, d_o, n_ch * n_ep * n_len))
        add_group(0, 16, 3, 1023, N_EPOCH, 1.023e6 / FS, np.array([-0.5, 0.0, 0.5], np.float32))
        add_group(16, 8, 5, 8184, 4 * N_EPOCH, 2.046e6 / FS, np.array([-1.2, -0.3, 0.0, 0.3, 1.2], np.float32))
        add_group(24, 8, 3, 2046, N_EPOCH, 2.046e6 / FS, np.array([-0.5, 0.0, 0.5], np.float32))
        # the three signal groups are independent engines: each launches on a stream of its own (as a receiver's per-signal tracking
        # groups would), so that the tail of one group's launch overlaps the head of another's; BENCH_HYBRID_ONE_STREAM=1: one stream
        one_stream = os.environ.get("BENCH_HYBRID_ONE_STREAM") == "1"
        hstreams = [tstream] + ([] if one_stream else [torch.cuda.Stream(device=dev) for _ in range(2)])
        def hybrid_step():
            for k_, (b, n_ep, d_p, d_o, _) in enumerate(hb):
                b.run_dev(n_ep, d_p.data_ptr(), d_o.data_ptr(), hstreams[k_ % len(hstreams)].cuda_stream)
        for _ in range(2):
            hybrid_step()
        torch.cuda.synchronize()
        h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0.record(tstream)
        for hs in hstreams[1:]:
            hs.wait_event(h0)
        for _ in range(args.steps):
            hybrid_step()
        for hs in hstreams[1:]:
            je = torch.cuda.Event()
            je.record(hs)
            tstream.wait_event(je)
        h1.record(tstream)
        torch.cuda.synchronize()
        hyb_ms = h0.elapsed_time(h1) / args.steps
        hyb_samples = sum(g[4] for g in hb)
        res = {"value": hyb_samples / (hyb_ms * 1e-3) / 1e6, "unit": "Msamples/s", "ms_per_step": hyb_ms,
            "hbm_gbps": 8.0 * hyb_samples / (hyb_ms * 1e-3) / 1e9, "realtime_factor_32ch": hyb_samples / (hyb_ms * 1e-3) / (32 * FS),
            "note": "one GPU's share of the 256-channel hybrid: 16 GPS L1 C/A (3 taps) + 8 Galileo E1 (5 taps, L = 8184, 4 ms) + 8 BeiDou B1I "
                    "(3 taps, L = 2046), 25 Msps, distinct IQ buffer per channel, three launches per step%s" % (" on one stream" if one_stream else ", one stream per signal group"),
            "roofline": mode_roofline(vceil, [(hb[0][4], 8.0, "gps_l1_3tap_f32", 1.0), (hb[1][4], 8.0, "galileo_5tap_f32", 1.0),
                (hb[2][4], 8.0, "gps_l1_3tap_f32", 1.0)], hyb_ms)}
        for g in hb:
            g[0].close()
        return res

    # ---- acquisition: BASELINE configs[3], 32 PRNs x 41 bins x 2 dwells @ 25 Msps; with G GPUs the (PRN, dwell) jobs shard like the
    # channels: this rank searches PRNs rank + 1, rank + 1 + G, ... on its first RF stream ----
    def run_acquisition(prn_ids, warmups):
        n_sat, n_bins_acq, n_dw = len(prn_ids), 41, 2
        acq = gnsscorr.PcpsAcquisition(ctx, n_sat, FS, 1, 1, np.float32(FS) * np.float32(0.001), 25000.0, 25,
            5000, 250, max_dwells=n_dw, use_cfar=False, num_doppler_bins_override=n_bins_acq)
        sampled = []
        for s_, pid in enumerate(prn_ids):
            code = gps_ca_code(pid % 32 + 1)
            idx = np.minimum((np.arange(N_EPOCH) * (1.023e6 / FS)).astype(np.int64), 1022)
            sampled.append(code[idx].astype(np.complex64))
            acq.set_local_code(s_, sampled[-1])
        # the searched block: two code periods of an RF stream of this rank's own, carrying ONE satellite -- the first PRN of the rank's
        # set, at 47 dB-Hz so that 2 x 1 ms finds it among n_sat x 41 x 25000 cells whatever the rank (the tracking streams hold their
        # satellites at 38-48 dB-Hz; the search time does not depend on the samples)
        ch_x = 0
        x, t_acq = make_channel_stream(torch, dev, gps_ca_code(prn_ids[0] % 32 + 1), 2 * N_EPOCH + 64, seed=5003 + 11 * rank, cn0_db_hz=47.0)
        torch.cuda.synchronize()
        def acq_search():
            acq.reset()
            acq.dwell_enqueue(x.data_ptr(), stream)
            acq.dwell_enqueue(x.data_ptr() + 8 * N_EPOCH, stream)
            acq.flush(stream)  # this search's own statistics kernel, enqueued behind its passes (no host synchronisation)
        def timed(n_warm, reps):
            for _ in range(n_warm):
                acq_search()
            torch.cuda.synchronize()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()  # one pair of events around all the searches (a pair per search idles the GPU ~12 us between searches)
            for _ in range(reps):
                acq_search()
            a1.record()
            torch.cuda.synchronize()
            return float(a0.elapsed_time(a1)) / reps
        reps = args.acq_reps
        acq_ms = timed(max(1, warmups), reps)
        # the same searches again behind ~6 ms of the same load (20 searches): the steady-clock figure, reported beside the headline one
        steady_ms = timed(20, reps) if args.segments > 0 else None
        # sanity: the searched stream carries ONE satellite -- its slot must win, at its code phase and in its Doppler bin
        ares = acq.fetch_results(stream)
        stats = np.array([r.test_statistics for r in ares])
        t_a = t_acq
        want_delay = ((CODE_LEN - t_a["tau0"]) % CODE_LEN) * FS / 1.023e6
        d_err = abs(ares[ch_x].indext - want_delay)
        assert int(np.argmax(stats)) == ch_x and min(d_err, N_EPOCH - d_err) <= 26 and abs(ares[ch_x].doppler_hz - t_a["doppler"]) <= 250, \
            ("acquisition lost its PRN", ch_x, stats[:4], ares[ch_x].indext, want_delay, ares[ch_x].doppler_hz, t_a["doppler"])
        # Algorithmic bytes per (PRN, bin, dwell) cell: read X 8N + read conj-FFT(code) 8N + write |.|^2 4N,
        # + 4N read when a later dwell accumulates: 20N for dwell 1, 24N for dwell 2
        cells = n_sat * n_bins_acq
        acq_alg = float(cells * N_EPOCH * (20 + 24))
        acq_gbps = acq_alg / (acq_ms * 1e-3) / 1e9
        aroof = {"bound": "hbm", "achieved": acq_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": acq_gbps / HBM_PEAK_GBPS,
            "algorithmic_bytes_per_search": acq_alg, "search_ms": acq_ms,
            "steady_state_search_ms": steady_ms, "steady_state_frac": (acq_alg / (steady_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if steady_ms else None,
            "note": "whole search (2 dwells: shared forward transforms, inverse row + column passes of every cell, its own statistics "
                    "kernel via gc_acq_flush) timed with HIP events on the launch stream; achieved = algorithmic bytes / that time; "
                    "steady_state_* = the same behind 20 untimed searches", "traffic": None}
        apath = os.path.join(ROOT, "profiles", "acq_latest.json")
        if os.path.exists(apath) and n_sat == 32:
            try:
                aj = json.load(open(apath))
                aroof["traffic"] = aj.get("hbm_bytes_per_search")
                aroof["traffic_source"] = "profiles/acq_latest.json (round %s rocprofv3 PMC passes of this workload; not measured in this run)" % aj.get("round")
                aroof["dominant_kernel"] = aj.get("dominant_kernel")
                aroof["dominant_kernel_us_per_launch"] = aj.get("dominant_kernel_us")
            except Exception:
                pass
        res = {"dwells_per_s": n_sat * n_dw / (acq_ms * 1e-3), "ms_per_search": acq_ms,
            "timed_searches": reps, "warmup_searches": max(1, warmups), "prns": [pid % 32 + 1 for pid in prn_ids],
            "workload": "GPS L1 C/A PCPS, 25 Msps, N=25000, %d PRNs x 41 Doppler bins x 2 dwells" % n_sat, "roofline": aroof}
        if not args.no_cpu and world == 1:
            # CPU baseline of the same search: the oracle's acquisition_core restatement (own float64 mixed-radix FFT, gcc -O3
            # -march=native, 1 thread) on a bounded sample: PRN-dwells of the same block sizes and Doppler grid
            from oracle import Oracle, host_cpu_model
            orc = Oracle(native=True)
            xh = x[:2 * N_EPOCH].cpu().numpy().view(np.complex64).reshape(-1)
            pc = orc.pcps(fs_in=FS, sampled_ms=1, ms_per_code=1, samples_per_ms=np.float32(FS) * np.float32(0.001), samples_per_code=25000.0,
                samples_per_chip=25, doppler_max=5125, doppler_step=250, max_dwells=n_dw)  # ceil(10250 / 250) = 41 bins
            pc.set_local_code(sampled[ch_x])
            n_done, t0c = 0, time.perf_counter()
            while time.perf_counter() - t0c < args.cpu_seconds * 0.5:
                pc.reset_grid()
                for d_ in range(n_dw):
                    pc.core(xh[d_ * N_EPOCH:])
                    n_done += 1
            dtc = time.perf_counter() - t0c
            res["cpu_baseline"] = {"value": n_done / dtc, "unit": "dwells/s", "cores": 1, "kind": "port",
                "sample": "%d PRN-dwells (41 Doppler bins x N=25000 each, 2-dwell searches of PRN 1) in %.1f s, oracle PCPS: float64 "
                          "mixed-radix FFT port, NOT the reference's FFTW3f float32 path (pcps_acquisition.cc:721-727), so far slower than "
                          "the real block; gcc -O3 -march=native built on this host (%s), 1 thread" % (n_done, dtc, host_cpu_model())}
        acq.close()
        return res

    # every rank's share of BASELINE configs[4] (the hybrid mix) and of configs[3] (its PRNs of the 32), timed behind the headline
    # region with no barrier in between: nothing waits on another rank's extras; the figures travel in per_gpu[]
    share = None
    if world > 1 or not (args.no_shared and args.no_acq):
        share = {"hybrid": None, "acquisition": None}
        if world > 1 or not args.no_shared:
            share["hybrid"] = run_hybrid()
        if not args.no_acq:
            share["acquisition"] = run_acquisition(sharding.shard_channels(32, world, rank), args.acq_warmup)
        if share["hybrid"] is not None:
            mine["hybrid_msamples_s"] = share["hybrid"]["value"]
            mine["hybrid_frac"] = share["hybrid"]["roofline"]["frac"]
            mine["hybrid_hbm_frac"] = share["hybrid"]["roofline"]["hbm_frac"]
        if share["acquisition"] is not None:
            mine["acq_dwells_per_s"] = share["acquisition"]["dwells_per_s"]
            mine["acq_ms_per_search"] = share["acquisition"]["ms_per_search"]
            mine["acq_hbm_frac"] = share["acquisition"]["roofline"]["frac"]
            mine["acq_prns"] = share["acquisition"]["prns"]
    per_gpu = sharding.gather_per_rank(mine, dist, world)

    # sanity: the prompt correlators see their signals (guards against timing a broken path)
    out = d_out.cpu().numpy().reshape(N_CHANNELS, E, N_TAPS, 2)
    pmag = np.hypot(out[:, :, 1, 0], out[:, :, 1, 1]).mean(axis=1)
    expect = np.array([t["amp"] * N_EPOCH for t in truths])
    assert np.all(pmag > 0.5 * expect), "prompt correlators lost the signal"

    result = None
    if rank == 0:
        extra = {}
        # the extras below (other than every rank's hybrid / acquisition share) are single-GPU side measurements: in a multi-GPU run
        # the other ranks would only wait for them
        if world > 1:
            args.no_shared = True
        # ---- shared-stream mode: all 32 channels on ONE RF stream (cache-served) ----
        if not args.no_shared:
            for ch in range(N_CHANNELS):
                batch.set_input_dev(ch, streams[0].data_ptr(), n_stream)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(args.steps):
                step()
            s1.record()
            torch.cuda.synchronize()
            shared_ms = s0.elapsed_time(s1) / args.steps
            extra["shared_stream"] = {"value": samples_per_step / (shared_ms * 1e-3) / 1e6, "unit": "Msamples/s",
                "ms_per_step": shared_ms, "note": "32 channels read one RF stream; input served from L2/Infinity Cache",
                # one stream of E x 25000 samples comes from HBM once; the other 31 reads are cache hits
                "roofline": mode_roofline(vceil, [(samples_per_step, 8.0 / N_CHANNELS, "gps_l1_3tap_f32", 1.0)], shared_ms)}
        extra["realtime_factor_256ch"] = value / world / (256 * FS / 1e6)

        # ---- int16 IQ in HBM (cshort front-end samples, converted on load): 4 B per channel-sample ----
        if not args.no_shared:
            b16 = gnsscorr.TrackingBatch(ctx, N_CHANNELS, N_TAPS, CODE_LEN)
            b16.set_input_format(gnsscorr.GC_IQ_I16)
            q_streams = []
            for ch in range(N_CHANNELS):
                qs = torch.clamp(torch.round(streams[ch] * 600.0), -32768, 32767).to(torch.int16).contiguous()
                q_streams.append(qs)
                b16.set_code(ch, codes[ch], shifts)
                b16.set_input_dev(ch, qs.data_ptr(), n_stream)
            b16.set_nominal_length(N_EPOCH)
            for _ in range(2):
                b16.run_dev(E, d_params.data_ptr(), d_out.data_ptr(), stream)
            torch.cuda.synchronize()
            i0, i1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            i0.record()
            for _ in range(args.steps):
                b16.run_dev(E, d_params.data_ptr(), d_out.data_ptr(), stream)
            i1.record()
            torch.cuda.synchronize()
            i16_ms = i0.elapsed_time(i1) / args.steps
            extra["int16_input"] = {"value": samples_per_step / (i16_ms * 1e-3) / 1e6, "unit": "Msamples/s", "ms_per_step": i16_ms,
                "hbm_gbps": 4.0 * samples_per_step / (i16_ms * 1e-3) / 1e9,
                "note": "same workload with lv_16sc_t IQ in HBM (distinct buffer per channel), 4 B per channel-sample",
                "roofline": mode_roofline(vceil, [(samples_per_step, 4.0, "gps_l1_3tap_i16", 1.0)], i16_ms)}
            b16.close()
            del q_streams

        # ---- BASELINE configs[2] shape: Galileo E1, 5 taps (VE/E/P/L/VL), L = 8184 (2 samples/chip), 4 ms epochs ----
        if not args.no_shared:
            n_gal, e_gal = 100000, max(1, E // 4)
            bg = gnsscorr.TrackingBatch(ctx, N_CHANNELS, 5, 8184)
            rng_g = np.random.Generator(np.random.PCG64(1003))
            gshifts = np.array([-1.2, -0.3, 0.0, 0.3, 1.2], np.float32)
            grecs = []
            for ch in range(N_CHANNELS):
                prim = np.sign(rng_g.standard_normal(4092)).astype(np.float32)  # memory-code stand-in: timing does not depend on the chips
                boc = np.empty(8184, np.float32)
                boc[0::2], boc[1::2] = prim, -prim
                bg.set_code(ch, boc, gshifts)
                bg.set_input_dev(ch, streams[ch].data_ptr(), n_stream)
                grecs.append([gnsscorr.epoch_params(k * n_gal, 0.1, float(np.float32(2 * np.pi * 1000.0 / FS)), 0.3, float(np.float32(2.046e6 / FS)), n_gal)
                    for k in range(e_gal)])
            bg.set_nominal_length(n_gal)  # the engine sizes a launch's code window by it (two half-period slices for L = 8184)
            if os.environ.get("BENCH_SLICES"):
                bg.set_slices(int(os.environ["BENCH_SLICES"]))
            d_gp = torch.from_numpy(gnsscorr.epoch_params_array(grecs).view(np.uint8)).to(dev)
            d_go = torch.zeros(N_CHANNELS * e_gal * 5, 2, device=dev, dtype=torch.float32)
            for _ in range(2):
                bg.run_dev(e_gal, d_gp.data_ptr(), d_go.data_ptr(), stream)
            torch.cuda.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            for _ in range(args.steps):
                bg.run_dev(e_gal, d_gp.data_ptr(), d_go.data_ptr(), stream)
            g1.record()
            torch.cuda.synchronize()
            gal_ms = g0.elapsed_time(g1) / args.steps
            gal_samples = N_CHANNELS * e_gal * n_gal
            extra["galileo_e1_5tap"] = {"value": gal_samples / (gal_ms * 1e-3) / 1e6, "unit": "Msamples/s", "ms_per_step": gal_ms,
                "hbm_gbps": 8.0 * gal_samples / (gal_ms * 1e-3) / 1e9,
                "note": "32 channels, 25 Msps, 5 taps, L = 8184, N = 100000 (4 ms), distinct IQ buffer per channel",
                "roofline": mode_roofline(vceil, [(gal_samples, 8.0, "galileo_5tap_f32", 1.0)], gal_ms)}
            bg.close()

        if share is not None and share["hybrid"] is not None:
            extra["hybrid_gps_galileo_beidou"] = share["hybrid"]
            if world > 1:
                # BASELINE configs[4]: the 256-channel hybrid over all GPUs = every rank's share side by side (no collective)
                hv = [g["hybrid_msamples_s"] for g in per_gpu]
                extra["hybrid_all_gpus"] = {"value": float(sum(hv)), "unit": "Msamples/s", "channels": 32 * world,
                    "realtime_factor": float(sum(hv)) * 1e6 / (32 * world * FS), "slowest_gpu_msamples_s": float(min(hv)),
                    "note": "sum of the per-GPU hybrid shares (per_gpu[].hybrid_msamples_s), each timed on its own GPU behind the headline region"}

        # ---- host-fed pipeline: one RF stream pushed over PCIe into the HBM ring while the channels track it ----
        if not args.no_shared:
            blk_epochs = 16
            blk = blk_epochs * N_EPOCH  # 16 ms of the stream per push (3.2 MB)
            n_warm = 8  # every block of the pinned source has been DMA-mapped once, the ring has wrapped
            n_push = args.steps + n_warm
            ring = gnsscorr.IqStream(ctx, capacity_samples=4 * blk, max_window_samples=N_EPOCH)
            bs = gnsscorr.TrackingBatch(ctx, N_CHANNELS, 3, CODE_LEN)
            for ch in range(N_CHANNELS):
                bs.set_code(ch, codes[ch], shifts)
                bs.set_input_stream(ch, ring)
            pinned = torch.empty(4 * blk, 2, dtype=torch.float32).pin_memory()
            pinned.copy_(streams[0][:4 * blk].cpu())
            srecs = []
            for k in range(n_push):
                # channel-major records of push k (absolute sample numbers); timing does not depend on the scalars
                srecs.append(gnsscorr.epoch_params_array([[gnsscorr.epoch_params(k * blk + e * N_EPOCH, 0.1, 1e-3, 0.3, float(np.float32(CODE_LEN / N_EPOCH)), N_EPOCH)
                    for e in range(blk_epochs)] for _ in range(N_CHANNELS)]))
            d_sp = torch.from_numpy(np.concatenate(srecs).view(np.uint8)).to(dev)
            d_so = torch.zeros(N_CHANNELS * blk_epochs * 3, 2, device=dev, dtype=torch.float32)
            rec_bytes = N_CHANNELS * blk_epochs * 48

            def feed(k):
                ring.push_pinned(pinned.data_ptr() + (k % 4) * blk * 8, blk)
                bs.set_read_floor(k * blk)
                bs.run_dev(blk_epochs, d_sp.data_ptr() + k * rec_bytes, d_so.data_ptr(), stream)

            for k in range(n_warm):
                feed(k)
            torch.cuda.synchronize()
            ring.synchronize()
            t0 = time.perf_counter()
            for k in range(n_warm, n_push):
                feed(k)
            tstream.synchronize()
            ring.synchronize()
            dt = time.perf_counter() - t0
            fed = (n_push - n_warm) * blk
            extra["host_fed_pipeline"] = {"value": N_CHANNELS * fed / dt / 1e6, "unit": "Msamples/s", "ms_per_push": dt / (n_push - n_warm) * 1e3,
                "h2d_gbps": 8.0 * fed / dt / 1e9, "stream_realtime_factor": fed / dt / FS,
                "note": "PCIe-inclusive: 16 ms blocks of ONE 25 Msps stream pushed from pinned host memory into the HBM ring "
                        "(own copy stream) while 32 channels correlate the previous block; never the headline value"}
            bs.close()
            ring.close()

        # ---- closed loop on the device: 256 channels x 25 Msps, DLL/PLL maths in the kernel, no host round trip ----
        if not args.no_shared:
            n_cl, e_cl = 256, min(E, 64)
            loop = gnsscorr.TrackingLoop(ctx, n_cl, CODE_LEN)
            t0_ = truths[0]
            delay = ((CODE_LEN - t0_["tau0"]) % CODE_LEN) * FS / 1.023e6
            lc = gnsscorr.LoopConf()
            for k_, v_ in dict(fs_in=float(FS), signal_carrier_freq_hz=1575.42e6, code_chip_rate_hz=1.023e6, code_period_s=0.001, carrier_lock_th=0.85,
                    code_length_chips=CODE_LEN, code_samples_per_chip=1, vector_length=N_EPOCH, pull_in_time_s=2, veml=0, pll_filter_order=3,
                    dll_filter_order=2, cn0_samples=20, cn0_min=25, max_lock_fail=50, pll_bw_hz=40.0, dll_bw_hz=2.0, fll_bw_hz=35.0,
                    early_late_space_chips=0.5, acq_delay_samples=float(np.round(delay)), acq_doppler_hz=float(np.round(t0_["doppler"] / 10) * 10)).items():
                setattr(lc, k_, v_)
            d_recs = torch.zeros(n_cl * e_cl * gnsscorr.LOOP_RECORD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            def cl_run():
                for ch in range(n_cl):
                    loop.set_input_dev(ch, streams[0].data_ptr(), n_stream)
                    loop.start(ch, lc, codes[0])
                torch.cuda.synchronize()
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                loop.run_dev(e_cl, d_recs.data_ptr(), stream)
                c1.record()
                torch.cuda.synchronize()
                return c0.elapsed_time(c1)
            cl_run()
            cl_ms = min(cl_run() for _ in range(3))
            recs = np.frombuffer(d_recs.cpu().numpy().tobytes(), gnsscorr.LOOP_RECORD_DTYPE).reshape(n_cl, e_cl)
            assert np.all(recs["valid"][:, :e_cl - 2] == 1)
            extra["closed_loop"] = {"channels": n_cl, "epochs": e_cl, "ms": cl_ms, "realtime_factor": e_cl * 1.0 / cl_ms,
                "value": n_cl * e_cl * N_EPOCH / (cl_ms * 1e-3) / 1e6, "unit": "Msamples/s",
                "note": "256 channels tracked in closed loop (DLL/PLL maths on the device), one workgroup per channel, shared RF stream",
                # per code period: the correlation at the VALU ceiling of the CUs in use (one per channel) + 2.6 us of loop maths, records
                # and barriers on one lane (measured with the correlation stubbed out), for e_cl periods in a row
                "roofline": mode_roofline(vceil, [(n_cl * e_cl * N_EPOCH, 8.0 / n_cl, "closed_loop_3tap_512", min(1.0, n_cl / 256.0))], cl_ms,
                    serial_us_per_unit=2.6 * e_cl)}
            loop.close()

            # Galileo E1 closed loop, 5 taps, 4 ms periods, data component alone vs pilot tracking (E1-C drives the loop,
            # the E1-B prompt is one extra LDS lookup on the prompt tap's chip index)
            n_g, e_g, n_len = 128, max(2, min(E, 64) // 4), 4 * N_EPOCH
            rng_g = np.random.Generator(np.random.PCG64(77))
            e1c = np.sign(rng_g.standard_normal(8184)).astype(np.float32)
            e1b = np.sign(rng_g.standard_normal(8184)).astype(np.float32)
            gres = {}
            for pilot in (False, True):
                gl = gnsscorr.TrackingLoop(ctx, n_g, 8184)
                glc = gnsscorr.LoopConf()
                for k_, v_ in dict(fs_in=float(FS), signal_carrier_freq_hz=1575.42e6, code_chip_rate_hz=1.023e6, code_period_s=0.004, carrier_lock_th=0.85,
                        code_length_chips=4092, code_samples_per_chip=2, vector_length=n_len, pull_in_time_s=2, veml=1, pll_filter_order=3,
                        dll_filter_order=2, cn0_samples=20, cn0_min=25, max_lock_fail=50, pll_bw_hz=15.0, dll_bw_hz=0.75, fll_bw_hz=10.0,
                        early_late_space_chips=0.15, very_early_late_space_chips=0.6, acq_delay_samples=0.0, acq_doppler_hz=1000.0).items():
                    setattr(glc, k_, v_)
                d_grecs = torch.zeros(n_g * e_g * gnsscorr.LOOP_RECORD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
                def gl_run():
                    for ch in range(n_g):
                        gl.set_input_dev(ch, streams[0].data_ptr(), n_stream)
                        gl.set_sync(ch, gnsscorr.LoopSyncConf.make(extend_correlation_symbols=1, track_pilot=pilot, symbols_per_bit=1), e1b if pilot else None)
                        gl.start(ch, glc, e1c)
                    torch.cuda.synchronize()
                    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    c0.record()
                    gl.run_dev(e_g, d_grecs.data_ptr(), stream)
                    c1.record()
                    torch.cuda.synchronize()
                    return c0.elapsed_time(c1)
                gl_run()
                gres[pilot] = min(gl_run() for _ in range(3))
                gl.close()
            extra["closed_loop_galileo_e1"] = {"channels": n_g, "periods": e_g, "ms_data_only": gres[False], "ms_pilot": gres[True],
                "realtime_factor_pilot": e_g * 4.0 / gres[True], "value": n_g * e_g * n_len / (gres[True] * 1e-3) / 1e6, "unit": "Msamples/s",
                "note": "128 channels x 5 taps x 100000-sample periods in closed loop on a shared stream; pilot = 5 taps + the data component's prompt",
                "roofline": mode_roofline(vceil, [(n_g * e_g * n_len, 8.0 / n_g, "closed_loop_5tap_1024_pilot", min(1.0, n_g / 256.0))], gres[True],
                    serial_us_per_unit=2.6 * e_g)}

        # ---- closed loop, one GPU's share of BASELINE configs[4]: 16 GPS L1 C/A + 8 Galileo E1 (5 taps, 4 ms) + 8 BeiDou B1I in three
        # engines on three streams ----
        if not args.no_shared:
            ms_total = min(E, 64)
            rng_c = np.random.Generator(np.random.PCG64(1005))
            side = [torch.cuda.Stream(device=dev) for _ in range(3)]
            def make_engines(slices):
                out = []
                specs = [(16, CODE_LEN, 1, 0, 1.023e6, N_EPOCH, 0.001, 0.5, 0.0, 1), (8, 8184, 2, 1, 1.023e6, 4 * N_EPOCH, 0.004, 0.15, 0.6, 4),
                    (8, 2046, 1, 0, 2.046e6, N_EPOCH, 0.001, 0.5, 0.0, 1)]
                for n_c, L, spc, veml, chip_rate, vlen, period, el, vel, ms_per in specs:
                    eng = gnsscorr.TrackingLoop(ctx, n_c, L)
                    eng.set_geometry(slices_per_channel=slices)
                    cfgc = gnsscorr.LoopConf()
                    for k_, v_ in dict(fs_in=float(FS), signal_carrier_freq_hz=1575.42e6, code_chip_rate_hz=chip_rate, code_period_s=period, carrier_lock_th=0.85,
                            code_length_chips=L // spc, code_samples_per_chip=spc, vector_length=vlen, pull_in_time_s=2, veml=veml, pll_filter_order=3,
                            dll_filter_order=2, cn0_samples=20, cn0_min=25, max_lock_fail=50, pll_bw_hz=40.0, dll_bw_hz=2.0, fll_bw_hz=35.0,
                            early_late_space_chips=el, very_early_late_space_chips=vel, acq_delay_samples=0.0, acq_doppler_hz=1000.0).items():
                        setattr(cfgc, k_, v_)
                    cd = codes[0] if L == CODE_LEN else np.sign(rng_c.standard_normal(L)).astype(np.float32)
                    n_per = ms_total // ms_per
                    recs_d = torch.zeros(n_c * n_per * gnsscorr.LOOP_RECORD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
                    out.append((eng, cfgc, cd, n_c, n_per, recs_d))
                return out
            def share_run(engs):
                for k_, (eng, cfgc, cd, n_c, n_per, recs_d) in enumerate(engs):
                    for ch in range(n_c):
                        eng.set_input_dev(ch, streams[(k_ * 8 + ch) % N_CHANNELS].data_ptr(), n_stream)
                        eng.start(ch, cfgc, cd)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for k_, (eng, cfgc, cd, n_c, n_per, recs_d) in enumerate(engs):
                    eng.run_dev(n_per, recs_d.data_ptr(), side[k_].cuda_stream)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0_) * 1e3
            engs = make_engines(0)
            share_run(engs)
            share_ms = min(share_run(engs) for _ in range(3))
            for g_ in engs:
                g_[0].close()
            # floor of one millisecond of signal: the three engines run side by side, each needs its periods one after the other
            def period_floor_us(key, n_samp):
                k_ = vceil.get(key)
                return (n_samp / (k_["ceiling_msamples_s"] * 1e6 / 256.0) * 1e6 + 2.6) if k_ else None
            fl = [period_floor_us("closed_loop_3tap_1024", N_EPOCH), period_floor_us("closed_loop_5tap_1024", 4 * N_EPOCH), period_floor_us("closed_loop_3tap_1024", N_EPOCH)]
            share_roof = None
            if all(f is not None for f in fl):
                floor_us_per_ms = max(fl[0], fl[1] / 4.0, fl[2])
                share_roof = {"bound": "valu + one-lane loop maths", "achieved": ms_total / share_ms, "ceiling": 1000.0 / floor_us_per_ms, "unit": "x real time",
                    "frac": (1000.0 / floor_us_per_ms and (ms_total / share_ms) / (1000.0 / floor_us_per_ms)),
                    "ceiling_source": "per code period: samples / (VALU ceiling of one CU, profiles/r04_valu_ceilings.json) + 2.6 us of one-lane loop maths; "
                                      "the slowest of the three concurrent engines sets the floor"}
            extra["closed_loop_cfg5_share"] = {"channels": 32, "ms_of_signal": ms_total, "ms": share_ms, "realtime_factor": ms_total / share_ms, "roofline": share_roof,
                "note": "16 GPS L1 C/A + 8 Galileo E1 (5 taps, 4 ms) + 8 BeiDou B1I channels x 25 Msps in closed loop, three engines on three "
                        "streams, one 1024-thread workgroup per channel, host wall time of the three run_dev calls (launch-inclusive).  DEPENDS ON "
                        "GPU_MAX_HW_QUEUES >= 6 in the process environment before HIP initialises (this script sets 8, see runtime_env): with the "
                        "runtime's default of 4 hardware queues two engines share one and run one after the other (47x).  A code "
                        "period costs ~11 us whatever the channel count (7.5 correlation by one CU + 2.6 one-lane loop maths): cutting periods "
                        "into slices over more CUs was built and measured slower (13.4 us: experiments build, DESIGN.md appendix A)"}

        if share is not None and share.get("acquisition") is not None:
            extra["acquisition"] = share["acquisition"]
            if world > 1:
                # BASELINE configs[3] over all GPUs: the 32 PRNs partitioned PRN i -> GPU (i - 1) mod G
                all_prns = sorted(p_ for g in per_gpu for p_ in g["acq_prns"])
                extra["acquisition_all_gpus"] = {"dwells_per_s": float(sum(g["acq_dwells_per_s"] for g in per_gpu)),
                    "search_ms_slowest_gpu": float(max(g["acq_ms_per_search"] for g in per_gpu)), "prns_partition_1_to_32": all_prns == list(range(1, 33)),
                    "note": "every GPU searches its PRNs (per_gpu[].acq_prns) x 41 bins x 2 dwells on its own RF stream copy; the whole 32-PRN "
                            "search takes the slowest GPU's time"}

        cpu = None
        if not args.no_cpu and world == 1:
            sig0 = streams[0].cpu().numpy().view(np.complex64).reshape(-1)
            t = truths[0]
            step_c = t["code_rate"] / FS
            recs = []
            for k in range(min(E, 64)):
                start = k * N_EPOCH
                cp = (t["tau0"] + start * step_c) % CODE_LEN
                rem = -cp
                if rem < -CODE_LEN / 2:
                    rem += CODE_LEN
                carr = (t["phi"] + 2 * np.pi * t["doppler"] * start / FS) % (2 * np.pi)
                recs.append((start, np.float32(carr), np.float32(2 * np.pi * t["doppler"] / FS), np.float32(rem), np.float32(step_c)))
            v, n_done, dt = cpu_baseline(codes[0], shifts, sig0, recs, args.cpu_seconds)
            from oracle import host_cpu_model
            cpu = {"value": v, "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": "%d channel-epochs of the same workload (GPS L1 C/A, N=25000, 3 taps) in %.1f s, oracle "
                          "(volk_gnsssdr generic restatement, gcc -O3 -march=native built on this host: %s), 1 thread" % (n_done, dt, host_cpu_model())}
            # the same port with one thread per channel on every host core this process may use
            n_thr = max(1, min(len(os.sched_getaffinity(0)), N_CHANNELS))
            if n_thr > 1:
                vt, nt, dtt = cpu_baseline_threads(codes[0], shifts, sig0, recs, min(6.0, args.cpu_seconds), n_thr, v)
                extra["cpu_baseline_all_cores"] = {"value": vt, "unit": "Msamples/s", "cores": n_thr, "kind": "port",
                    "sample": "%d channel-epochs in %.1f s, one thread per channel" % (nt, dtt)}

        # HBM traffic of the tracking kernel: rocprofv3 PMC counters cannot be read from inside the run, so the figure comes from
        # the committed counter passes of the SAME workload (profiles/collect.sh) and is labelled as such; null for another workload
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and E == 256:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "profiles/traffic_latest.json (round %s rocprofv3 PMC passes: FETCH_SIZE x 2 + WRITE_SIZE on this workload; not measured in this run)" % tj.get("round")
            except Exception:
                traffic = None
        result = {
            "metric": "Msamples/s tracked (N-channel multicorrelator, whole job)",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "GPS L1 C/A, 32 channels/GPU, 25 Msps, 3-tap E/P/L multicorrelator, "
                                   "%d code periods (25000 samples) per channel per step, distinct IQ buffer per channel" % E,
                "channels_per_gpu": N_CHANNELS, "epochs_per_step": E, "samples_per_epoch": N_EPOCH,
                "parallelism": "channels sharded over %d GPU(s), no collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                # <NTAPS, HDR, HDC, FMT = GC_IQ_F32, CC, SC16>: the name rocprofv3 prints
                "kernel": TRK_KERNEL_NAME, "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                # diagnostic pass behind the timed region (see above): per-launch ms of consecutive 5-launch segments of a COLD start
                # (no pre-roll) with the same W-launch lead-in, and where they settle
                "segments_ms": segments_ms, "segment_launches": SEG_LEN,
                "steady_state_kernel_ms": steady_ms,
                "steady_state_frac": (alg_bytes / (steady_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if steady_ms else None,
                "cold_start_kernel_ms": cold_ms,
                "cold_start_frac": (alg_bytes / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if cold_ms else None,
                "segments_note": "untimed pass AFTER the timed region: GPU idle for %.2f s -> %d warm-up launches -> synchronise -> %d x %d "
                                 "launches back to back, one HIP event per segment.  cold_start_* = mean of its first %d launches = what "
                                 "the timed region measures with --preroll-ms 0; steady_state_* = median of its second half.  frac / "
                                 "kernel_ms above are the timed region's (behind the named pre-roll)"
                                 % (COLD_IDLE_S, args.warmup, args.segments, SEG_LEN, args.steps)},
            "preroll": {"launches": n_preroll, "approx_ms": args.preroll_ms,
                "note": "untimed launches of the same step in front of the W warm-up steps, so that the timed region lies behind the chip's "
                        "start-of-load power transient (~8 ms); the timed region is unchanged: K full steps between barrier + synchronise. "
                        "--preroll-ms 0 removes it; roofline.cold_start_frac is the figure without it"},
            "cpu_baseline": cpu,
            "runtime_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
            "per_gpu": [{k: g.get(k) for k in ("rank", "msamples_s", "hbm_frac", "kernel_ms", "steady_state_kernel_ms", "steady_state_hbm_frac", "cold_start_kernel_ms",
                "hybrid_msamples_s", "hybrid_frac", "hybrid_hbm_frac", "acq_dwells_per_s", "acq_ms_per_search", "acq_hbm_frac", "acq_prns")} for g in per_gpu],
            "slowest_rank": max(per_gpu, key=lambda g: g["elapsed_s"])["rank"],
        }
        result.update(extra)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
