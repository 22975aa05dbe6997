"""Turns the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof_*) into the small files committed under
profiles/: per-kernel duration tables and traffic_latest.json (HBM bytes per launch of the tracking kernel, with the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md's HBM section: the counter reports 64 B per 128-B request of a
16-B/lane streaming load, so it is doubled; WRITE_SIZE is exact).  The acquisition passes give acq_latest.json: HBM bytes per cfg4 search over every acq_* kernel, per-kernel durations and the SQ
counters of the dominant kernels.  Usage: python profiles/summarise.py [round_tag] [full trk acq sq]"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
PASSES = sys.argv[2:] or ["line", "full", "trk", "trksq", "acq", "sq"]
KERNEL = "trk_multicorrelator_kernel<3, false, false, 0, false, false, false>"


def counter_mean(path, counter):
    vals = []
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter and r["Kernel_Name"].startswith("void " + KERNEL):
            vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


def find(d, suffix):
    """rocprofv3 nests its output under <dir>/<host>/...: first file below d that ends with suffix"""
    for root, _, files in os.walk(os.path.join(OUT, d)):
        for f in sorted(files):
            if f.endswith(suffix):
                return os.path.join(root, f)
    raise FileNotFoundError((d, suffix))


def short(name):
    name = name[5:] if name.startswith("void ") else name
    return name.split("(")[0]


def acq_summary():
    """acq_* kernels of the cfg4 search: durations (stats pass), HBM bytes (FETCH x 2 + WRITE) and SQ counters, per search."""
    stats_path = find("prof_acq", "kernel_stats.csv")
    shutil.copy(stats_path, os.path.join(PROF, TAG + "_acq_kernel_stats.csv"))
    searches_stats = 1 + 9   # collect.sh: 1 warm-up + --acq-reps 9
    searches_pmc = 1 + 4
    kern = {}
    for r in csv.DictReader(open(stats_path)):
        n = short(r["Name"])
        if n.startswith("acq_"):
            kern[n] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
    def counter_sums(d, counters):
        sums = {}
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            n = short(r["Kernel_Name"])
            if n.startswith("acq_") and r["Counter_Name"] in counters:
                sums.setdefault(n, {}).setdefault(r["Counter_Name"], [0.0, 0])
                sums[n][r["Counter_Name"]][0] += float(r["Counter_Value"])
                sums[n][r["Counter_Name"]][1] += 1
        return sums
    fetch = counter_sums("prof_acq_fetch", {"FETCH_SIZE"})
    write = counter_sums("prof_acq_write", {"WRITE_SIZE"})
    total_fetch_kb = sum(v["FETCH_SIZE"][0] for v in fetch.values())
    total_write_kb = sum(v["WRITE_SIZE"][0] for v in write.values())
    per_kernel = {}
    for n in sorted(kern):
        f = fetch.get(n, {}).get("FETCH_SIZE", [0.0, 1])
        w = write.get(n, {}).get("WRITE_SIZE", [0.0, 1])
        per_kernel[n] = dict(kern[n], ms_per_search=kern[n]["total_ms"] / searches_stats,
            fetch_kb_per_launch=f[0] / max(1, f[1]), write_kb_per_launch=w[0] / max(1, w[1]),
            hbm_mb_per_search=(2.0 * f[0] + w[0]) * 1024.0 / searches_pmc / 1e6)
    dom = max(kern, key=lambda n: kern[n]["total_ms"])
    alg = 32 * 41 * 25000 * (20 + 24)
    out = {"round": int(TAG[1:3]), "workload": "bench.py acquisition: GPS L1 C/A PCPS, 25 Msps, N = 25000, 32 PRNs x 41 bins x 2 dwells",
        "searches_in_stats_pass": searches_stats, "searches_in_pmc_passes": searches_pmc,
        "correction": "FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md HBM section), WRITE_SIZE exact; the 32 set_local_code transforms of the set-up "
                      "run the same kernels on one cell each and are included (< 1 %)",
        "hbm_bytes_per_search": (2.0 * total_fetch_kb + total_write_kb) * 1024.0 / searches_pmc,
        "algorithmic_bytes_per_search": alg, "dominant_kernel": dom, "dominant_kernel_us": kern[dom]["avg_us"],
        "kernel_ms_per_search": sum(k["total_ms"] for k in kern.values()) / searches_stats, "kernels": per_kernel}
    out["traffic_over_algorithmic"] = out["hbm_bytes_per_search"] / alg
    if "sq" in PASSES:
        sq = {}
        for d in ("prof_acq_sq1", "prof_acq_sq2"):
            try:
                path = find(d, "counter_collection.csv")
            except FileNotFoundError:
                continue
            for r in csv.DictReader(open(path)):
                n = short(r["Kernel_Name"])
                if n.startswith("acq_"):
                    e = sq.setdefault(n, {}).setdefault(r["Counter_Name"], [0.0, 0])
                    e[0] += float(r["Counter_Value"])
                    e[1] += 1
        out["sq_counters_mean_per_launch"] = {n: {c: v[0] / v[1] for c, v in sorted(cs.items())} for n, cs in sorted(sq.items())}
    json.dump(out, open(os.path.join(PROF, "acq_latest.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(PROF, TAG + "_acq_summary.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ("kernels", "sq_counters_mean_per_launch")}, indent=1))
    for n, k in per_kernel.items():
        print("%-60s %5d calls %8.1f us  %7.3f ms/search  %8.1f MB/search" % (n[:60], k["calls"], k["avg_us"], k["ms_per_search"], k["hbm_mb_per_search"]))


def main():
    if "acq" in PASSES:
        acq_summary()
    if "line" in PASSES:
        for line in open(os.path.join(OUT, "bench_line.json")):
            if line.startswith("{"):
                open(os.path.join(PROF, TAG + "_bench_line.json"), "w").write(line)
    if "full" in PASSES:
        shutil.copy(find("prof_full", "kernel_stats.csv"), os.path.join(PROF, TAG + "_full_bench_kernel_stats.csv"))
        for line in open(os.path.join(OUT, "prof_full.log")):
            if line.startswith("{"):
                open(os.path.join(PROF, TAG + "_bench_line_under_rocprof.json"), "w").write(line)
    if "trk" in PASSES:
        trk_summary()
    if "trksq" in PASSES:
        trk_sq_summary()


def trk_sq_summary():
    """SQ counters of every tracking kernel the bench line runs (one --pmc pass over bench.py with its extras), mean per launch."""
    acc = {}
    for r in csv.DictReader(open(find("prof_trk_sq", "counter_collection.csv"))):
        n = short(r["Kernel_Name"])
        if n.startswith("trk_"):
            e = acc.setdefault(n, {}).setdefault(r["Counter_Name"], [0.0, 0])
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    out = {"round": int(TAG[1:3]), "command": "profiles/collect.sh pass trksq (bench.py --no-cpu --no-acq --preroll-ms 0 --segments 0 --steps 3)",
        "units": "SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles (MI355X_MICROARCH.md); SQ_INSTS_* wave-instructions",
        "kernels": {}}
    for n, cs in sorted(acc.items()):
        m = {c: v[0] / v[1] for c, v in sorted(cs.items())}
        m["launches"] = max(v[1] for v in cs.values())
        if m.get("SQ_WAVE_CYCLES"):
            m["valu_active_share_of_wave_cycles"] = m.get("SQ_ACTIVE_INST_VALU", 0.0) / m["SQ_WAVE_CYCLES"]
            m["wait_inst_share_of_wave_cycles"] = m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
            m["wait_any_share_of_wave_cycles"] = m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
        if m.get("SQ_WAVES"):
            m["valu_instructions_per_wave"] = m.get("SQ_INSTS_VALU", 0.0) / m["SQ_WAVES"]
        out["kernels"][n] = m
    json.dump(out, open(os.path.join(PROF, TAG + "_trk_sq_counters.json"), "w"), indent=1)
    for n, m in out["kernels"].items():
        print("%-90s VALU %.0f /wave, VALU-active %.2f, wait-inst %.2f, wait-any %.2f" % (n[:90], m.get("valu_instructions_per_wave", 0), m.get("valu_active_share_of_wave_cycles", 0),
            m.get("wait_inst_share_of_wave_cycles", 0), m.get("wait_any_share_of_wave_cycles", 0)))


def trk_summary():
    shutil.copy(find("prof_trk", "kernel_stats.csv"), os.path.join(PROF, TAG + "_tracking_kernel_stats.csv"))
    # per-dispatch trace of the headline kernel (driver-shaped run: pre-roll, 5 warm-up, 20 timed, then the cold diagnostic pass): start
    # relative to the first dispatch, duration and the gap to the previous one -- the evidence behind the start-of-load transient
    rows = [r for r in csv.DictReader(open(find("prof_trk", "kernel_trace.csv"))) if r["Kernel_Name"].startswith("void " + KERNEL)]
    with open(os.path.join(PROF, TAG + "_tracking_kernel_trace.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["dispatch", "start_us_since_first", "duration_us", "gap_to_previous_us"])
        t0, prev_end = int(rows[0]["Start_Timestamp"]), None
        for i, r in enumerate(rows):
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            w.writerow([i, "%.1f" % ((st - t0) / 1e3), "%.1f" % ((en - st) / 1e3), "%.1f" % (((st - prev_end) / 1e3) if prev_end else 0.0)])
            prev_end = en
    fetch, nf = counter_mean(find("prof_fetch", "counter_collection.csv"), "FETCH_SIZE")
    write, nw = counter_mean(find("prof_write", "counter_collection.csv"), "WRITE_SIZE")
    for name, src in (("fetch", find("prof_fetch", "counter_collection.csv")), ("write", find("prof_write", "counter_collection.csv"))):
        rows = [r for r in csv.DictReader(open(src)) if r["Kernel_Name"].startswith("void " + KERNEL)]
        with open(os.path.join(PROF, "%s_tracking_%s_size.csv" % (TAG, name)), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            w.writeheader()
            for r in rows:
                w.writerow({k: (r[k][:80] if k == "Kernel_Name" else r[k]) for k in w.fieldnames})
    traffic = {
        "round": int(TAG[1:3]), "kernel": KERNEL,
        "workload": "bench.py default (32 ch x 256 epochs x 25000 samples, distinct IQ per channel)",
        "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write, "dispatches": [nf, nw],
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streaming loads, so it is doubled "
                      "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact",
        "hbm_bytes_per_launch": 2.0 * fetch * 1024.0 + write * 1024.0,
        "algorithmic_bytes_per_launch": 32 * 256 * 25000 * 8,
        "command": "profiles/collect.sh (separate --pmc passes with --kernel-trace only)",
    }
    json.dump(traffic, open(os.path.join(PROF, "traffic_latest.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
