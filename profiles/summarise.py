"""Turns the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof_*) into the small files committed under
profiles/: per-kernel duration tables and traffic_latest.json (HBM bytes per launch of the tracking kernel, with the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md's HBM section: the counter reports 64 B per 128-B request of a
16-B/lane streaming load, so it is doubled; WRITE_SIZE is exact).  Usage: python profiles/summarise.py [round_tag]"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
KERNEL = "trk_multicorrelator_kernel<3, false, false, 0, false, false>"


def counter_mean(path, counter):
    vals = []
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter and r["Kernel_Name"].startswith("void " + KERNEL):
            vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


def main():
    shutil.copy(os.path.join(OUT, "prof_full", "full_kernel_stats.csv"), os.path.join(PROF, TAG + "_full_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(OUT, "prof_trk", "trk_kernel_stats.csv"), os.path.join(PROF, TAG + "_tracking_kernel_stats.csv"))
    fetch, nf = counter_mean(os.path.join(OUT, "prof_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
    write, nw = counter_mean(os.path.join(OUT, "prof_write", "write_counter_collection.csv"), "WRITE_SIZE")
    for name, src in (("fetch", "prof_fetch/fetch_counter_collection.csv"), ("write", "prof_write/write_counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(os.path.join(OUT, src))) if r["Kernel_Name"].startswith("void " + KERNEL)]
        with open(os.path.join(PROF, "%s_tracking_%s_size.csv" % (TAG, name)), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            w.writeheader()
            for r in rows:
                w.writerow({k: (r[k][:80] if k == "Kernel_Name" else r[k]) for k in w.fieldnames})
    traffic = {
        "round": int(TAG[1:]), "kernel": KERNEL,
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
    for line in open(os.path.join(OUT, "prof_full.log")):
        if line.startswith("{"):
            open(os.path.join(PROF, TAG + "_bench_line_under_rocprof.json"), "w").write(line)


if __name__ == "__main__":
    main()
