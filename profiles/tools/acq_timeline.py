#!/usr/bin/env python3
"""Timeline of ONE acquisition search from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): every kernel of the search with its
start offset, duration and the idle gap in front of it, and the sums (busy / gaps / span).  The search is delimited by its statistics
kernel (acq_final_kernel); `--which k` picks the k-th search from the end (default 3: a steady one).

    python3 profiles/tools/acq_timeline.py gpurun_out/<dir>/<host>/<pid>_kernel_trace.csv [--which 3]"""
import argparse
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--which", type=int, default=3)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.csv)):
        name = r["Kernel_Name"]
        if not name.startswith("acq_") and "acq_" not in name.split("(")[0]:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0]))
    rows.sort()
    finals = [i for i, r in enumerate(rows) if r[2].startswith("acq_final_kernel")]
    assert len(finals) > a.which + 1, "not enough searches in the trace"
    lo, hi = finals[-a.which - 1] + 1, finals[-a.which] + 1
    seg = rows[lo:hi]
    t0 = seg[0][0]
    prev_end = None
    out = []
    for s, e, n in seg:
        gap = 0 if prev_end is None else s - prev_end
        out.append({"kernel": n, "start_us": (s - t0) / 1e3, "dur_us": (e - s) / 1e3, "gap_before_us": gap / 1e3})
        prev_end = max(e, prev_end or e)
    busy = sum(o["dur_us"] for o in out)
    gaps = sum(max(0.0, o["gap_before_us"]) for o in out)
    # distance to the next search's first kernel: the period of back-to-back searches
    nxt = rows[hi][0] if hi < len(rows) else None
    summary = {"kernels": len(out), "busy_us": busy, "gaps_us": gaps, "span_us": (seg[-1][1] - t0) / 1e3, "period_us": (nxt - t0) / 1e3 if nxt else None}
    if a.json:
        print(json.dumps({"timeline": out, "summary": summary}))
        return
    for o in out:
        print("%9.1f  +%6.1f  %7.1f  %s" % (o["start_us"], o["gap_before_us"], o["dur_us"], o["kernel"][:110]))
    print(summary)


if __name__ == "__main__":
    main()
