#!/usr/bin/env python
"""Summarises a rocprofv3 --kernel-trace CSV: per-kernel count / total / average duration, restricted to the
dispatches between the two k_smx_marker kernels bench.py launches around its timed region.

    python tools/prof_summary.py <dir-or-csv> [out.md] [out.json key]

With `out.json key` the per-kernel averages are also merged into out.json under `key` (profiles/trace_timed_region[_C3].json,
what bench.py quotes beside the launch's own start / stop events: roofline.trace).
"""
import csv
import json
import glob
import os
import re
import sys


def find_csv(path):
    if os.path.isfile(path):
        return path
    c = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    if not c:
        raise SystemExit("no *kernel_trace.csv under " + path)
    return c[-1]


def main():
    src = find_csv(sys.argv[1])
    rows = list(csv.DictReader(open(src)))
    for r in rows:
        r["_s"], r["_e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["_s"])
    marks = [r for r in rows if "k_smx_marker" in r["Kernel_Name"]]
    lo, hi = (marks[0]["_e"], marks[1]["_s"]) if len(marks) >= 2 else (rows[0]["_s"], rows[-1]["_e"])
    sel = [r for r in rows if r["_s"] >= lo and r["_e"] <= hi and "k_smx_marker" not in r["Kernel_Name"]]
    agg = {}
    for r in sel:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += r["_e"] - r["_s"]
    total = sum(a[1] for a in agg.values())
    span = hi - lo
    lines = ["# kernel summary of the timed region (%s)" % os.path.basename(src), "",
             "region span %.3f ms, %d dispatches, busy %.3f ms (%.1f %% of span)" % (
                 span / 1e6, len(sel), total / 1e6, 100.0 * total / max(span, 1)), "",
             "| kernel | calls | total ms | avg us | % of busy |", "|---|---|---|---|---|"]
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.3f | %.2f | %.1f |" % (name, n, t / 1e6, t / n / 1e3, 100.0 * t / max(total, 1)))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    if len(sys.argv) > 4:
        path, key = sys.argv[3], sys.argv[4]
        d = json.load(open(path)) if os.path.exists(path) else {}
        frames = max((n for name, (n, t) in agg.items() if "k_scan_visible" in name or "k_query_lanes" in name), default=0)
        d[key] = {"span_ms": span / 1e6, "dispatches": len(sel), "frames": frames,
                  "kernels": {name: {"calls": n, "avg_us": t / n / 1e3} for name, (n, t) in agg.items()}}
        json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    print(out)


if __name__ == "__main__":
    main()
