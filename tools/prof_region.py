#!/usr/bin/env python
"""The whole timed region of a short bench.py run (rocprofv3 --kernel-trace CSV): where do fill and drain go?  Prints the
offsets of the region's first and last kernels from the markers, the start-to-start period of pass A per frame and the
first / last dispatches.

    python tools/prof_region.py <dir-or-csv> [out.md]
"""
import csv
import re
import sys

from prof_summary import find_csv


def main():
    src = find_csv(sys.argv[1])
    rows = list(csv.DictReader(open(src)))
    for r in rows:
        r["_s"], r["_e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["_s"])
    marks = [r for r in rows if "k_smx_marker" in r["Kernel_Name"]]
    lo, hi = marks[0]["_e"], marks[1]["_s"]
    sel = [r for r in rows if r["_s"] >= lo and r["_e"] <= hi]
    name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    scans = [r for r in sel if "k_scan_visible" in r["Kernel_Name"]]
    out = ["region between the markers: %.1f us, %d dispatches, %d frames" % ((hi - lo) / 1e3, len(sel), len(scans)),
           "first dispatch %.1f us after the first marker, last one ends %.1f us before the second" %
           ((sel[0]["_s"] - lo) / 1e3, (hi - sel[-1]["_e"]) / 1e3),
           "pass A start-to-start periods (us): " + " ".join("%.0f" % ((b["_s"] - a["_s"]) / 1e3) for a, b in zip(scans, scans[1:])),
           "first pass A starts %.1f us after the marker; the last pass A starts %.1f us before the second marker" %
           ((scans[0]["_s"] - lo) / 1e3, (hi - scans[-1]["_s"]) / 1e3), "", "first 14 and last 10 dispatches:"]
    qs = {}
    for r in sel[:14] + sel[-10:]:
        q = qs.setdefault(r.get("Queue_Id", 0), len(qs))
        out.append("%9.1f %7.1f  q%d %s" % ((r["_s"] - lo) / 1e3, (r["_e"] - r["_s"]) / 1e3, q, name(r)))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
