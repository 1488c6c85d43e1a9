#!/usr/bin/env python
"""Prints the dispatch timeline of a few frames from the middle of bench.py's timed region (rocprofv3
--kernel-trace CSV): start offset, duration, queue, kernel -- to see what runs beside what.

    python tools/prof_timeline.py <dir-or-csv> [n_frames] [out.md]
"""
import csv
import re
import sys

from prof_summary import find_csv


def main():
    src = find_csv(sys.argv[1])
    nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = list(csv.DictReader(open(src)))
    for r in rows:
        r["_s"], r["_e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["_s"])
    marks = [r for r in rows if "k_smx_marker" in r["Kernel_Name"]]
    lo, hi = (marks[0]["_e"], marks[1]["_s"]) if len(marks) >= 2 else (rows[0]["_s"], rows[-1]["_e"])
    sel = [r for r in rows if r["_s"] >= lo and r["_e"] <= hi]
    # frames are delimited by the dispatches of pass A (k_scan_visible), the first launch of an Integrate call
    clears = [i for i, r in enumerate(sel) if "k_scan_visible" in r["Kernel_Name"]]
    mid = len(clears) // 2
    a, b = clears[mid], clears[min(mid + nfr, len(clears) - 1)]
    t0 = sel[a]["_s"]
    qkey = "Queue_Id" if "Queue_Id" in sel[0] else None
    queues = {}
    lines = ["| start us | dur us | end us | queue | kernel |", "|---|---|---|---|---|"]
    for r in sel[a:b]:
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        q = queues.setdefault(r[qkey], len(queues)) if qkey else 0
        lines.append("| %8.1f | %6.1f | %8.1f | %d | %s |" % ((r["_s"] - t0) / 1e3, (r["_e"] - r["_s"]) / 1e3,
                                                          (r["_e"] - t0) / 1e3, q, name))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
