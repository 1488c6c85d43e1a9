#!/usr/bin/env python
"""The PCIe-inclusive pass of bench.py in a rocprofv3 kernel trace: the stretch between the first and the last k_copy_rows
dispatch -- per-kernel averages, per-queue busy time, the period per frame.    python tools/host_pass_trace.py <dir-or-csv> [out.md]"""
import csv
import re
import sys

from prof_summary import find_csv


def main():
    rows = list(csv.DictReader(open(find_csv(sys.argv[1]))))
    for r in rows:
        r["_s"], r["_e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["_s"])
    cp = [r for r in rows if "k_copy_rows" in r["Kernel_Name"]]
    if len(cp) < 8:
        raise SystemExit("no k_copy_rows dispatches in the trace")
    cp = cp[len(cp) // 4:]                      # (skip the pass's warm-up frames)
    lo, hi = cp[0]["_s"], cp[-1]["_e"]
    sel = [r for r in rows if r["_s"] >= lo and r["_e"] <= hi]
    name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    agg, queues = {}, {}
    for r in sel:
        a = agg.setdefault(name(r), [0, 0]); a[0] += 1; a[1] += r["_e"] - r["_s"]
        q = queues.setdefault(r.get("Queue_Id", "?"), [0, set()]); q[0] += r["_e"] - r["_s"]; q[1].add(name(r))
    frames = sum(1 for r in sel if "k_scan_visible" in r["Kernel_Name"])
    out = ["# PCIe-inclusive pass: %d frames in %.3f ms = %.1f us per frame" % (frames, (hi - lo) / 1e6, (hi - lo) / 1e3 / max(frames, 1)), "",
           "| kernel | calls | avg us |", "|---|---|---|"]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append("| %s | %d | %.2f |" % (n, c, t / c / 1e3))
    out += ["", "| queue | busy % | kernels |", "|---|---|---|"]
    for q, (t, names) in sorted(queues.items(), key=lambda kv: -kv[1][0]):
        out.append("| %s | %.0f | %s |" % (q, 100.0 * t / (hi - lo), ", ".join(sorted(names))[:150]))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
