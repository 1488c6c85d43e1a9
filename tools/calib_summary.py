#!/usr/bin/env python
"""profiles/r17_counter_calibration.md from the outputs of tools/calib.sh: FETCH_SIZE / WRITE_SIZE (KB, per dispatch) of
the known-byte-count kernels of tools/calib.hip against the bytes they are known to move, and the VALU issue rates.

    python tools/calib_summary.py <rocprof dir FETCH_SIZE> <rocprof dir WRITE_SIZE> <log with the 'expect' lines> <valu.txt>"""
import csv
import glob
import os
import re
import sys


def counters(d):
    cc = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    out = {}
    if not cc:
        return out
    for r in csv.DictReader(open(cc[-1])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
        out.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return out


def main():
    f, w = counters(sys.argv[1]), counters(sys.argv[2])
    print("# Counter calibration (tools/calib.hip, tools/calib.sh): what do FETCH_SIZE / WRITE_SIZE report for known byte counts?\n")
    print("One launch per kernel over a 3 GiB buffer (12 x the Infinity Cache); FETCH_SIZE and WRITE_SIZE in KB from separate")
    print("rocprofv3 --pmc passes.  `read x` = known bytes read / (FETCH_SIZE x 1024): the factor FETCH_SIZE has to be multiplied")
    print("with for this pattern; `write x` likewise.  Known bytes count every byte a lane requests once (16 per record), NOT the")
    print("64- or 128-byte lines they sit in -- a factor below 1 therefore is the line granularity showing, not a counter error.\n")
    print("| kernel | pattern | known read MB | FETCH_SIZE MB | read x | known written MB | WRITE_SIZE MB | write x |")
    print("|---|---|---|---|---|---|---|---|")
    for line in open(sys.argv[3]):
        m = re.match(r"expect (\S+) (\d+) (\d+)(?: # (.*))?", line)
        if not m:
            continue
        k, rb, wb, note = m.group(1), int(m.group(2)), int(m.group(3)), (m.group(4) or "")
        fk = f.get(k, {}).get("FETCH_SIZE")
        wk = w.get(k, {}).get("WRITE_SIZE")
        fs = "%.1f" % (fk * 1024 / 1e6) if fk is not None else "-"
        ws = "%.1f" % (wk * 1024 / 1e6) if wk is not None else "-"
        rx = "%.3f" % (rb / (fk * 1024)) if fk and rb else "-"
        wx = "%.3f" % (wb / (wk * 1024)) if wk and wb else "-"
        print("| %s | %s | %.1f | %s | %s | %.1f | %s | %s |" % (k, note, rb / 1e6, fs, rx, wb / 1e6, ws, wx))
    print("\n## VALU issue (calib valu)\n")
    print("256-lane workgroups (one wavefront per SIMD of a CU), `waves/SIMD` workgroups per CU, 16 independent accumulators per")
    print("lane, 64 instructions per loop body.\n\n```")
    sys.stdout.write(open(sys.argv[4]).read())
    print("```")


if __name__ == "__main__":
    main()
