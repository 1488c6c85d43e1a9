#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters inside the timed region of a bench.py run (between the two
k_smx_marker dispatches).  Usage: python tools/pmc_summary.py <rocprofv3 output dir> [out.md] [merge.json]

With a third argument the averages are merged into that JSON file ({kernel: {counter: average per launch}}), so
that separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit into one) end up in one place;
profiles/pmc_traffic.json is what bench.py reads for roofline.traffic."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)


def main():
    d = sys.argv[1]
    cc = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(cc)))
    # dispatch ids of the markers
    ids = sorted(int(r["Dispatch_Id"]) for r in rows if "k_smx_marker" in r["Kernel_Name"])
    ids = sorted(set(ids))
    lo, hi = (ids[0], ids[1]) if len(ids) >= 2 else (-1, 1 << 62)
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for r in rows:
        di = int(r["Dispatch_Id"])
        if not (lo < di < hi):
            continue
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(di)
    counters = sorted({c for k in agg for c in agg[k]})
    lines = ["# PMC per-kernel averages over the timed region (%s)" % os.path.relpath(cc, d), "",
             "| kernel | launches | " + " | ".join(counters) + " |", "|---|---|" + "---|" * len(counters)]
    for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
        n = len(cnt[k])
        lines.append("| %s | %d | " % (k, n) + " | ".join("%.4g" % (agg[k][c] / n) for c in counters) + " |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    if len(sys.argv) > 3:
        merged = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
        for k in agg:
            merged.setdefault(k, {}).update({c: agg[k][c] / len(cnt[k]) for c in agg[k]})
            merged[k]["launches"] = len(cnt[k])
        json.dump(merged, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    print(out)


if __name__ == "__main__":
    main()
