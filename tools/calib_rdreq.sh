#!/bin/bash
# Second calibration pass: the request-size counters behind FETCH_SIZE (its expression books 128-byte requests under
# TCC_BUBBLE, which stays 0 on gfx950 -- hence the factor 2 on streams and 1 on 64-byte gathers).   bash tools/calib_rdreq.sh <tag>
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out; mkdir -p $OUT
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/calib_$N
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/calib_$N -o run -- build/calib pmc > /tmp/calib_$N.log 2>&1 || tail -3 /tmp/calib_$N.log
  python - "$N" <<'PY' >> $OUT/${TAG}_calib_requests.md
import csv, glob, os, re, sys
d = "/tmp/calib_" + sys.argv[1]
cc = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
if not cc:
    print("(no output for %s)" % sys.argv[1]); sys.exit(0)
agg = {}
for r in csv.DictReader(open(cc[-1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
    agg.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for k in agg for c in agg[k]})
print("| kernel | " + " | ".join(names) + " |\n|---|" + "---|" * len(names))
for k in agg:
    if k.startswith("cal_"):
        print("| %s | " % k + " | ".join("%.4g" % agg[k].get(c, 0) for c in names) + " |")
print()
PY
done
cat $OUT/${TAG}_calib_requests.md
