#!/bin/bash
# in-frame timeline (stage stamps) of one or more builds: bash tools/r5_tl.sh <tag> <config> name1 name2 ...  (NEW = in-tree)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=$1; CFG=$2; shift; shift
for v in "$@"; do
  if [ $v = NEW ]; then unset SMX_LIB_PATH; else export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_$v.so; fi
  for rep in 1 2; do
  timeout 600 python bench.py --config $CFG --gpus 1 --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --quiet $SMX_BENCH_FLAGS 2>/dev/null | tail -1 > gpurun_out/${T}_${v}_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/${T}_${v}_$rep.json"))
t = d.get("in_frame_timeline_us") or {}
k = d["roofline"]["kernels"]
print("%-8s %7.1f |" % ("$v", d["value"]), " ".join("%s %.1f" % (k_.split(" ")[0][:9], v) for k_, v in t.items()),
      "| alone:", " ".join("%s %.1f" % (n[:9], v["alone_ms"] * 1e3) for n, v in k.items() if n in ("neighbor_scan", "reg_accumulate", "reg_step", "integrate+new_flags", "update_neighbors+create")))
PY
  done
done
