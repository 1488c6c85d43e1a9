#!/bin/bash
# C5 A/B: bash tools/r5_c5.sh <tag> "<lib>:<mode>" ...   (lib NEW = in-tree)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=$1; shift
for rep in 1 2; do
for vm in "$@"; do
  v=${vm%%:*}; m=${vm##*:}
  if [ $v = NEW ]; then unset SMX_LIB_PATH; else export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_$v.so; fi
  timeout 600 python bench.py --config C5 --steps 3 --warmup 1 --cpu-frames 0 --quiet --nn-mode $m 2>/dev/null | tail -1 > gpurun_out/${T}_${v}_${m}.json
  python - <<PY
import json
d = json.load(open("gpurun_out/${T}_${v}_${m}.json"))
print("%-6s mode %s  %.3f G q/s  %.2f ms | x2: %.2f ms | batch %.2f ms | staged/q %.1f tests/q %.1f tiles %d build %.2f ms" % ("$v", "$m", d["value"] / 1e9, d["ms_per_step"],
      d["radius_x2"]["ms"], d["general_batch_entry_point"]["ms"], d["distributions"]["staged_candidates_per_query"], d["distributions"]["distance_tests_per_query"], d["distributions"]["tiles"], d["index_build"]["ms"]))
PY
done
done
