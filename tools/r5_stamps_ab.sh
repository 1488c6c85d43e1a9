#!/bin/bash
# cost of the stage stamps in the timed window: bench.py with --stage-timing stamps (default) / off, alternating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=${1:-r5s}; CFG=${2:-C2}
for rep in 1 2 3; do
for m in stamps off; do
  timeout 600 python bench.py --config $CFG --gpus 1 --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --timing-frames 100 --growth-frames 0 --quiet --stage-timing $m 2>/dev/null | tail -1 > gpurun_out/${T}_${CFG}_${m}_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/${T}_${CFG}_${m}_$rep.json"))
c = d.get("stage_timing_cost") or {}
print("%-7s %7.1f | interleaved:" % ("$m", d["value"]), {k: (round(v["value"]), round(v["vs_off"], 3)) for k, v in c.items() if isinstance(v, dict)})
PY
done
done
