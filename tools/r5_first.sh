#!/bin/bash
# round 5, first GPU call: GPU tests + the driver's bench command (with the embedded C3 / C5 lines)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=${1:-r5a}
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${T}_gputests.txt
cat gpurun_out/${T}_gputests.txt
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${T}_bench.err | tail -1 > gpurun_out/${T}_bench_line.json ) 2>&1 | tail -3
tail -5 gpurun_out/${T}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${T}_bench_line.json"))
print("C2", d["value"], d["ms_per_step"], "frame frac", d["roofline"]["frame"])
print("growth", json.dumps(d.get("growth_phase"))[:900])
print("timing", json.dumps(d.get("stage_timing_cost"))[:1200])
print("stage_ms", d["stage_ms"], d["stage_ms_by_event_records"])
for k, v in (d.get("other_configs") or {}).items():
    print(k, json.dumps(v)[:700])
print("parity", d["cpu_baseline"].get("parity_check"))
PY
