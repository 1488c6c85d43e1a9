#!/bin/bash
# quick GPU check: selected tests + one C2 bench line without the embedded configs / CPU legs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=${1:-r5q}; SEL=${2:-"timings or native or stage_times"}
timeout 900 python -m pytest tests -m gpu -x -q -k "$SEL" 2>&1 | tail -15 | tee gpurun_out/${T}_tests.txt
for rep in 1 2; do
timeout 600 python bench.py --gpus 1 --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --quiet 2> gpurun_out/${T}_bench.err | tail -1 > gpurun_out/${T}_bench_line_$rep.json
python - <<PY
import json
d = json.load(open("gpurun_out/${T}_bench_line_$rep.json"))
print("C2", round(d["value"], 1), "growth", d["growth_phase"] and {k: d["growth_phase"][k] for k in ("value", "new_slots_per_frame", "live_surfels_at_start", "live_surfels_at_end")})
print("curve", [(c["live"], c["frames_per_s"], c["new_slots_per_frame"]) for c in d["growth_phase"]["growth_curve"]])
print("timing", {k: (round(v["value"], 1), round(v["vs_off"], 3)) for k, v in d["stage_timing_cost"].items() if isinstance(v, dict)})
print("read", d["stage_timing_cost"]["stamps_read_every_frame_nowait"])
print("stage_ms", {k: round(v, 4) for k, v in d["stage_ms"].items()}, {k: round(v, 4) for k, v in d["stage_ms_by_event_records"].items()})
PY
done
tail -3 gpurun_out/${T}_bench.err
