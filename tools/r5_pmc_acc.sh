#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the regulariser kernels at C2 (one PMC pass each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=${1:-r5p}
CMDS="python bench.py --steps 20 --warmup 5 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet"
for C in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/prof_pmc
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_pmc -o run -- $CMDS > /tmp/pmc_$C.log 2>&1
  python tools/pmc_summary.py /tmp/prof_pmc gpurun_out/${T}_${C}.md /tmp/${T}.json > /dev/null || tail -5 /tmp/pmc_$C.log
done
python - <<PY
import json
d = json.load(open("/tmp/${T}.json"))
for k, v in d.items():
    if k.startswith("k_reg") or k.startswith("k_neighbor"):
        print(k, {a: round(b / 1024, 1) for a, b in v.items() if a != "launches"}, "MB (FETCH x2 = read)")
PY
