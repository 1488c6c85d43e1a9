#!/bin/bash
# how often does a run fall into the slow mode (period ~165 us instead of ~150)?  bash tools/r5_modes.sh <tag> <runs> [ENV=VAL ...]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=$1; N=$2; shift; shift
for e in "$@"; do export "$e"; done
for i in $(seq $N); do
  timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --quiet $SMX_BENCH_FLAGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d.get('in_frame_timeline_us') or {}
h=d.get('handover_probe_us') or {}
print('%s %7.1f period %.1f internal-wait %.1f probe %s | %s' % ('$T', d['value'], t.get('period (integrate begin -> next integrate begin)',0), t.get('internal stream: step end -> next integrate begin',0), list((h.get('before_the_run') or {}).values()), list((h.get('behind_the_timed_window') or {}).values())))" | tee -a gpurun_out/${T}_modes.txt
done
