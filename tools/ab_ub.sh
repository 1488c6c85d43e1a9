#!/bin/bash
# Timing-only upper bounds, same box:  bash tools/ab_ub.sh <tag> <reps> "" no-front-wait no-upd-wait "no-front-wait,no-upd-wait" ...
TAG=$1; REPS=$2; shift; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in $(seq $REPS); do
  for v in "$@"; do
    f=""; [ -n "$v" ] && [ "$v" != "-" ] && f="--ub $v"
    timeout 200 python bench.py --full-line --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet $f 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-28s %7.1f frames/s' % ('${v:--}', d['value']))" | tee -a gpurun_out/${TAG}_ub.txt
  done
done
