#!/bin/bash
# Same-box runs of timing-only upper bounds (bench.py --ub ..., whose line has no roofline block):  bash tools/ab_ub.sh <reps> "flags A" "flags B" ...  ("-" = none)
REPS=$1; shift
cd "$GRAFT_REPO_ROOT"
for rep in $(seq $REPS); do
  for v in "$@"; do
    f="$v"; [ "$v" = "-" ] && f=""
    timeout 300 python bench.py --config ${CFG:-C2} --steps ${STEPS:-300} --warmup 20 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet $f 2>/dev/null | V="$v" python -c "import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-34s %.1f' % (os.environ['V'], d['value']))"
  done
done | sort -k1,1 -s
