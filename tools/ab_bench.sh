#!/bin/bash
# Same-box A/B of two builds (box-to-box variation of the pool is a few percent, more than most single changes):
#   build the baseline, cp surfelmeshing_amd/libsmx.so build/ab/libsmx_head.so, build the candidate, then on the GPU
#   box: bash tools/ab_bench.sh [repetitions]
REPS=${1:-3}
for rep in $(seq $REPS); do
  for v in HEAD NEW; do
    if [ $v = HEAD ]; then export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_head.so; else unset SMX_LIB_PATH; fi
    timeout 300 python bench.py --full-line --steps 300 --warmup 20 --cpu-frames 0 --quiet 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$v', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms']*1e3,1))"
  done
done
