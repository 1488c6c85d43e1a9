#!/bin/bash
# Same-box A/B with the per-kernel picture: for build/ab/libsmx_head.so (HEAD) and the in-tree library (NEW), the plain
# bench value (x REPS) and one rocprofv3 kernel trace of the timed region (summary table + two-frame timeline).
#   bash tools/ab_trace.sh <tag> [reps] [extra bench flags]
set -u
TAG=${1:-abXX}; REPS=${2:-2}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
CMD="python bench.py --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --quiet $*"
for v in HEAD NEW; do
  if [ $v = HEAD ]; then export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_head.so; else unset SMX_LIB_PATH; fi
  [ $v = HEAD ] && [ ! -f build/ab/libsmx_head.so ] && continue
  for rep in $(seq $REPS); do
    timeout 300 $CMD 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$v', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms']*1e3,1))" | tee -a $OUT/${TAG}_values.txt
  done
  rm -rf /tmp/prof_trace
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o run -- $CMD > /tmp/ab_$v.log 2>&1
  python tools/prof_summary.py /tmp/prof_trace $OUT/${TAG}_${v}_summary.md > /dev/null
  python tools/prof_timeline.py /tmp/prof_trace 2 $OUT/${TAG}_${v}_timeline.md > /dev/null
  echo "== $v"; sed -n 3,18p $OUT/${TAG}_${v}_summary.md
done
