#!/bin/bash
# Same-box A/B of libraries at C5 (50 M-point neighbour search):  bash tools/ab_c5.sh <tag> <reps> NEW name ...   (build/ab/libsmx_<name>.so)
TAG=$1; REPS=$2; shift; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in $(seq $REPS); do
  for v in "$@"; do
    if [ $v = NEW ]; then unset SMX_LIB_PATH; else export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_$v.so; fi
    timeout 600 python bench.py --full-line --config C5 --steps 3 --warmup 1 --cpu-frames 0 --quiet 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); ds=d['distributions']
        print('%-8s %.3f G queries/s  step %.2f ms | tests/query %.1f staged/query %.1f | r x2: %.2f G/s | batch entry: %.2f G/s | build %.2f ms' % ('$v', d['value']/1e9, d['ms_per_step'], ds['distance_tests_per_query'], ds['staged_candidates_per_query'], d['radius_x2']['queries_per_s']/1e9, d['general_batch_entry_point']['queries_per_s']/1e9, d['index_build']['ms']))" | tee -a gpurun_out/${TAG}_c5.txt
  done
done
