#!/bin/bash
# One bench run per argument (an environment assignment list, e.g. "SMX_EXP=1"; "-" = none), REPS times; prints frames/s,
# the pairs statistics and the kernels' stand-alone times.   bash tools/ab_env.sh <tag> <reps> "-" "SMX_EXP=1" ...
TAG=$1; REPS=$2; shift; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in $(seq $REPS); do
  for v in "$@"; do
    e="$v"; [ "$v" = "-" ] && e=""
    env $e timeout 300 python bench.py --full-line --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --quiet $SMX_BENCH_FLAGS 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; k=r.get('kernels',{}); s=d['distributions']
        print('%-14s %7.1f  %s in-frame %.1f | pairs %s ovf %s max %s | alone:' % ('$v', d['value'], r['kernel'], r['avg_launch_ms']*1e3, s.get('n_pairs'), s.get('n_overflow_pairs'), s.get('max_tile_pairs')), ' '.join('%s %.1f' % (n[:9], v['alone_ms']*1e3) for n,v in k.items()))" | tee -a gpurun_out/${TAG}_env.txt
  done
done
