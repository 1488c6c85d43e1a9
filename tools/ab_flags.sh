#!/bin/bash
# Same-box A/B of bench.py flag sets with the in-tree library (after its GPU tests), C2 and C3:  bash tools/ab_flags.sh <tag> <reps> "flags A" "flags B" ...   ("-" = none)
TAG=$1; REPS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for cfg in C2 C3; do
  echo "== $cfg"
  for rep in $(seq $REPS); do
    for v in "$@"; do
      f="$v"; [ "$v" = "-" ] && f=""
      timeout 300 python bench.py --full-line --config $cfg --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --quiet $f 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; k=r.get('kernels',{})
        print('%-18s %7.1f | in-frame:' % ('$v', d['value']), ' '.join('%s %.0f' % (n[:9], (v.get('in_frame_ms') or 0)*1e3) for n,v in k.items()), '| alone:', ' '.join('%s %.1f' % (n[:9], v['alone_ms']*1e3) for n,v in k.items() if n in ('bilateral','outlier_fusion','erode_normals_radii')))" | tee -a $OUT/${TAG}_$cfg.txt
    done
  done
done
