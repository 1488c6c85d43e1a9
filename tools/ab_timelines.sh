#!/bin/bash
# Same-box runs with the stamp timeline of each (one JSON line per run: flags, value, timeline, in-frame kernel times):
#   bash tools/ab_timelines.sh <tag> <C2|C3> <reps> "flags A" "flags B" ...     ("-" = none; "LIB=name ..." as in ab_cfg.sh)
TAG=$1; CFG=$2; REPS=$3; shift; shift; shift
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out; mkdir -p $OUT
STEPS=300; WARM=20; [ $CFG = C3 ] && { STEPS=100; WARM=10; }
for rep in $(seq $REPS); do
  for v in "$@"; do
    f="$v"; [ "$v" = "-" ] && f=""
    unset SMX_LIB_PATH
    case "$f" in LIB=*) n="${f%% *}"; n="${n#LIB=}"; export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_$n.so; f="${f#LIB=$n}";; esac
    timeout 300 python bench.py --full-line --config $CFG --steps $STEPS --warmup $WARM --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet $f 2>/dev/null | V="$v" python -c "import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline'].get('kernels',{})
        print(json.dumps({'flags': os.environ['V'], 'v': round(d['value'],1), 'tl': d.get('in_frame_timeline_us'), 'k': {n: round((v.get('in_frame_ms') or 0)*1e3,1) for n,v in k.items()}}))" >> $OUT/${TAG}_$CFG.jsonl
  done
done
python - <<PY
import json
for l in open("$OUT/${TAG}_$CFG.jsonl"):
    d=json.loads(l); tl=d['tl'] or {}
    g=lambda s: next((v for k,v in tl.items() if k.startswith(s)), 0)
    print('%-26s %7.1f period %.1f | front %.1f | step->int %.1f blend->int %.1f upd->passA %.1f | scan %.0f tiles %.0f blend %.0f int %.0f upd %.0f pB %.0f acc %.0f step %.0f' % (d['flags'], d['v'], g('period'), g('front'), g('internal stream: step'), g('hand-over to the internal'), g('hand-over to the caller'), g('scan_visible'), g('assoc_tiles'), g('blend'), g('integrate'), g('update'), g('neighbor_scan'), g('reg_accumulate'), g('reg_step')))
PY
