#!/bin/bash
# Same-box A/B of bench.py flag sets / libraries for ONE config, short runs (no CPU leg, no extra passes):
#   bash tools/ab_cfg.sh <tag> <C2|C3> <reps> "flags A" "flags B" ...     ("-" = none; "LIB=name ..." selects build/ab/libsmx_<name>.so)
TAG=$1; CFG=$2; REPS=$3; shift; shift; shift
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out; mkdir -p $OUT
STEPS=300; WARM=20; [ $CFG = C3 ] && { STEPS=100; WARM=10; }
for rep in $(seq $REPS); do
  for v in "$@"; do
    f="$v"; [ "$v" = "-" ] && f=""
    unset SMX_LIB_PATH
    case "$f" in LIB=*) n="${f%% *}"; n="${n#LIB=}"; export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_$n.so; f="${f#LIB=$n}";; esac
    timeout 300 python bench.py --full-line --config $CFG --steps $STEPS --warmup $WARM --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet $f 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; k=r.get('kernels',{}); tl=d.get('in_frame_timeline_us') or {}
        print('%-22s %7.1f | period %.1f | in-frame:' % ('$v', d['value'], tl.get('period (integrate begin -> next integrate begin)') or 0), ' '.join('%s %.0f' % (n[:8], (v.get('in_frame_ms') or 0)*1e3) for n,v in k.items()))" | tee -a $OUT/${TAG}_$CFG.txt
  done
done
