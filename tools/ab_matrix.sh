#!/bin/bash
# (an argument may also carry bench.py flags after '--': "VAR=a -- --no-reg-culling")
# bash tools/ab_matrix.sh "VAR=a VAR2=b" "VAR=c" ...   -- one bench run per argument (an environment), twice
for rep in 1 2; do
for v in "$@"; do
  envs="${v%%--*}"; flags=""; case "$v" in *--*) flags="${v#*--}";; esac
  env $envs timeout 300 python bench.py --full-line $flags --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --quiet 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$v', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms']*1e3,1), round(d['roofline_valu']['avg_launch_ms_alone']*1e3,1))"
done; done
