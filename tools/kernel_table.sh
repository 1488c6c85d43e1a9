#!/bin/bash
# bash tools/kernel_table.sh "ENV=.. [-- bench flags]" ...   -- per-kernel times (untimed pass, events around every launch) per environment
for v in "$@"; do
  envs="${v%%--*}"; flags=""; case "$v" in *--*) flags="${v#*--}";; esac
  env $envs timeout 300 python bench.py --full-line $flags --steps 200 --warmup 20 --cpu-frames 0 --host-frames 0 --quiet 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernels']
        print('$v', round(d['value'],1), ' '.join('%s=%.1f' % (n, v['alone_ms']*1e3) for n, v in k.items()))"
done
