#!/bin/bash
# Same-box A/B of neighbour-search builds at C5 (after the nn GPU tests of the in-tree build): bash tools/r17_c5ab.sh <tag> name...
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_nn.py -m gpu -x -q 2>&1 | tail -4
for v in "$@"; do
  if [ $v = NEW ]; then unset SMX_LIB_PATH; else export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_$v.so; fi
  timeout 600 python bench.py --config C5 --cpu-frames 0 --quiet 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-8s %.3f G q/s  step %.2f ms | build %.2f ms | r x2 %.3f G/s | batch %.3f G/s' % ('$v', d['value']/1e9, d['ms_per_step'], d['index_build']['ms'], d['radius_x2']['queries_per_s']/1e9, d['general_batch_entry_point']['queries_per_s']/1e9))" | tee -a $OUT/${TAG}_c5.txt
done
