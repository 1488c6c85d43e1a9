#!/bin/bash
# Same-box comparison of several builds: bash tools/ab_libs.sh <tag> <reps> name1 name2 ...   (build/ab/libsmx_<name>.so;
# the name NEW = the in-tree library).  Prints frames/s and the kernels' stand-alone times of the untimed calibration pass.
TAG=$1; REPS=$2; shift; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in $(seq $REPS); do
  for v in "$@"; do
    if [ $v = NEW ]; then unset SMX_LIB_PATH; else export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_$v.so; fi
    timeout 300 python bench.py --full-line --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --quiet $SMX_BENCH_FLAGS 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; k=r.get('kernels',{})
        print('%-8s %7.1f  %s in-frame %.1f | alone:' % ('$v', d['value'], r['kernel'], r['avg_launch_ms']*1e3), ' '.join('%s %.1f' % (n[:9], v['alone_ms']*1e3) for n,v in k.items()))" | tee -a gpurun_out/${TAG}_libs.txt
  done
done
