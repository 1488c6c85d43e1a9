#!/bin/bash
# Same-box A/B of build/ab/libsmx_<name>.so builds against the in-tree library at C2 and C3, after the GPU tests of the
# in-tree build.   bash tools/r17_ab.sh <tag> <reps> name1 [name2 ...]      (NEW = in-tree)
TAG=$1; REPS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/${TAG}_gputests.txt
cat $OUT/${TAG}_gputests.txt
for cfg in C2 C3; do
  echo "== $cfg"
  SMX_BENCH_FLAGS="--config $cfg" bash tools/ab_libs.sh ${TAG}_$cfg $REPS "$@"
done
