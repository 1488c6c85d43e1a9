#!/bin/bash
# Same-box A/B at C2 and C3 without the test run:  bash tools/ab_only.sh <tag> <reps> name1 name2 ...   (NEW = in-tree)
TAG=$1; REPS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in ${SMX_AB_CONFIGS:-C2 C3}; do
  echo "== $cfg"
  SMX_BENCH_FLAGS="--config $cfg" bash tools/ab_libs.sh ${TAG}_$cfg $REPS "$@"
done
