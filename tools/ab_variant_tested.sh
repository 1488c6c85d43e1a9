#!/bin/bash
# GPU tests of ONE variant library, then the same-box A/B at C2 and C3:  bash tools/ab_variant_tested.sh <tag> <reps> <variant> [others...]
TAG=$1; REPS=$2; V=$3; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out; mkdir -p $OUT
SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_$V.so timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $OUT/${TAG}_gputests_$V.txt
for cfg in C2 C3; do
  echo "== $cfg"
  SMX_BENCH_FLAGS="--config $cfg" bash tools/ab_libs.sh ${TAG}_$cfg $REPS NEW "$@"
done
