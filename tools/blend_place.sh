#!/bin/bash
# Who stands in the way of the blend's workgroups?  -DSMX_STAMPS build, N runs per flag set, the recorded frame's blend entries
# (tools/blend_modes.py).   bash tools/blend_place.sh <tag> <runs> "flags A" "flags B" ...   ("-" = none)
TAG=$1; N=$2; shift; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_stamps.so
k=0
for v in "$@"; do
  f="$v"; [ "$v" = "-" ] && f=""
  k=$((k+1))
  for rep in $(seq $N); do
    timeout 300 python bench.py --config C2 --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet $f --dump-stamps gpurun_out/${TAG}_v${k}_stamps_$rep.npz > /dev/null 2>&1
  done
  echo "== $v"; python tools/blend_modes.py gpurun_out/${TAG}_v${k}_stamps_*.npz | grep -v "gate:" | cut -c1-150
done
