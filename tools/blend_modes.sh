#!/bin/bash
# Diagnosis of the slow mode: N bench runs with a -DSMX_STAMPS build, the per-workgroup wall clocks of the last frame's blend and
# edge kernel next to the run's frame rate (tools/blend_modes.py prints them).   bash tools/blend_modes.sh <tag> <runs>
TAG=$1; N=${2:-8}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export SMX_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsmx_stamps.so
for rep in $(seq $N); do
  timeout 300 python bench.py --config C2 --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet --dump-stamps gpurun_out/${TAG}_stamps_$rep.npz > /dev/null 2>&1
done
python tools/blend_modes.py gpurun_out/${TAG}_stamps_*.npz
