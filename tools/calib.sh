#!/bin/bash
# Counter / issue-rate calibration on the GPU box: build/calib was compiled in the authoring container
# (hipcc --offload-arch=gfx950 -O3 tools/calib.hip -o build/calib).   bash tools/calib.sh <tag>
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
timeout 300 build/calib valu > $OUT/${TAG}_calib_valu.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/calib_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/calib_$C -o run -- build/calib pmc > $OUT/${TAG}_calib_pmc_$C.log 2>&1
done
python tools/calib_summary.py /tmp/calib_FETCH_SIZE /tmp/calib_WRITE_SIZE $OUT/${TAG}_calib_pmc_FETCH_SIZE.log $OUT/${TAG}_calib_valu.txt > $OUT/${TAG}_counter_calibration.md
cat $OUT/${TAG}_counter_calibration.md
