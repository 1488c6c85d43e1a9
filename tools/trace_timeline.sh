#!/bin/bash
# kernel-trace timeline of the timed region for the current environment: bash tools/trace_timeline.sh <tag>
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o run -- python bench.py --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --quiet $SMX_BENCH_FLAGS > gpurun_out/${TAG}_bench.log 2>&1
python tools/prof_summary.py /tmp/prof_trace gpurun_out/${TAG}_summary.md > /dev/null
python tools/prof_timeline.py /tmp/prof_trace 2 gpurun_out/${TAG}_timeline.md > /dev/null
grep -h '^{"metric"' gpurun_out/${TAG}_bench.log | head -1 | cut -c1-140
