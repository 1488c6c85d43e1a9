#!/bin/bash
# The kernel-trace half of tools/profile_round.sh (no PMC passes): summary of the timed region, two-frame timeline,
# kernel stats, and the plain bench line.  Usage: bash tools/profile_trace_only.sh <tag>
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
CMD="python bench.py --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --quiet"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o run -- $CMD > $OUT/${TAG}_bench.log 2>&1
python tools/prof_summary.py /tmp/prof_trace $OUT/${TAG}_bench_timed_region_summary.md > /dev/null
python tools/prof_timeline.py /tmp/prof_trace 2 $OUT/${TAG}_timeline_two_frames.md > /dev/null
cp "$(ls /tmp/prof_trace/*/*kernel_stats.csv /tmp/prof_trace/*kernel_stats.csv 2>/dev/null | tail -1)" $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null
timeout 600 python bench.py 2> $OUT/${TAG}_bench_plain.err | tail -1 > $OUT/${TAG}_bench_line.json
cut -c1-300 $OUT/${TAG}_bench_line.json
