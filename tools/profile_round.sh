#!/bin/bash
# Collects a round's rocprofv3 evidence on the GPU box for one bench configuration: kernel trace of the bench run
# (summary of the timed region + a two-frame timeline) and separate PMC passes (FETCH_SIZE, WRITE_SIZE, L2 requests;
# for C5 also the SQ instruction counters).  Usage: bash tools/profile_round.sh <tag> [C2|C3|C5]
# The PMC file (gpurun_out/pmc_traffic[_C3|_C5].json; copy to profiles/) is stamped with the hash of the kernel sources
# and with the map size / counts of the PMC runs' OWN bench line, so that roofline.traffic / algorithmic bytes is
# like-for-like (bench.py refuses a file of another build or another map size).
set -u
TAG=${1:-rXX}; CFG=${2:-C2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
SUF=""; [ $CFG != C2 ] && SUF="_$CFG"
PMCJ=$OUT/pmc_traffic$SUF.json
rm -f $PMCJ
case $CFG in
  C2) CMD="python bench.py --full-line --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet"
      CMDS="python bench.py --full-line --steps 20 --warmup 5 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet";;
  C3) CMD="python bench.py --full-line --config C3 --steps 100 --warmup 10 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet"
      CMDS="python bench.py --full-line --config C3 --steps 10 --warmup 3 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet";;
  C5) CMD="python bench.py --full-line --config C5 --steps 3 --warmup 1 --cpu-frames 0 --quiet --no-other-configs"
      CMDS="python bench.py --full-line --config C5 --steps 2 --warmup 1 --cpu-frames 0 --quiet --no-other-configs";;
esac
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o run -- $CMD > $OUT/${TAG}${SUF}_bench.log 2>&1
TRJ=$OUT/trace_timed_region$SUF.json
rm -f $TRJ
python tools/prof_summary.py /tmp/prof_trace $OUT/${TAG}${SUF}_bench_timed_region_summary.md $TRJ long > /dev/null
[ $CFG != C5 ] && python tools/prof_timeline.py /tmp/prof_trace 2 $OUT/${TAG}${SUF}_timeline_two_frames.md > /dev/null
cp "$(ls /tmp/prof_trace/*/*kernel_stats.csv /tmp/prof_trace/*kernel_stats.csv 2>/dev/null | tail -1)" $OUT/${TAG}${SUF}_bench_kernel_stats.csv 2>/dev/null
# the same trace of the SHORT command (C2: the driver's --steps 20 --warmup 5): bench.py quotes the trace whose step count is its own
rm -rf /tmp/prof_trace_s
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace_s -o run -- $CMDS > $OUT/${TAG}${SUF}_bench_short.log 2>&1
python tools/prof_summary.py /tmp/prof_trace_s $OUT/${TAG}${SUF}_bench_short_timed_region_summary.md $TRJ short > /dev/null
CGROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum")
[ $CFG = C5 ] && CGROUPS+=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES")
for C in "${CGROUPS[@]}"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/prof_pmc
  timeout 900 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_pmc -o run -- $CMDS > /tmp/pmc_$N.log 2>&1
  python tools/pmc_summary.py /tmp/prof_pmc $OUT/${TAG}${SUF}_${N}.md $PMCJ > /dev/null || tail -5 /tmp/pmc_$N.log
  grep -h '^{"metric"' /tmp/pmc_$N.log | head -1 > /tmp/pmc_line_$N.json
done
python - <<PYEOF
import json, sys
sys.path.insert(0, ".")
import bench
p = "$PMCJ"
d = json.load(open(p))
line = open("/tmp/pmc_line_FETCH_SIZE.json").read().strip()
dist = json.loads(line).get("distributions", {}) if line else {}
d["_meta"] = {"source_sha": bench.source_sha(), "tag": "$TAG", "config": "$CFG",
              "surfel_slots": dist.get("surfels_size", dist.get("n_points")),
              "distributions_of_the_pmc_run": dist,
              "command": "rocprofv3 --pmc <counter group> -- $CMDS (one pass per group)"}
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
t = "$TRJ"
import os
if os.path.exists(t):
    tr = json.load(open(t))
    tr["_meta"] = {"source_sha": bench.source_sha(), "tag": "$TAG", "config": "$CFG",
                   "long": "rocprofv3 --kernel-trace --stats -- $CMD", "short": "rocprofv3 --kernel-trace --stats -- $CMDS",
                   "note": "per-kernel averages over the dispatches between the two marker kernels of bench.py's timed region"}
    json.dump(tr, open(t, "w"), indent=1, sort_keys=True)
PYEOF
grep -h '^{"metric"' $OUT/${TAG}${SUF}_bench.log | head -1 | cut -c1-400
