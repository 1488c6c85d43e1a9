#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box: kernel trace of the default-size bench run (summary of
# the timed region + a two-frame timeline) and separate PMC passes.  Usage: bash tools/profile_round.sh <tag>
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/pmc_traffic.json
CMD="python bench.py --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --quiet"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o run -- $CMD > $OUT/${TAG}_bench.log 2>&1
python tools/prof_summary.py /tmp/prof_trace $OUT/${TAG}_bench_timed_region_summary.md > /dev/null
python tools/prof_timeline.py /tmp/prof_trace 2 $OUT/${TAG}_timeline_two_frames.md > /dev/null
cp "$(ls /tmp/prof_trace/*/*kernel_stats.csv /tmp/prof_trace/*kernel_stats.csv 2>/dev/null | tail -1)" $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null
CMDS="python bench.py --steps 20 --warmup 5 --cpu-frames 0 --host-frames 0 --quiet"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/prof_pmc
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_pmc -o run -- $CMDS > /tmp/pmc_$N.log 2>&1
  python tools/pmc_summary.py /tmp/prof_pmc $OUT/${TAG}_${N}.md $OUT/pmc_traffic.json > /dev/null || tail -5 /tmp/pmc_$N.log
done
# stamp the PMC file with the kernel sources it was collected on (bench.py refuses it for any other build) and the map size
python - <<EOF
import json, sys
sys.path.insert(0, ".")
import bench
p = "$OUT/pmc_traffic.json"
d = json.load(open(p))
line = [l for l in open("$OUT/${TAG}_bench.log") if l.startswith('{"metric"')]
slots = json.loads(line[0])["distributions"]["surfels_size"] if line else None
d["_meta"] = {"source_sha": bench.source_sha(), "tag": "$TAG", "surfel_slots": slots,
              "command": "rocprofv3 --pmc <counter> -- $CMDS (one pass per counter group)"}
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
EOF
grep -h '^{"metric"' $OUT/${TAG}_bench.log | head -1 | cut -c1-400
