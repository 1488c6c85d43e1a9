#!/bin/bash
# Round 4, first GPU-box call: GPU tests at the new HEAD, the three upper-bound runs (C2 and C3, interleaved with the plain
# run on the same box), the counter / VALU-issue calibration, and the blend's LDS / wait counters.   bash tools/r17_first.sh
TAG=r17a
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/${TAG}_gputests.txt
for cfg in C2 C3; do
  rm -f $OUT/${TAG}_ub_$cfg.jsonl
  for rep in 1 2; do
    for ub in "" hoist-pre no-reg hoist-pre,no-reg front-only hoist-pre,front-only; do
      flags="--config $cfg --cpu-frames 0 --host-frames 0 --quiet"; [ -n "$ub" ] && flags="$flags --ub $ub"
      timeout 300 python bench.py $flags 2>>$OUT/${TAG}_ub.err | tail -1 >> $OUT/${TAG}_ub_$cfg.jsonl
    done
  done
done
python - <<'PY' > gpurun_out/r17a_upper_bounds_raw.md
import json
for cfg in ("C2", "C3"):
    print("## %s" % cfg)
    for l in open("gpurun_out/r17a_ub_%s.jsonl" % cfg):
        if not l.startswith("{"): continue
        d = json.loads(l)
        ub = ",".join(d.get("upper_bound", [])) or "(plain)"
        extra = ""
        if "roofline" in d:
            k = d["roofline"]["kernels"]
            extra = " | in-frame us: " + " ".join("%s %.0f" % (n[:10], 1e3 * (v.get("in_frame_ms") or 0)) for n, v in k.items())
            extra += " | alone us: " + " ".join("%s %.1f" % (n[:10], 1e3 * v["alone_ms"]) for n, v in k.items())
        else:
            extra = " | in-frame us (before the bound): " + " ".join("%s %.0f" % (n[:10], 1e3 * v) for n, v in d["in_frame_ms_before_the_bound_was_applied"].items())
        print("- %-24s %8.1f frames/s  %.4f ms%s" % (ub, d["value"], d["ms_per_step"], extra))
PY
cat gpurun_out/r17a_upper_bounds_raw.md | cut -c1-400
bash tools/calib.sh r17 | tail -60
# the blend's "LDS / barriers" bound as counters (one short pass per group)
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/prof_pmc
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_pmc -o run -- python bench.py --steps 20 --warmup 5 --cpu-frames 0 --host-frames 0 --quiet > /tmp/pmc_$N.log 2>&1
  python tools/pmc_summary.py /tmp/prof_pmc $OUT/${TAG}_${N}.md > /dev/null || tail -5 /tmp/pmc_$N.log
done
cat $OUT/${TAG}_gputests.txt
