#!/bin/bash
# The round's evidence in one GPU-box call: GPU tests, profile rounds (C2, C3, C5: kernel trace + separate PMC passes), the
# driver's bench command (C2 line with C3 / C5 embedded, reading the PMC files of this same build), the plain lines of every
# config, the full-size reference pin.   bash tools/final_evidence.sh <tag>
TAG=${1:-rXX}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/${TAG}_gputests.txt
for C in C2 C3 C5; do bash tools/profile_round.sh $TAG $C > /dev/null 2>&1; done
cp gpurun_out/pmc_traffic*.json gpurun_out/trace_timed_region*.json profiles/ 2>/dev/null
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${TAG}_driver_cmd.err | tail -1 > gpurun_out/${TAG}_driver_cmd_bench_line.json
for f in bench_detail.json bench_detail_C3.json bench_detail_C5.json; do cp gpurun_out/$f gpurun_out/${TAG}_driver_cmd_$f 2>/dev/null; done
timeout 900 python bench.py --no-other-configs 2> gpurun_out/${TAG}_c2.err | tail -1 > gpurun_out/${TAG}_c2_bench_line.json
cp gpurun_out/bench_detail.json gpurun_out/${TAG}_c2_bench_detail.json
timeout 900 python bench.py --config C3 2> gpurun_out/${TAG}_c3.err | tail -1 > gpurun_out/${TAG}_c3_bench_line.json
cp gpurun_out/bench_detail_C3.json gpurun_out/${TAG}_c3_bench_detail.json
timeout 900 python bench.py --config C5 2> gpurun_out/${TAG}_c5.err | tail -1 > gpurun_out/${TAG}_c5_bench_line.json
cp gpurun_out/bench_detail_C5.json gpurun_out/${TAG}_c5_bench_detail.json
timeout 300 python bench.py --config C1 2> gpurun_out/${TAG}_c1.err | tail -1 > gpurun_out/${TAG}_c1_bench_line.json
timeout 600 python tests/tools/ref_pin_fullsize.py 3 gpurun_out/${TAG}_ref_pin_fullsize.json > gpurun_out/${TAG}_ref_pin_fullsize.txt 2>&1
cat gpurun_out/${TAG}_gputests.txt
for f in driver_cmd c2 c3 c5 c1; do cut -c1-260 gpurun_out/${TAG}_${f}_bench_line.json; echo; done
tail -3 gpurun_out/${TAG}_ref_pin_fullsize.txt
