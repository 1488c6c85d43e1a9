#!/bin/bash
# What are the shader / memory clocks and the power draw while the bench's timed region runs?  (A long timed region,
# rocm-smi sampled beside it.)   bash tools/clock_probe.sh
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( python bench.py --steps 20000 --warmup 20 --cpu-frames 0 --host-frames 0 --quiet 2>/dev/null | tail -1 | cut -c1-200 ) &
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|mclk\|power\|GPU use\|fclk" | tr '\n' ';'; echo
  sleep 0.4
done
wait $BP
echo "idle:"; /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ';'; echo
