#!/bin/bash
# the PCIe-inclusive pass: staged copy kernels against the copy engine in the preprocessing queue
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2 3; do
for m in "" "--copy-engine-uploads"; do
  timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --cpu-frames 0 --host-frames 200 --timing-frames 0 --growth-frames 0 --quiet $m 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['host_frames']
print('%-24s resident %7.1f  PCIe-inclusive %7.1f  (%s)' % ('$m' or 'staged', d['value'], h['value'], h['uploads']))"
done
done
