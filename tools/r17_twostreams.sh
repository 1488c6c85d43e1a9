#!/bin/bash
# Two ranks of bench.py on ONE GPU over gloo (the C4 path without the 8-GPU node), at the full C2 size, beside the plain
# single-stream run on the same box: how much of the chip does one stream's dependency chain leave idle?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_multistream_gloo.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2; do
  timeout 300 python bench.py --cpu-frames 0 --host-frames 0 --quiet 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one stream  ', round(d['value'],1), d['n_gpus'])"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --cpu-frames 0 --host-frames 0 --quiet 2>$OUT/r17g_two.err | tail -1 > $OUT/r17g_two_streams_one_gpu_$rep.json
  python -c "import sys,json; d=json.loads(open('$OUT/r17g_two_streams_one_gpu_$rep.json').read()); print('two streams ', round(d['value'],1), d['n_gpus'], d['ms_per_step'])"
done
