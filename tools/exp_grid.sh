for g in "SMX_GRID_FRONT=4096" "SMX_GRID_FRONT=2048" "SMX_GRID_FRONT=1024" "SMX_GRID_FRONT=512"; do
  env $g timeout 200 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --quiet > gpurun_out/e.log 2>&1
  python - <<PY
import json
ok=False
for line in open('gpurun_out/e.log'):
    if line.startswith('{"metric"'):
        d=json.loads(line); ok=True
        print("$g", round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms']*1e3,1))
if not ok: print("$g", "FAILED", open('gpurun_out/e.log').read()[-500:])
PY
done
