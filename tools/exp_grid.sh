for g in "" "SMX_PRE_NORMAL=1"; do
  env $g timeout 200 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --quiet > gpurun_out/e.log 2>&1
  python - <<PY
import json
for line in open('gpurun_out/e.log'):
    if line.startswith('{"metric"'):
        d=json.loads(line); k=d['roofline']['kernels_untimed_pass']
        print("$g", round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms']*1e3,1))
PY
done
