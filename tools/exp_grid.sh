for g in "X=1" "X=2"; do
  env $g timeout 200 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --quiet > gpurun_out/e.log 2>&1
  python - <<PY
import json
ok=False
for line in open('gpurun_out/e.log'):
    if line.startswith('{"metric"'):
        d=json.loads(line); ok=True; k=d['roofline']['kernels_untimed_pass']
        print("$g", round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms']*1e3,1), {n:round(v['ms_with_event_overhead']*1e3,1) for n,v in k.items()})
if not ok: print("$g", "FAILED", open('gpurun_out/e.log').read()[-500:])
PY
done
