for g in 0; do
  SMX_EXP=$g timeout 200 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --quiet > gpurun_out/e$g.log 2>&1
  python - <<PY
import json
for line in open('gpurun_out/e$g.log'):
    if line.startswith('{"metric"'):
        d=json.loads(line); k=d['roofline']['kernels_untimed_pass']
        print($g, round(d['value'],1), {n:round(v['ms_with_event_overhead']*1e3,1) for n,v in k.items() if n in ('neighbor_scan','reg_accumulate','reg_step')})
PY
done
