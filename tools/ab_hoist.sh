cd "$GRAFT_REPO_ROOT"
for rep in $(seq 14); do
  for f in "--ub hoist-pre" ""; do
    timeout 300 python bench.py --config C2 --steps 300 --warmup 20 --cpu-frames 0 --host-frames 0 --timing-frames 0 --growth-frames 0 --no-other-configs --quiet $f 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-16s %.1f' % ('$f', d['value']))"
  done
done | sort -k1,1 -s
