export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --steps 2 --warmup 1 --no-single --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_source'))
"
