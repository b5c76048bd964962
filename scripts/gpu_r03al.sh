export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --mode RAW --steps 3 --warmup 1 --no-single --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03al_bench_raw.json
timeout 600 python bench.py --corpus config4 --steps 2 --warmup 1 --no-single --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03al_bench_config4.json
for f in gpurun_out/r03al_bench_raw.json gpurun_out/r03al_bench_config4.json; do python -c "
import json,sys
d=json.load(open('$f'))
print('$f', d['value'], d['ms_per_step'], d.get('realtime_factor'), d['roofline']['frac'], d['config'].get('kernel'), d['config'].get('segments_per_gpu'), d['scaling'])
"; done
