export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --utterances 32 --steps 3 --warmup 1 --no-single --no-cpu-baseline > gpurun_out/r03ah_bench_512seg.json 2> gpurun_out/r03ah.err; echo rc=$?
tail -1 gpurun_out/r03ah_bench_512seg.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['realtime_factor'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['segments_per_gpu'], d['config']['groups_in_flight_per_cluster'])
"
