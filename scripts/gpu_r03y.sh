export TMPDIR=/tmp; mkdir -p gpurun_out
SECONDS=0; timeout 900 python bench.py > gpurun_out/r03y_bench.json 2> gpurun_out/r03y_bench.err; echo "bench rc=$? wall=${SECONDS}s"
tail -1 gpurun_out/r03y_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','realtime_factor')})
print('roofline', d.get('roofline'))
c=d['config']
print('kernel', c.get('kernel'), c.get('groups_in_flight_per_cluster'))
print('single', [(x['N_frames'], x['us_per_step'], x['realtime_factor']) for x in c.get('single_utterance',[])])
print('raw', c.get('raw',{}).get('realtime_factor'), 'config5', c.get('config5',{}).get('realtime_factor'))
print('config3', c.get('config3'))
print('cpu', d.get('cpu_baseline'))
"
