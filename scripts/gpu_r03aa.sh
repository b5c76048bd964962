export TMPDIR=/tmp; mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 > gpurun_out/r03aa_torchrun.json 2> gpurun_out/r03aa_torchrun.err; echo "rc=$? wall=${SECONDS}s"
tail -1 gpurun_out/r03aa_torchrun.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','n_gpus','steps','ms_per_step','scaling')}, d['roofline']['frac'], d['roofline']['traffic'], d['config']['parallelism'][:60], d['config'].get('config3',{}).get('realtime_factor'))
"
tail -3 gpurun_out/r03aa_torchrun.err | cut -c1-200
