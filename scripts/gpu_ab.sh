#!/bin/bash
# A/B timing of loop-kernel builds on the GPU box:  scripts/gpu_ab.sh <tag> "<probe args>" <so> [<so> ...]   ('-' = the in-tree library)
# Every build runs the same scripts/gpu_perf_probe.py sweep; logs: gpurun_out/<tag>_ab_<name>.log / .json
TAG=$1; ARGS=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
for SO in "$@"; do
  if [ "$SO" = "-" ]; then NAME=tree; SOARG=""; else NAME=$(basename $SO .so | sed 's/libwavernn_amd_//'); SOARG="--so $SO"; fi
  echo "== $NAME"
  timeout 600 python scripts/gpu_perf_probe.py --out gpurun_out/${TAG}_ab_${NAME}.json $ARGS $SOARG 2>&1 | grep -v '^Trainable\|amdgpu.ids' |
    python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.rstrip()[:200]); continue
    print(r.get('variant'), r.get('B'), r.get('us_per_round_step', r.get('error')), (r.get('info') or {}).get('kernel', ''), 'dev', r.get('max_dev_vs_first'))
" | tee gpurun_out/${TAG}_ab_${NAME}.log
done
