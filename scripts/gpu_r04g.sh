#!/bin/bash
# Round 4, session g: stage order at depth 1-3 once more on the final kernel (MOL and RAW), tests touched by the planner change.
TAG=r04g
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out
echo "== probe MOL"; timeout 300 python scripts/gpu_perf_probe.py --T 2000 --B 12,128,192 --variants d1lf,d1pf,d2lf,d2pf,d3lf,d3pf --out $OUT/${TAG}_probe_mol.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee $OUT/${TAG}_probe_mol.log
echo "== probe RAW"; timeout 300 python scripts/gpu_perf_probe.py --mode RAW --T 600 --B 12,128,256,512 --variants d1lf,d1pf,d2lf,d2pf,d4lf,d4pf,d8lf,d8pf --out $OUT/${TAG}_probe_raw.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee $OUT/${TAG}_probe_raw.log
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sparse or prune or inplace or pack" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -5 | tee $OUT/${TAG}_tests.log
