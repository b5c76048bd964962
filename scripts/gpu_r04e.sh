#!/bin/bash
# Round 4, session e: per-segment / per-slab aux tables (f1 part 2), caller-side degrade: parity of everything that runs the duo kernel.
TAG=r04e
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out
echo "== parity"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py -m gpu -q -x -k "duo or auto or MOL or mol or full_size or corpus or workspace or continuation or dist" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -8 | tee $OUT/${TAG}_parity.log
echo "== probe"; timeout 300 python scripts/gpu_perf_probe.py --T 1500 --B 12,256,512 --variants d1,d4,d8 --out $OUT/${TAG}_probe.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee $OUT/${TAG}_probe.log
