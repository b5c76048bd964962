#!/bin/bash
# Round 4, session d: conditioning formed in the loop (f1) + fc3 tile 0 in LDS: parity of the duo paths, timing, bench line.
TAG=r04d
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out
echo "== parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "duo or auto or MOL or mol or full_size or corpus" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -8 | tee $OUT/${TAG}_parity.log
echo "== probe"; timeout 300 python scripts/gpu_perf_probe.py --T 1500 --B 12,128,192,256,512 --variants d1,d2,d3,d4,d8 --out $OUT/${TAG}_probe.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee $OUT/${TAG}_probe.log
echo "== bench"; timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/${TAG}_bench.json | cut -c1-400
