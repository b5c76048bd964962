#!/bin/bash
# Round 4, session f: RAW (the bit-exact mode) on the duo kernel: every RAW parity test, then RAW timing against the loop kernel.
TAG=r04f
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out
echo "== RAW parity"; timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=6 -k "RAW or raw" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -25 | tee $OUT/${TAG}_parity_raw.log
echo "== MoL regression (bench workload)"; timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "bench_workload" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -4 | tee $OUT/${TAG}_parity_mol.log
echo "== probe RAW"; timeout 600 python scripts/gpu_perf_probe.py --mode RAW --T 600 --B 12,128,256 --variants g1,d1,g2,d2,g4,d4,d4lf --out $OUT/${TAG}_probe_raw.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee $OUT/${TAG}_probe_raw.log
