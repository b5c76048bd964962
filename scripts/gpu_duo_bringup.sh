#!/bin/bash
# First GPU session of wrnn_duo_kernel: a bounded parity subset, then a timing sweep against wrnn_loop_kernel in the same process.
TAG=${1:-r03b}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== parity (duo subset)"
timeout 420 python -m pytest tests/test_gpu_parity.py -q -x -s -k "duo" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tee gpurun_out/${TAG}_duo_parity.log | tail -25
echo "== probe"
timeout 240 python scripts/gpu_perf_probe.py --T 1000 --B 128,256,512 --variants g2,d2,g4,d4,g8,d8 --out gpurun_out/${TAG}_probe_duo.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -20
