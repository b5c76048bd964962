#!/bin/bash
# GPU session of wrnn_duo_kernel: a bounded parity subset, the placement read-out, a timing sweep against wrnn_loop_kernel.
TAG=${1:-r03b}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== parity (duo subset)"
timeout 420 python -m pytest tests/test_gpu_parity.py -q -x -s -k "${2:-duo}" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tee gpurun_out/${TAG}_duo_parity.log | tail -8
echo "== placement"
timeout 120 python scripts/gpu_duo_placement.py 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -12
echo "== probe"
timeout 240 python scripts/gpu_perf_probe.py --T 1000 --B ${3:-128,256,512} --variants ${4:-g2,d2,g4,d4,g8,d8} --out gpurun_out/${TAG}_probe_duo.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-230 | tail -20
